"""BASELINE.json's full-size configurations through properties that do not depend on the size (the oracle cannot
finish 4096 games x 400 simulations in test time):

  * games are independent, so the first K games of the full batch must be bit-identical — visit counts, evaluations,
    priors, chosen moves — to a K-game batch fed the same openings and noise, and THAT batch is small enough to check
    against the oracle search (driven by the same HIP network);
  * conservation: simulations = games x calls, root visits = calls, children's visits add up to the root's minus the
    expansion visit, every target policy is a distribution, the node pools stay inside their capacity."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_util import require_gpu
from test_gpu_engine import _agent_over
from test_gpu_tree import assert_same_roots

pytestmark = pytest.mark.gpu


# config 2 (north star): 5x5, 4096 games, 400 simulations per move, net5.  config 4: 6x6, 2048 games, 800, net6_simhash.
@pytest.mark.parametrize("arch,n,blocks,B,sims,K", [(5, 5, 20, 4096, 400, 24), (6, 6, 16, 2048, 800, 8)])
def test_full_size_batch_agrees_with_a_small_one_and_with_the_oracle(oracle, arch, n, blocks, B, sims, K):
    A = require_gpu()
    from takzero_amd import weights as W

    net = A.Net(arch=arch, n=n, blocks=blocks)   # the default precision (fp16 storage)
    net.load_tensors(W.init_weights(arch, n=n, blocks=blocks, seed=123))
    rng = np.random.default_rng(7)
    choice = rng.integers(0, 16, B)
    amax = 512 if n < 6 else 1024
    big = A.BatchedMCTS(B, n, 4, agent=net)
    small = A.BatchedMCTS(K, n, 4, agent=net)
    ora = O.OracleSearch(oracle, K, n, 4, agent_kind=0, agent_fn=_agent_over(net))
    big.new_openings(choice)
    small.new_openings(choice[:K])
    ora.new_openings(choice[:K])
    betas = np.zeros(B, np.float32)
    for m in (big, small, ora):
        m.simulate(betas[:m.batch], 1)
    info = big.root_info()
    width = int(info["n_children"].max())
    assert width <= amax
    noise = np.zeros((B, width), np.float32)
    for g in range(B):
        k = int(info["n_children"][g])
        noise[g, :k] = rng.dirichlet([0.05] * k).astype(np.float32)
    for m in (big, small, ora):
        m.apply_noise(noise[:m.batch], 0.2)
        m.simulate(betas[:m.batch], sims)
    # (1) the small batch is the oracle's, bit for bit; (2) the head of the full batch is the small batch
    assert_same_roots(small, ora, "K-game batch")
    bi, si = big.root_info(), small.root_info()
    bc, sc = big.root_children(width), small.root_children(width)
    for f in ("visit_count", "n_children", "eval_tag", "eval_bits", "std_dev"):
        assert np.array_equal(bi[f][:K], si[f]), f
    for f in ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev"):
        assert np.array_equal(bc[f][:K], sc[f]), f
    acts = big.select_best_actions()
    assert np.array_equal(acts[:K], small.select_best_actions()) and np.array_equal(acts[:K], ora.select_best_actions())
    # conservation laws over the whole batch
    sims_total, evals = big.counters()
    open_ = bi["eval_tag"] == A.EVAL_VALUE   # a root the solver has proven stops being simulated (batched.rs:71-75)
    assert open_.sum() > 0.9 * B
    assert sims_total <= B * (sims + 1) and sims_total >= int(open_.sum()) * (sims + 1) and 0 < evals <= sims_total
    assert np.array_equal(bi["visit_count"][open_], np.full(int(open_.sum()), sims + 1, np.uint32))
    valid = np.arange(width)[None, :] < bi["n_children"][:, None]
    assert np.array_equal((bc["visits"] * valid).sum(axis=1)[open_], (bi["visit_count"] - 1)[open_])
    priors = (bc["prob"] * valid).sum(axis=1, dtype=np.float64)
    assert np.all(np.abs(priors - 1.0) < 1e-4)
    used, cap = big.pool_usage()
    assert used < cap
    # one move later the trees are still consistent (subtree reuse at full size)
    big.step(acts)
    term = big.restart_terminal_envs(rng.integers(0, 16, B))
    after = big.root_info()
    kept = term == A.TERMINAL_NONE
    chosen = (bc["move_idx"] == acts[:, None]) & valid
    assert np.array_equal(after["visit_count"][kept], (bc["visits"] * chosen).sum(axis=1)[kept])
