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


def test_reanalyze_at_full_size_config5(oracle, tmp_path):
    """BASELINE configs[4] / SURVEY 8d config 5 at full size on one GPU (reanalyze/src/main.rs:38,135-177): a replay file of
    >= 128 000 positions (random self-play of 4096 games, written by the native driver), every pre-move state expanded on the
    device, replay line i belonging to rank i mod world; one iteration samples 4096 positions without replacement, resets
    every tree and searches 1 600 simulations with net5.  Checked through size-independent properties: the rank split is a
    partition of the single-rank buffer, the head of the batch is bit-identical to a K-position batch of the same positions,
    that batch is bit-identical to the oracle search over the same network, conservation laws hold over all 4096 trees,
    and the 4096 target lines parse, list exactly the root's children and carry the value the oracle's restatement gives."""
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import reanalyze as RA
    from takzero_amd import selfplay as SP
    from takzero_amd import weights as W

    n, B, sims, K, MIN_POSITIONS = 5, 4096, 1600, 8, 128_000
    # the replay file: uniformly random games from the openings (the Dummy agent's search is never used for the moves)
    gen = A.BatchedMCTS(B, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    sp = SP.NativeSelfPlay(gen, 0, seed=5, search="random")
    rpath = tmp_path / "replays.txt"
    with open(rpath, "wb") as f:
        while True:
            sp.play_move()
            f.write(sp.take_text(1))
            sp.take_text(0)
            if sp.counters()["replays"] >= 4200:
                break
    sp.close()
    gen.close()
    lines = open(rpath, "rb").read().splitlines()
    net = A.Net(arch=A.ARCH_NET5)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    big = A.BatchedMCTS(B, n, 4, agent=net)
    ra = RA.NativeReanalyze(big, sims, seed=3, rank=0, world=1, search="puct")
    total = ra.feed(rpath)
    moves_in_file = sum(len(ln.split()) - 5 for ln in lines)   # [TPS "a b c"] m1 ... result: tokens minus the tag's 4 and the result
    assert total == moves_in_file and total >= MIN_POSITIONS, (total, moves_in_file)
    # 8 GPUs: replay line i belongs to rank i mod 8 - the eight buffers partition the single-rank one
    split = []
    small_eng = A.BatchedMCTS(256, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    for r in range(8):
        rr = RA.NativeReanalyze(small_eng, 16, seed=3, rank=r, world=8, search="puct")
        split.append(rr.feed(rpath))
        assert split[-1] == sum(len(ln.split()) - 5 for ln in lines[r::8])
        rr.close()
    small_eng.close()
    assert sum(split) == total and min(split) > total // 10
    # one iteration at full size
    ra.iterate()
    text = ra.take_text()
    bi = big.root_info()
    width = int(bi["n_children"].max())
    bc = big.root_children(width)
    states = big.get_positions()
    assert len({s.tobytes() for s in states}) > 0.95 * B          # sampled without replacement from distinct games' positions
    # head K == a K-position batch of the same positions == the oracle over the same network
    small = A.BatchedMCTS(K, n, 4, agent=net)
    ora = O.OracleSearch(oracle, K, n, 4, agent_kind=0, agent_fn=_agent_over(net))
    small.set_positions(np.arange(K), states[:K])
    ora.set_positions(np.arange(K), states[:K])
    zero = np.zeros(K, np.float32)
    small.simulate(zero, sims)
    ora.simulate(zero, sims)
    assert_same_roots(small, ora, "K-position batch")
    si, sc = small.root_info(), small.root_children(width)
    for f in ("visit_count", "n_children", "eval_tag", "eval_bits", "std_dev"):
        assert np.array_equal(bi[f][:K], si[f]), f
    for f in ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev"):
        assert np.array_equal(bc[f][:K], sc[f]), f
    # conservation over all 4096 trees (fresh trees: a root has exactly `sims` visits unless the solver proved it earlier)
    open_ = bi["eval_tag"] == A.EVAL_VALUE
    assert open_.sum() > 0.8 * B
    assert np.array_equal(bi["visit_count"][open_], np.full(int(open_.sum()), sims, np.uint32))
    valid = np.arange(width)[None, :] < bi["n_children"][:, None]
    assert np.array_equal((bc["visits"] * valid).sum(axis=1)[open_], (bi["visit_count"] - 1)[open_])
    used, cap = big.pool_usage()
    assert used < cap and big.pool_overflows() == 0
    # the 4096 targets: one per position, in batch order; policy = improved_policy(most_visited_count()) over the children
    targets, consumed, skipped = F.parse_targets(text, n, 4)
    assert len(targets) == B and skipped == 0 and consumed == len(text)
    best = big.select_best_actions()
    for g in list(range(K)) + list(range(K, B, 97)):
        st, mv, pol, value, ube = targets[g]
        nc = int(bi["n_children"][g])
        # (TPS does not carry reversible_plies, target.rs:322-326: compare the positions as text)
        assert A.state_to_tps(st) == A.state_to_tps(states[g]) and np.array_equal(mv, bc["move_idx"][g, :nc])
        assert abs(float(np.sum(pol, dtype=np.float64)) - 1.0) < 1e-4 and -1.0 <= value <= 1.0 and 0.0 <= ube <= 4.0
        if bi["eval_tag"][g] == A.EVAL_VALUE:   # value = the selected (= best) child's evaluation, negated first (reanalyze/src/main.rs:188-195)
            j = int(np.nonzero(bc["move_idx"][g, :nc] == best[g])[0][0])
            tag, bits = int(bc["eval_tag"][g, j]), bc["eval_bits"][g, j]
            if tag == A.EVAL_VALUE:
                want = -A.eval_to_f32(tag, bits)
            else:
                want = A.eval_to_f32({A.EVAL_WIN: A.EVAL_LOSS, A.EVAL_LOSS: A.EVAL_WIN, A.EVAL_DRAW: A.EVAL_DRAW}[tag], int(bits) + 1)
            assert np.float32(value) == np.float32(want), (g, value, want)
    sims_total, evals = big.counters()
    print("config 5 at full size: %d positions in the buffer, split over 8 ranks %r, %d simulations, %d network leaves"
          % (total, split, sims_total, evals))
