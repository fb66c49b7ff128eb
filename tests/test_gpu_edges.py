"""Edge cases of the search / agent boundary against the oracle: a batch of one, roots that are already terminal,
positions with very many legal moves, zero simulations, empty overwrites, a network batch that does not fill a tile,
games that end inside the batch while others go on, error codes for what the reference asserts."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu
from test_gpu_engine import _agent_over
from test_gpu_tree import assert_same_roots

pytestmark = pytest.mark.gpu


def test_batch_of_one_with_the_real_net(oracle):
    A = require_gpu()
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_TEST, n=5, precision=A.PREC_F16, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=5, blocks=1, seed=2))
    gpu = A.BatchedMCTS(1, 5, 4, agent=net, node_capacity=1 << 14)
    ora = O.OracleSearch(oracle, 1, 5, 4, agent_kind=0, agent_fn=_agent_over(net))
    for m in (gpu, ora):
        m.new_openings(np.array([5], np.int32))
        m.simulate(np.zeros(1, np.float32), 0)      # zero simulations: nothing happens
    assert gpu.root_info()["visit_count"][0] == 0 == ora.root_info()["visit_count"][0]
    for m in (gpu, ora):
        m.simulate(np.zeros(1, np.float32), 37)     # a network batch of a single position (tile of 8 boards)
    assert_same_roots(gpu, ora, "B=1")
    assert gpu.counters() == ora.counters()


def test_terminal_roots_are_not_simulated_and_mixed_batches_keep_going(oracle):
    """batched.rs:68-89: a game whose root is terminal takes no part in a simulation; the others do."""
    A = require_gpu()
    n, B = 3, 6
    gpu = A.BatchedMCTS(B, n, 0, agent_kind=A.AGENT_SIMPLE, node_capacity=1 << 12)
    ora = O.OracleSearch(oracle, B, n, 0, agent_kind=2)
    won = "1,1,1/x3/2,2,x 2 4"          # white already has a road: terminal for the side to move
    full = "1,2,1/2,1S,2/1,2,1 2 5"     # board full: flat count decides
    live = "x3/x,1,x/2,x2 1 2"
    tps = [won, live, full, live, won, live]
    states = O.states_array([O.state_from_tps(oracle, t, n, 0) for t in tps])
    for m in (gpu, ora):
        m.set_positions(np.arange(B), states)
        m.simulate(np.full(B, 0.5, np.float32), 60)
    assert_same_roots(gpu, ora, "mixed terminal / live roots")
    info = gpu.root_info()
    assert list(info["is_terminal_env"]) == [1, 0, 1, 0, 1, 0]
    assert gpu.counters() == ora.counters()
    # stepping a batch in which some games are over: those are skipped (batched.rs:137), then restarted
    acts = gpu.select_best_actions()
    assert np.array_equal(acts[[1, 3, 5]], ora.select_best_actions()[[1, 3, 5]])
    gpu.step(acts)
    ora.step(acts)
    choice = np.arange(B, dtype=np.int32)
    assert np.array_equal(gpu.restart_terminal_envs(choice), ora.restart_terminal(choice))
    assert gpu.get_positions().tobytes() == ora.get_positions().tobytes()


def test_positions_with_very_many_legal_moves(oracle):
    """Late 6x6 positions with tall stacks: hundreds of legal moves per node, children in possible_moves order."""
    A = require_gpu()
    n, B = 6, 8
    wide = ["x6/x6/x2,212121,212121,x2/x2,212121,212121,x2/x6/x6 1 30",     # four white-topped stacks of six: 592 moves
            "x6/x6/x2,212121,x3/x3,212121,x2/x6/x6 1 20",                   # two: 350 moves
            "x6/x,21212121,x4/x6/x3,2121212121,x2/x6/x6 1 30"]              # stacks taller than the carry limit
    states = [O.state_from_tps(oracle, t, n, 4) for t in wide] + random_positions(oracle, O, n, 4, B - len(wide), 9, min_ply=40,
                                                                                  max_ply=90)
    widest = max(len(O.possible_moves(oracle, s)) for s in states)
    assert widest == 592
    gpu = A.BatchedMCTS(B, n, 4, agent_kind=A.AGENT_SIMPLE, node_capacity=1 << 16)
    ora = O.OracleSearch(oracle, B, n, 4, agent_kind=2)
    for m in (gpu, ora):
        m.set_positions(np.arange(B), O.states_array(states) if m is gpu else states)
        m.simulate(np.zeros(B, np.float32), 50)
    assert_same_roots(gpu, ora, "wide roots")
    ch = gpu.root_children()
    assert ch["move_idx"].shape[1] == widest
    assert list(ch["move_idx"][0]) == list(O.possible_moves(oracle, states[0]))


def test_empty_overwrite_and_boundary_errors(oracle):
    A = require_gpu()
    from takzero_amd import weights as W

    gpu = A.BatchedMCTS(4, 4, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    gpu.new_openings(np.zeros(4, np.int32))
    before = gpu.get_positions().tobytes()
    gpu.set_positions(np.zeros(0, np.int32), np.zeros(0, A.STATE_DTYPE))  # nothing to overwrite
    assert gpu.get_positions().tobytes() == before
    with pytest.raises(A.TakzeroError):
        gpu.set_positions(np.array([4], np.int32), gpu.get_positions()[:1])  # index out of range
    with pytest.raises(A.TakzeroError):
        gpu.apply_noise(np.zeros((4, 8), np.float32), 0.2)  # noise on un-expanded roots (noise.rs:12-15 asserts)
    with pytest.raises(A.TakzeroError):
        gpu.gumbel_sequential_halving(np.zeros(4, np.float32), 4, 10, np.zeros((4, 512), np.float32))  # budget % (k log2 k)
    net = A.Net(arch=A.ARCH_TEST, n=4, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=4, blocks=1, seed=1))
    with pytest.raises(A.TakzeroError):
        net.policy_value_uncertainty(np.zeros(0, A.STATE_DTYPE), [])  # empty batch (net5.rs:226-227 asserts)
    unloaded = A.Net(arch=A.ARCH_TEST, n=4, blocks=1)
    with pytest.raises(A.TakzeroError):
        unloaded.forward_raw(gpu.get_positions())  # no weights yet


def test_fp16_storage_refuses_a_weight_outside_its_range_and_keeps_the_old_model():
    """fp16 storage (the default): a folded weight beyond 65504 would become inf; the load fails with TZ_ENUMERIC and, as with
    any failed load (selfplay/src/main.rs:112-115), the previous weights stay active; bf16 storage takes the same tensors."""
    A = require_gpu()
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=5, blocks=1, seed=2)
    net = A.Net(arch=A.ARCH_TEST, n=5, precision=A.PREC_F16, blocks=1).load_tensors(w)
    oracle = O.load()
    states = O.states_array(random_positions(oracle, O, 5, 4, 6, 4, max_ply=20))
    before = net.forward_raw(states)
    bad = {k: v.copy() for k, v in w.items()}
    name = next(k for k in bad if k.endswith("conv2d.weight") and "res_block" in k)
    bad[name].flat[3] = 1.0e6
    with pytest.raises(A.TakzeroError):
        net.load_tensors(bad)
    after = net.forward_raw(states)
    for x, y in zip(before, after):
        assert np.array_equal(x, y)
    A.Net(arch=A.ARCH_TEST, n=5, precision=A.PREC_BF16, blocks=1).load_tensors(bad)


@pytest.mark.parametrize("prec", ["f16x2", "f16c8"])
def test_split_precisions_refuse_what_they_cannot_run_and_say_so(prec):
    """The two split arithmetics live in the fused trunk launch only: a net without residual blocks is refused at the first forward
    (TZ_EINVAL, message names both precisions), and like fp16 storage they refuse a folded weight outside fp16's range at load."""
    A = require_gpu()
    from takzero_amd import weights as W

    oracle = O.load()
    states = O.states_array(random_positions(oracle, O, 4, 4, 3, 2, max_ply=10))
    net = A.Net(arch=A.ARCH_TEST, n=4, precision=A.PREC_NAMES[prec], blocks=0).load_tensors(W.init_weights(W.ARCH_TEST, n=4, blocks=0, seed=1))
    with pytest.raises(A.TakzeroError, match="at least one residual block"):
        net.forward_raw(states)
    net.close()
    w = W.init_weights(W.ARCH_TEST, n=4, blocks=1, seed=2)
    good = A.Net(arch=A.ARCH_TEST, n=4, precision=A.PREC_NAMES[prec], blocks=1).load_tensors(w)
    before = good.forward_raw(states)
    bad = {k: v.copy() for k, v in w.items()}
    name = next(k for k in bad if k.endswith("conv2d.weight") and "res_block" in k)
    bad[name].flat[3] = 1.0e6
    with pytest.raises(A.TakzeroError):
        good.load_tensors(bad)
    for x, y in zip(before, good.forward_raw(states)):
        assert np.array_equal(x, y)
    good.close()
