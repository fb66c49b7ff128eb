"""GPU parity of the HIP tree kernels against the CPU oracle, bit for bit (visit counts, evals,
priors, logits, std_dev, chosen moves), with the reference's fake agents Dummy / Simple
(agent.rs:16-87) so no network rounding is involved.  All calls go through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu

pytestmark = pytest.mark.gpu
FIELDS = ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev")


def assert_same_roots(gpu, ora, ctx=""):
    gi, oi = gpu.root_info(), ora.root_info()
    for f in ("visit_count", "n_children", "eval_tag", "eval_bits", "is_terminal_env", "ply"):
        assert np.array_equal(gi[f], oi[f]), (ctx, f, np.nonzero(gi[f] != oi[f])[0][:8], gi[f][:8], oi[f][:8])
    for f in ("std_dev", "logit", "probability"):
        assert np.array_equal(gi[f].view(np.uint32), oi[f].view(np.uint32)), (ctx, f)
    amax = max(1, int(gi["n_children"].max()))
    gc, oc = gpu.root_children(amax), ora.root_children(amax)
    for f in FIELDS:
        a, b = gc[f], oc[f]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), (ctx, f, np.argwhere(a != b)[:5])


def test_device_movegen_terminal_and_play_match_oracle(oracle):
    """Every device rule (possible_moves order, play, result) against the oracle over random playouts:
    one simulate() on a fresh tree expands the root, whose children are the legal moves in order."""
    A = require_gpu()
    for n, hk in ((3, 0), (4, 4), (5, 4), (6, 4)):
        B = 64
        gpu = A.BatchedMCTS(B, n, hk, agent_kind=A.AGENT_DUMMY, node_capacity=4096)
        rng = np.random.default_rng(n)
        states = [O.state_default(oracle, n, hk) for _ in range(B)]
        for ply in range(70):
            arr = O.states_array(states)
            gpu.set_positions(np.arange(B), arr)
            gpu.simulate(np.zeros(B, np.float32), 1)
            info = gpu.root_info()
            ch = gpu.root_children(max(1, int(info["n_children"].max())))
            nxt = []
            for g, s in enumerate(states):
                term = oracle.tzo_terminal(C.byref(s))
                assert int(info["is_terminal_env"][g]) == (term != -1), (n, ply, g, O.to_tps(oracle, s))
                if term != -1:
                    want_tag = {0: 1, 1: 2, 2: 3}[term]
                    assert info["eval_tag"][g] == want_tag and info["n_children"][g] == 0
                    nxt.append(O.state_default(oracle, n, hk))
                    continue
                mv = O.possible_moves(oracle, s)
                got = list(ch["move_idx"][g, :info["n_children"][g]])
                assert got == mv, (n, ply, g, O.to_tps(oracle, s), [O.ptn(oracle, n, m) for m in got][:10])
                nxt.append(O.play(oracle, s, mv[int(rng.integers(len(mv)))]))
            # device play: step with the same moves and compare the resulting states
            states = nxt
        gpu.close()


def test_device_step_matches_oracle_play(oracle):
    A = require_gpu()
    n, hk, B = 5, 4, 32
    gpu = A.BatchedMCTS(B, n, hk, agent_kind=A.AGENT_DUMMY, node_capacity=4096)
    ora = O.OracleSearch(oracle, B, n, hk, agent_kind=1)
    rng = np.random.default_rng(5)
    choice = rng.integers(0, 16, B)
    gpu.new_openings(choice)
    ora.new_openings(choice)
    betas = np.zeros(B, np.float32)
    for ply in range(60):
        assert gpu.get_positions().tobytes() == ora.get_positions().tobytes(), ply
        gpu.simulate(betas, 1)
        ora.simulate(betas, 1)
        info = ora.root_info()
        ch = ora.root_children(512)
        acts = np.array([ch["move_idx"][g, rng.integers(max(1, info["n_children"][g]))] for g in range(B)], np.uint16)
        gpu.step(acts)
        ora.step(acts)
        choice = rng.integers(0, 16, B)
        tg, to = gpu.restart_terminal_envs(choice), ora.restart_terminal(choice)
        assert np.array_equal(tg, to), ply


@pytest.mark.parametrize("agent,moves,limit", [(1, ["a3", "c1", "c2", "c3", "b3", "c3-"], 5000),
                                               (2, ["a3", "a1", "b1", "c1"], 50000)])
def test_find_tinue_on_device(oracle, agent, moves, limit):
    """mcts.rs:346-411 on the GPU engine: same proof, same number of simulations as the oracle."""
    A = require_gpu()
    s = O.state_default(oracle, 3, 0)
    for m in moves:
        s = O.play(oracle, s, O.from_ptn(oracle, 3, m))
    gpu = A.BatchedMCTS(1, 3, 0, agent_kind=agent, node_capacity=1 << 20)
    ora = O.OracleSearch(oracle, 1, 3, 0, agent_kind=agent)
    gpu.set_positions([0], O.states_array([s]))
    ora.set_positions([0], [s])
    beta = np.ones(1, np.float32)
    sims = 0
    while sims < limit:
        step = 100
        ora.simulate(beta, step)
        gpu.simulate(beta, step)
        sims += step
        assert_same_roots(gpu, ora, "tinue sims=%d" % sims)
        if ora.root_info()["eval_tag"][0] == 1:
            break
    info = gpu.root_info()
    assert info["eval_tag"][0] == 1, "root should be a proven win"
    ch = gpu.root_children()
    losing = [A.move_to_ptn(3, ch["move_idx"][0, i]) for i in range(info["n_children"][0]) if ch["eval_tag"][0, i] == 2]
    assert losing and (losing[0] == "b1" if agent == 1 else losing[0] in ("b2", "c2"))


@pytest.mark.parametrize("n,hk,agent,B,sims,moves", [(5, 4, 2, 48, 60, 12), (4, 4, 1, 32, 40, 10), (6, 4, 2, 16, 50, 6),
                                                     (3, 0, 2, 32, 200, 14)])
def test_selfplay_loop_bit_exact(oracle, n, hk, agent, B, sims, moves):
    """simulate -> apply_noise -> simulate xN -> select -> step -> restart, with tree reuse, beta != 0 for half the
    games; every root statistic compared after every move (batched.rs:63-203, noise.rs:10-26)."""
    A = require_gpu()
    gpu = A.BatchedMCTS(B, n, hk, agent_kind=agent, node_capacity=1 << 15)
    ora = O.OracleSearch(oracle, B, n, hk, agent_kind=agent)
    rng = np.random.default_rng(1000 + n)
    choice = rng.integers(0, 16, B)
    gpu.new_openings(choice)
    ora.new_openings(choice)
    betas = np.where(np.arange(B) < B // 2, 0.25, 0.0).astype(np.float32)
    for mv in range(moves):
        gpu.simulate(betas, 1)
        ora.simulate(betas, 1)
        info = ora.root_info()
        amax = max(1, int(info["n_children"].max()))
        noise = np.zeros((B, amax), np.float32)
        for g in range(B):
            k = int(info["n_children"][g])
            if k:
                noise[g, :k] = rng.dirichlet([0.3] * k).astype(np.float32)
        has_kids = info["n_children"] > 0
        if has_kids.all():
            gpu.apply_noise(noise, 0.2)
            ora.apply_noise(noise, 0.2)
        gpu.simulate(betas, sims)
        ora.simulate(betas, sims)
        assert_same_roots(gpu, ora, "move %d" % mv)
        ag, ao = gpu.select_best_actions(), ora.select_best_actions()
        assert np.array_equal(ag, ao), mv
        ip_g, ip_o = gpu.improved_policy(float(sims), amax), ora.improved_policy(float(sims), amax)
        assert np.array_equal(ip_g.view(np.uint32), ip_o.view(np.uint32)), mv
        assert np.array_equal(gpu.ube_target(0.25).view(np.uint32), ora.ube_target(0.25).view(np.uint32)), mv
        # play the second most natural choice sometimes so that reused subtrees vary
        acts = ao.copy()
        ch = ora.root_children(amax)
        for g in range(0, B, 3):
            k = int(info["n_children"][g])
            if k:
                acts[g] = ch["move_idx"][g, int(rng.integers(k))]
        gpu.step(acts)
        ora.step(acts)
        assert_same_roots(gpu, ora, "after step %d" % mv)
        choice = rng.integers(0, 16, B)
        assert np.array_equal(gpu.restart_terminal_envs(choice), ora.restart_terminal(choice))
    assert gpu.counters() == ora.counters()


def test_gumbel_sequential_halving_bit_exact(oracle):
    """batched.rs:207-409 with caller-supplied Gumbel noise."""
    A = require_gpu()
    n, hk, B = 5, 4, 16
    gpu = A.BatchedMCTS(B, n, hk, agent_kind=A.AGENT_SIMPLE, node_capacity=1 << 15)
    ora = O.OracleSearch(oracle, B, n, hk, agent_kind=2)
    rng = np.random.default_rng(77)
    states = random_positions(oracle, O, n, hk, B, 3, min_ply=2, max_ply=30)
    gpu.set_positions(np.arange(B), O.states_array(states))
    ora.set_positions(np.arange(B), states)
    betas = np.where(np.arange(B) % 2 == 0, 0.25, 0.0).astype(np.float32)
    for it in range(2):
        gumbel = rng.gumbel(size=(B, 512)).astype(np.float32)
        sg = gpu.gumbel_sequential_halving(betas, 16, 128, gumbel)
        so = ora.gumbel_sh(betas, 16, 128, gumbel)
        assert np.array_equal(sg, so), it
        assert_same_roots(gpu, ora, "gumbel %d" % it)
        gpu.step(sg)
        ora.step(so)
        choice = rng.integers(0, 16, B)
        assert np.array_equal(gpu.restart_terminal_envs(choice), ora.restart_terminal(choice))


def test_full_node_pool_degrades_or_errors(oracle, monkeypatch):
    """The reference's trees are unbounded; a pool here is not.  Default: the search goes on (the leaf is evaluated, not
    expanded) and the event is counted; TZ_STRICT_CAPACITY: TZ_ECAPACITY."""
    A = require_gpu()
    gpu = A.BatchedMCTS(4, 5, 4, agent_kind=A.AGENT_DUMMY, node_capacity=256)   # room for the root's children and two more nodes'
    gpu.new_openings(np.zeros(4, np.int32))
    gpu.simulate(np.zeros(4, np.float32), 20)
    info = gpu.root_info()
    assert gpu.pool_overflows() > 0 and np.array_equal(info["visit_count"], np.full(4, 20, np.uint32))
    ch = gpu.root_children()
    assert np.array_equal(ch["visits"].sum(axis=1), np.full(4, 19))   # every simulation after the first went through a child
    acts = gpu.select_best_actions()
    gpu.step(acts)                                                    # and the game goes on
    gpu.simulate(np.zeros(4, np.float32), 5)
    monkeypatch.setenv("TZ_STRICT_CAPACITY", "1")
    strict = A.BatchedMCTS(4, 5, 4, agent_kind=A.AGENT_DUMMY, node_capacity=256)
    strict.new_openings(np.zeros(4, np.int32))
    with pytest.raises(A.TakzeroError) as e:
        strict.simulate(np.zeros(4, np.float32), 20)
    assert e.value.code == -5


def test_device_f32_primitives_match_host(oracle):
    """exp / ln / sqrt / divide / powi on the device are bit-identical with the host versions the oracle uses."""
    A = require_gpu()
    import ctypes as C
    lib = A._lib.load()
    rng = np.random.default_rng(0)
    n = 200000
    a = np.concatenate([rng.random(n // 2, dtype=np.float32) * 4, rng.random(n // 2, dtype=np.float32) * 1e-3]).astype(np.float32)
    b = (rng.random(n, dtype=np.float32) * 1000 + np.float32(1e-3)).astype(np.float32)
    out = np.zeros(n, np.float32)

    def run(op, x, y):
        x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
        A.check(lib.tz_device_math(op, x.ctypes.data, y.ctypes.data, out.ctypes.data, len(x)))
        return out[:len(x)].copy()

    neg = -a * 10
    got = run(0, neg, b)
    want = np.array([oracle.tzo_expf(float(v)) for v in neg[:20000]], np.float32)
    assert np.array_equal(got[:20000].view(np.uint32), want.view(np.uint32))
    got = run(1, b, b)
    want = np.array([oracle.tzo_logf(float(v)) for v in b[:20000]], np.float32)
    assert np.array_equal(got[:20000].view(np.uint32), want.view(np.uint32))
    assert np.array_equal(run(2, a, b).view(np.uint32), np.sqrt(a).view(np.uint32))
    assert np.array_equal(run(3, a, b).view(np.uint32), (a / b).view(np.uint32))
    k = np.arange(0, 600, dtype=np.float32)
    got = run(4, k, k)
    want = np.array([oracle.tzo_powif(0.997, int(v)) for v in k], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(run(5, a, b).view(np.uint32), ((a + b) * a).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("n,agent", [(4, 2), (5, 1)])
def test_nodes_below_the_root_match_the_oracle(oracle, n, agent):
    """Node.children is a public field of the reference (node/mod.rs:14-23; puzzle and visualize_search walk it): tz_search_node
    returns the node reached by a path of moves - statistics and children, bit for bit the oracle's - down the principal
    variation of every game, and refuses a path that leaves the tree."""
    A = require_gpu()
    B, sims = 12, 300
    gpu = A.BatchedMCTS(B, n, 4, agent_kind=agent, node_capacity=1 << 15)
    ora = O.OracleSearch(oracle, B, n, 4, agent_kind=agent)
    choice = np.arange(B) % 16
    gpu.new_openings(choice)
    ora.new_openings(choice)
    betas = np.where(np.arange(B) % 2 == 0, 0.0, 0.25).astype(np.float32)
    gpu.simulate(betas, sims)
    ora.simulate(betas, sims)
    deepest = 0
    for g in range(B):
        path = []
        while True:
            got, want = gpu.node(g, path), ora.node(g, path)
            assert want is not None
            for f in ("visit_count", "n_children", "eval_tag", "eval_bits", "std_dev", "logit", "probability", "ply", "is_terminal_env"):
                assert got[0][f] == want[0][f] or (np.isnan(got[0][f]) and np.isnan(want[0][f])), (g, path, f)
            for f in ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev"):
                assert np.array_equal(got[1][f].view(np.uint8), want[1][f].view(np.uint8)), (g, path, f)
            if len(got[1]["visits"]) == 0 or got[1]["visits"].max() == 0:
                break
            path.append(int(got[1]["move_idx"][int(np.argmax(got[1]["visits"]))]))   # most visited child
        deepest = max(deepest, len(path))
        root_moves = set(int(m) for m in gpu.node(g, [])[1]["move_idx"])
        absent = next(m for m in range(A.policy_size(n)) if m not in root_moves)
        with pytest.raises(A.TakzeroError):
            gpu.node(g, [absent])
        assert ora.node(g, [absent]) is None
    assert deepest >= 2
