"""GPU tests of the drivers around the hot path: self-play target/replay files, validated replay expansion,
reanalyze iteration (BASELINE configs 1, 4, 5 at test sizes)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu

pytestmark = pytest.mark.gpu


def test_play_moves_validates_like_the_oracle(oracle):
    A = require_gpu()
    n, hk, B = 5, 4, 64
    gpu = A.BatchedMCTS(B, n, hk, agent_kind=A.AGENT_DUMMY, node_capacity=256)
    rng = np.random.default_rng(11)
    states = random_positions(oracle, O, n, hk, B, 21, min_ply=0, max_ply=45)
    gpu.set_positions(np.arange(B), O.states_array(states))
    for it in range(6):
        acts = rng.integers(0, A.policy_size(n), B).astype(np.uint16)
        legal = [O.possible_moves(oracle, s) for s in states]
        for g in range(0, B, 2):  # make half of them legal on purpose
            acts[g] = legal[g][int(rng.integers(len(legal[g])))] if legal[g] else acts[g]
        acts[5] = 0xFFFF
        ok = gpu.play_moves(acts)
        got = gpu.get_positions()
        for g in range(B):
            term = oracle.tzo_terminal(C.byref(states[g])) != -1
            if acts[g] == 0xFFFF:
                want = 0
            elif term:
                want = -1
            else:
                want = 1 if int(acts[g]) in legal[g] else 0
            assert ok[g] == want, (it, g)
            if want == 1:
                states[g] = O.play(oracle, states[g], int(acts[g]))
            assert got[g].tobytes() == bytes(states[g]), (it, g)


def _selfplay_run(A, tmp_path, n=4, B=48, sims=12, moves=70, search="puct"):
    from takzero_amd import formats as F
    from takzero_amd import selfplay as SP
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_BF16, blocks=1)
    net.load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=7))
    mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 13)
    sp = SP.SelfPlay(mcts, sims, seed=3, search=search, sampled_actions=4 if search != "puct" else 64)
    tpath, rpath = tmp_path / "targets-selfplay.txt", tmp_path / "replays.txt"
    nt = nr = 0
    with open(tpath, "w") as tf, open(rpath, "w") as rf:
        for _ in range(moves):
            targets, replays = sp.play_move()
            for st, mv, pol, value, ube in targets:
                tf.write(F.format_target(n, st, mv, pol, value, ube))
                nt += 1
            for start, acts, result in replays:
                rf.write(F.format_replay(n, start, acts, result))
                nr += 1
    return net, mcts, sp, tpath, rpath, nt, nr


def test_selfplay_writes_parseable_files_and_replays_revalidate(oracle, tmp_path):
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import reanalyze as RA

    n = 4
    net, mcts, sp, tpath, rpath, nt, nr = _selfplay_run(A, tmp_path, n=n)
    assert nr > 0 and nt > 0, "some games should have finished"
    # every target line parses back; policy is a distribution over exactly the legal moves of the position
    for line in open(tpath):
        st, mv, pol, value, ube = F.parse_target(line, n, 4)
        ost = O.TzState.from_buffer_copy(st.tobytes())
        assert sorted(O.possible_moves(oracle, ost)) == sorted(int(m) for m in mv)
        assert -1.0 <= value <= 1.0 and 0.0 <= ube <= 4.0
        assert abs(float(pol.sum(dtype=np.float64)) - 1.0) < 0.15  # visits / root visits: (N-1)/N with reuse
        assert F.format_target(n, st, mv, pol, value, ube) == line  # byte-stable round trip
    # every replay line re-validates on the device and through the oracle, and ends in a terminal position
    buf = RA.PositionBuffer(mcts, n, 4)
    added = buf.read_new(str(rpath))
    total_moves = 0
    for line in open(rpath):
        start, moves = F.parse_replay(line, n, 4)
        s = O.TzState.from_buffer_copy(start.tobytes())
        for m in moves:
            assert oracle.tzo_terminal(C.byref(s)) == -1
            s = O.play(oracle, s, int(m))
        assert oracle.tzo_terminal(C.byref(s)) != -1
        res = oracle.tzo_result(C.byref(s))  # 1 white, 2 black, 3 draw
        tag = line.split()[-1]
        assert tag in {1: ("R-0", "F-0"), 2: ("0-R", "0-F"), 3: ("1/2-1/2",)}[res], (tag, res)
        total_moves += len(moves)
    assert added == total_moves == len(buf.positions)
    assert buf.read_new(str(rpath)) == 0  # incremental tail: nothing new


def test_terminal_details_give_ptn_results(oracle):
    A = require_gpu()
    from takzero_amd import formats as F

    gpu = A.BatchedMCTS(3, 3, 0, agent_kind=A.AGENT_DUMMY, node_capacity=64)
    tps = ["x3/x3/1,1,x 1 3", "1,2,1/2,1S,2/1,2,x 1 5", "x3/x3/x3 1 1"]
    gpu.set_positions([0, 1, 2], np.array([A.state_from_tps(t, 3, 0) for t in tps], dtype=A.STATE_DTYPE))
    ok = gpu.play_moves(np.array([A.move_from_ptn(3, "c1"), A.move_from_ptn(3, "c1"), A.move_from_ptn(3, "a1")], np.uint16))
    assert list(ok) == [1, 1, 1]
    term = gpu.restart_terminal_envs(np.zeros(3, np.int32))
    reason, winner = gpu.terminal_details()
    assert list(term) == [A.TERMINAL_LOSS, A.TERMINAL_DRAW, A.TERMINAL_NONE]
    assert F.result_string(reason[0], winner[0]) == "R-0"
    assert F.result_string(reason[1], winner[1]) == "1/2-1/2" and reason[1] == 2


def test_reanalyze_iteration(oracle, tmp_path):
    """config 5 at test size: replay file -> position buffer -> sample -> fresh search -> one target per position."""
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import reanalyze as RA

    n = 4
    net, mcts, sp, tpath, rpath, nt, nr = _selfplay_run(A, tmp_path, n=n, moves=60)
    for search, sims in (("puct", 24), ("gumbel", 16)):
        re = RA.Reanalyze(mcts, sims, seed=1, search=search, sampled_actions=4)
        assert re.buffer.read_new(str(rpath)) >= mcts.batch
        targets = re.iterate()
        assert len(targets) == mcts.batch
        for st, mv, pol, value, ube in targets:
            ost = O.TzState.from_buffer_copy(st.tobytes())
            assert list(O.possible_moves(oracle, ost)) == [int(m) for m in mv]  # child order = possible_moves order
            assert abs(float(pol.sum(dtype=np.float64)) - 1.0) < 1e-4 and -1.0 <= value <= 1.0
            line = F.format_target(n, st, mv, pol, value, ube)
            back = st.copy()
            back["reversible_plies"] = 0  # not representable in TPS (target.rs:322-326)
            assert F.parse_target(line, n, 4)[0].tobytes() == back.tobytes()
    # the per-game entry point equals the scalar one called once per distinct visitation count
    ch = mcts.root_children()
    mvc = ch["visits"].max(axis=1).astype(np.float32)
    each = mcts.improved_policy_each(mvc, ch["visits"].shape[1])
    for v in np.unique(mvc):
        rows = mvc == v
        assert np.array_equal(each[rows], mcts.improved_policy(float(v), ch["visits"].shape[1])[rows])
    # sharding: rank r of 2 sees every second replay line
    b0, b1 = RA.PositionBuffer(mcts, n, 4, 0, 2), RA.PositionBuffer(mcts, n, 4, 1, 2)
    full = RA.PositionBuffer(mcts, n, 4)
    assert b0.read_new(str(rpath)) + b1.read_new(str(rpath)) == full.read_new(str(rpath))


def test_compete_matches_oracle(oracle):
    """evaluation::compete (evaluation/src/main.rs:224-319): two nets, two trees per game, same Gumbel draws on both
    engines -> identical game outcomes (every move of every game is the same)."""
    A = require_gpu()
    from takzero_amd import evaluation as E
    from takzero_amd import weights as W
    from test_gpu_engine import _agent_over

    n, B = 4, 12
    nets = []
    for seed in (1, 2):
        net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_F16, blocks=1)
        net.load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=seed))
        nets.append(net)
    games = random_positions(oracle, O, n, 4, B, 99, min_ply=2, max_ply=3)
    res = []
    for engine in ("gpu", "oracle"):
        if engine == "gpu":
            w = A.BatchedMCTS(B, n, 4, agent=nets[0], node_capacity=1 << 13)
            b = A.BatchedMCTS(B, n, 4, agent=nets[1], node_capacity=1 << 13)
            g = O.states_array(games)
        else:
            w = O.OracleSearch(oracle, B, n, 4, agent_kind=0, agent_fn=_agent_over(nets[0]))
            b = O.OracleSearch(oracle, B, n, 4, agent_kind=0, agent_fn=_agent_over(nets[1]))
            g = O.states_array(games)
        ev = E.compete(w, b, g, 0.0, 0.0, np.random.default_rng(5), sampled_actions=4, search_budget=16, max_moves=40)
        res.append((ev.wins, ev.losses, ev.draws, w.get_positions().tobytes(), b.get_positions().tobytes()))
    assert res[0] == res[1]
    assert sum(res[0][:3]) <= B
