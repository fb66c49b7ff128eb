"""puzzle binary (puzzle/src/main.rs): SQLite queries on CPU; the batch solver against the oracle on the GPU."""
import sqlite3

import numpy as np
import pytest


def _make_db(path, rows):
    con = sqlite3.connect(str(path))
    con.execute("CREATE TABLE games (id INTEGER PRIMARY KEY, size INTEGER)")
    con.execute("CREATE TABLE puzzles (game_id INTEGER, tps TEXT, solution TEXT, tinue_length INTEGER, "
                "tinue_avoidance_length INTEGER, tiltak_2komi_eval REAL, tiltak_2komi_second_move_eval REAL)")
    for i, (size, tps, sol, tl, al, e1, e2) in enumerate(rows):
        con.execute("INSERT INTO games VALUES (?, ?)", (i + 1, size))
        con.execute("INSERT INTO puzzles VALUES (?, ?, ?, ?, ?, ?, ?)", (i + 1, tps, sol, tl, al, e1, e2))
    con.commit()
    con.close()


CAPS = "x6/x6/x2,1C,2C,x2/x6/x6/x6 1 3"


def test_puzzle_queries_follow_the_reference_filters(tmp_path):
    import takzero_amd.api as A
    from takzero_amd import puzzle as P

    db = tmp_path / "puzzles.db"
    _make_db(db, [
        (6, CAPS, "a1", 3, None, 0.1, 0.1),        # tinue 3: selected
        (6, CAPS, "b1", 3, None, 0.1, 0.9),        # second-move eval too high
        (6, "x6/x6/x6/x6/x6/x6 1 1", "c1", 3, None, 0.1, 0.1),  # no capstones on the board
        (5, CAPS, "d1", 3, None, 0.1, 0.1),        # wrong size
        (6, CAPS, "e1", 3, 2, 0.1, 0.1),           # has an avoidance length: not a pure tinue
        (6, CAPS, "f1", 5, None, 0.1, 0.1),        # other depth
        (6, CAPS, "a2", None, 2, 0.2, 0.9),        # avoidance 2: selected (filters on tiltak_2komi_eval)
        (6, CAPS, "b2", None, 2, 0.7, 0.1),        # eval too high
    ])
    st, sol = P.load_puzzles(db, "tinue", 3)
    assert [A.move_to_ptn(6, int(m)) for m in sol] == ["a1"]
    assert A.state_to_tps(st[0]) == CAPS
    st, sol = P.load_puzzles(db, "tinue", 5)
    assert [A.move_to_ptn(6, int(m)) for m in sol] == ["f1"]
    st, sol = P.load_puzzles(db, "avoidance", 2)
    assert [A.move_to_ptn(6, int(m)) for m in sol] == ["a2"]
    assert len(P.load_puzzles(db, "avoidance", 4)[0]) == 0
    r = P.PuzzleResult(4, 3, 1)
    assert r.solve_rate() == 0.75 and r.prove_rate() == 0.25


@pytest.mark.gpu
def test_benchmark_counts_match_the_oracle():
    import oracle_lib as O
    from gpu_util import random_positions, require_gpu

    A = require_gpu()
    from takzero_amd import puzzle as P
    from takzero_amd import weights as W
    from test_gpu_engine import _agent_over

    oracle = O.load()
    n, B = 4, 16
    net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_F16, blocks=1)
    net.load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=4))
    # 40 positions (2.5 batches: the last batch is short), close enough to the end that the solver proves some
    states = O.states_array(random_positions(oracle, O, n, 4, 40, 21, min_ply=8, max_ply=16))
    # "solutions": what a deeper oracle search picks; only the agreement of the two engines is under test
    ref = O.OracleSearch(oracle, 40, n, 4, agent_kind=1)
    ref.set_positions(np.arange(40), states)
    ref.simulate(np.zeros(40, np.float32), 200)
    solutions = ref.select_best_actions()
    got = []
    for engine in ("gpu", "oracle"):
        for win in (True, False):
            if engine == "gpu":
                m = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 13)
            else:
                m = O.OracleSearch(oracle, B, n, 4, agent_kind=0, agent_fn=_agent_over(net))
            r = P.benchmark(m, states, solutions, win, 8, 48, np.random.default_rng(P.SEED))
            got.append((engine, win, r.attempted, r.solved, r.proven))
    assert [g[2:] for g in got[:2]] == [g[2:] for g in got[2:]], got
    assert got[0][2] == 40


@pytest.mark.gpu
def test_tinue_in_one_is_solved_and_proven():
    from gpu_util import require_gpu

    A = require_gpu()
    from takzero_amd import puzzle as P

    n = 3
    m = A.BatchedMCTS(4, n, 0, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    tps = ["x3/x3/1,1,x 1 3", "x3/1,1,x/x3 1 3"]
    states = np.array([A.state_from_tps(t, n, 0) for t in tps], dtype=A.STATE_DTYPE)
    sol = np.array([A.move_from_ptn(n, "c1"), A.move_from_ptn(n, "c2")], np.uint16)
    r = P.benchmark(m, states, sol, True, 16, 64, np.random.default_rng(1))  # 4 sampled root actions would miss the move
    assert (r.attempted, r.solved, r.proven) == (2, 2, 2)
