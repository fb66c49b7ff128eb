"""Pins the CPU oracle with every known-answer test the reference holds for the hot path
(SURVEY.md §4 / §8c).  Each test names the reference test it restates."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def test_softmax_works(oracle):
    # takzero/src/search/node/policy.rs:173-187
    x = np.array([1, 2, 3, 4, 5], dtype=np.float32)
    for libm in (0, 1):
        oracle.tzo_set_use_libm(libm)
        out = np.zeros(5, np.float32)
        oracle.tzo_softmax(x.ctypes.data_as(C.POINTER(C.c_float)), 5, out.ctypes.data_as(C.POINTER(C.c_float)))
        want = np.array([0.011656231, 0.03168492, 0.08612855, 0.23412165, 0.6364086], np.float32)
        assert np.all(np.abs(out - want) < np.finfo(np.float32).eps)
    oracle.tzo_set_use_libm(0)


def _bits(v):
    return int(np.float32(v).view(np.uint32))


def test_eval_order(oracle):
    # takzero/src/search/eval.rs:170-194
    V, W, L, D = 0, 1, 2, 3
    contempt_plus = np.float32(-0.05) + np.float32(0.1)
    evals = [(V, _bits(1.0)), (V, _bits(contempt_plus)), (V, _bits(-1.0)), (W, 5), (W, 10), (D, 5), (D, 10),
             (L, 5), (L, 10)]
    import functools

    got = sorted(evals, key=functools.cmp_to_key(lambda a, b: oracle.tzo_eval_cmp(a[0], a[1], b[0], b[1])))
    want = [(L, 5), (L, 10), (V, _bits(-1.0)), (D, 10), (D, 5), (V, _bits(contempt_plus)), (V, _bits(1.0)),
            (W, 10), (W, 5)]
    assert got == want


def test_eval_to_f32(oracle):
    # eval.rs:95-105 : 0.997^ply * {v, 1, -1, 0}
    assert oracle.tzo_eval_to_f32(0, _bits(0.25)) == np.float32(0.25)
    assert oracle.tzo_eval_to_f32(1, 0) == 1.0
    assert oracle.tzo_eval_to_f32(2, 1) == -np.float32(0.997)
    assert oracle.tzo_eval_to_f32(3, 7) == 0.0
    # powi is square-and-multiply in f32 (compiler-builtins __powisf2)
    a = np.float32(0.997)
    r, base, b = np.float32(1.0), a, 13
    while True:
        if b & 1:
            r = np.float32(r * base)
        b >>= 1
        if b == 0:
            break
        base = np.float32(base * base)
    assert oracle.tzo_eval_to_f32(1, 13) == r


def _tinue(oracle, moves, agent, max_visits):
    arr = (C.c_char_p * len(moves))(*[m.encode() for m in moves])
    out = C.c_uint16()
    n = oracle.tzo_kat_find_tinue(3, 0, arr, len(moves), agent, 1.0, max_visits, C.byref(out))
    return n, out.value


def test_find_tinue_easy(oracle):
    # takzero/src/search/node/mcts.rs:346-376
    n, mv = _tinue(oracle, ["a3", "c1", "c2", "c3", "b3", "c3-"], 1, 5000)
    assert n > 0, n
    assert O.ptn(oracle, 3, mv) == "b1"


def test_find_tinue_deeper(oracle):
    # takzero/src/search/node/mcts.rs:379-411
    n, mv = _tinue(oracle, ["a3", "a1", "b1", "c1"], 2, 50000)
    assert n > 0, n
    assert O.ptn(oracle, 3, mv) in ("b2", "c2")


def test_safe_cracker_value_propagation(oracle):
    # takzero/src/search/node/mcts.rs:414-445
    assert oracle.tzo_kat_safecrack(100000) == 0


def test_distribution_stays_1_after_noise(oracle):
    # takzero/src/search/node/noise.rs:49-67 (Dirichlet sample drawn here with numpy, alpha 0.5)
    s = O.OracleSearch(oracle, 1, 3, 0, agent_kind=1)
    s.simulate([0.0], 1)
    ch = s.root_children(64)
    n = int(s.root_info()["n_children"][0])
    assert n == 9
    assert abs(ch["prob"][0, :n].sum(dtype=np.float32) - 1.0) < 1.1 * np.finfo(np.float32).eps
    rng = np.random.default_rng(123)
    noise = np.zeros((1, 64), np.float32)
    noise[0, :n] = rng.dirichlet([0.5] * n).astype(np.float32)
    s.apply_noise(noise, 0.2)
    ch = s.root_children(64)
    assert abs(np.sum(ch["prob"][0, :n], dtype=np.float32) - 1.0) < 4 * np.finfo(np.float32).eps
    sm = np.zeros(n, np.float32)
    lg = np.ascontiguousarray(ch["logit"][0, :n])
    oracle.tzo_softmax(lg.ctypes.data_as(C.POINTER(C.c_float)), n, sm.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.all(np.abs(sm - ch["prob"][0, :n]) < np.finfo(np.float32).eps)


# ---------------------------------------------------------------- repr.rs:261-409 plane encodings
def _planes(rows):
    return np.array([v for r in rows for v in r], dtype=np.float32)


def test_repr_starting_position(oracle):
    # repr.rs:262-301
    s = O.state_default(oracle, 3, 0)
    got = O.game_repr(oracle, s)
    want = np.zeros(24 * 9, np.float32)
    want[18 * 9:19 * 9] = 1.0  # my stones ratio
    want[20 * 9:21 * 9] = 1.0  # opponent stones ratio
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_repr_complicated_position(oracle):
    # repr.rs:303-360
    x, o = 1.0, 0.0
    p, q, d = np.float32(5.0) / np.float32(21.0), np.float32(10.0) / np.float32(21.0), np.float32(-3.0) / np.float32(25.0)
    Z = [o] * 25
    rows = [
        [o, o, o, x, o, o, x, o, o, o, o, x, o, o, x, x, o, x, o, o, o, o, o, o, o],
        [o, o, o, o, o, o, o, o, o, o, o, o, o, x, o, o, o, o, o, o, o, o, o, o, o],
        [o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, x, o, o, o, o, o, o, o, o],
        [o, o, x, o, o, o, o, x, o, o, o, o, x, o, o, o, o, o, o, o, o, o, x, o, o],
        [o, o, x, o, o, x, o, o, o, o, o, x, o, o, o, o, o, o, o, o, o, o, x, o, o],
        [o, o, o, o, o, x, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o],
        Z, Z, Z, Z, Z, Z, Z,
        [o, o, o, o, o, o, o, x, x, x, o, o, o, o, o, o, o, o, x, o, o, o, x, o, o],
        [o, o, x, o, o, x, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, x],
        [o, o, o, o, o, o, o, o, o, o, o, o, x, o, o, o, o, o, o, o, o, o, o, o, o],
        [o, o, o, o, o, x, o, o, o, o, o, x, o, o, o, o, o, o, o, o, o, o, o, o, o],
        Z,
        [o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, x, o, o],
        Z, Z, Z, Z, Z, Z, Z,
        [p] * 25, Z, [q] * 25, Z, [x] * 25, [d] * 25,
    ]
    want = _planes(rows)
    s = O.state_from_tps(oracle, "x2,1221,x,1S/2,2C,2,1,x/x,212,21C,2S,2/2211S,2,21,1,1/x2,221S,2,x 2 23", 5, 4)
    got = O.game_repr(oracle, s)
    assert got.shape == want.shape == (32 * 25,)
    assert np.array_equal(got, want), np.nonzero(got != want)


def test_repr_tall_stack(oracle):
    # repr.rs:362-409
    x, o = 1.0, 0.0
    p, q, d = np.float32(5.0) / np.float32(10.0), np.float32(4.0) / np.float32(10.0), np.float32(0.5) / np.float32(9.0)
    Z = [o] * 9
    M = [o, o, o, o, x, o, o, o, o]
    rows = [Z, Z, Z, M, Z, Z, M, M, Z,
            Z, M, Z, Z, M, M, Z, Z, M,
            [p] * 9, Z, [q] * 9, Z, Z, [d] * 9]
    want = _planes(rows)
    s = O.state_from_tps(oracle, "x3/x,21212112212S,x/x3 1 12", 3, -1)
    got = O.game_repr(oracle, s)
    assert np.array_equal(got, want), np.nonzero(got != want)


def test_policy_layout_and_legal_set(oracle):
    # repr.rs:413-499 : Simple agent's logits scattered by move_index for TPS 2,1,x/1S,221,x/x,2S,2 1 6
    f, w, s_, o = 4.0, 2.0, 1.0, 0.0
    Z = [o] * 9
    rows = [
        [f, o, o, o, o, f, o, o, f],
        [w, o, o, o, o, w, o, o, w],
        Z,
        [o, o, o, o, s_, o, o, o, o],  # 3#+3
        [o, o, o, o, s_, o, o, o, o],  # 2#+2
        Z,
        [o, o, o, s_, s_, o, o, o, o],  # 1#+1
        Z, Z,
        [o, o, o, o, s_, o, o, o, o],  # 3#>3
        [o, o, o, o, s_, o, o, o, o],  # 2#>2
        Z,
        [o, o, o, s_, s_, o, o, s_, o],  # 1#>1
        Z, Z,
        Z, Z, Z,
        [o, o, o, s_, o, o, o, s_, o],  # 1#-1
        Z, Z,
        Z, Z, Z,
        [o, o, o, o, o, o, o, s_, o],  # 1#<1
        Z, Z,
    ]
    want = _planes(rows)
    st = O.state_from_tps(oracle, "2,1,x/1S,221,x/x,2S,2 1 6", 3, 0)
    moves = O.possible_moves(oracle, st)
    got = np.zeros(27 * 9, np.float32)
    for m in moves:
        name = O.ptn(oracle, 3, m)
        val = 1.0 if any(c in name for c in "+-<>") else (2.0 if name.startswith("S") else 3.0 if name.startswith("C") else 4.0)
        got[m] = val
    assert got.shape == want.shape
    assert np.array_equal(got, want), np.nonzero(got != want)
    assert len(set(moves)) == len(moves) == 18


def test_move_count_table(oracle):
    # repr.rs:16-34 possible_moves::<N>() / python/action_space.py:4-42 : distinct moves per board size
    want = {3: 126, 4: 480, 5: 1575, 6: 4572}
    for n, total in want.items():
        out_ch = oracle.tzo_output_channels(n)
        assert out_ch == 3 + 4 * (2 ** n - 2)
        cnt = 0
        for idx in range(out_ch * n * n):
            name = O.ptn(oracle, n, idx)
            # a spread is on-board iff the last square it reaches exists
            ch, sq = divmod(idx, n * n)
            if ch < 3:
                cnt += ch < 2 or n >= 5  # no capstones below 5x5 (python/action_space.py:4-7)
                continue
            y, x = divmod(sq, n)
            squares = 1 if name[-1] in "+-<>" else len(name) - 1 - max(name.find(c) for c in "+-<>")
            d = next(c for c in name if c in "+-<>")
            reach = {"+": n - 1 - y, "-": y, "<": x, ">": n - 1 - x}[d]
            cnt += squares <= reach
            assert O.from_ptn(oracle, n, name) == idx
        assert cnt == total, (n, cnt)


# ---------------------------------------------------------------- move ordering (runs/*.txt data)
def _order_key(name):
    """Ordering of fast-tak's possible_moves as visible in the reference's dumps: file, rank,
    then placements Flat<Wall<Cap; spreads by carry, direction + - < >, drop string descending."""
    if not any(c in name for c in "+-<>"):
        piece = {"S": 1, "C": 2}.get(name[0], 0)
        sq = name[-2:]
        return (sq[0], sq[1], 0, piece, 0, ())
    i = max(name.find(c) for c in "+-<>")
    d = "+-<>".index(name[i])
    carry = int(name[0]) if name[0].isdigit() else 1
    sq = name[i - 2:i]
    drops = name[i + 1:] or str(carry)
    return (sq[0], sq[1], 1, carry, d, tuple(-int(c) for c in drops.ljust(8, "0")))


def test_move_order_matches_reference_dumps(oracle):
    path = os.path.join(HERE, "golden", "runs_puct_move_order.txt")
    lines = [l.split() for l in open(path)]
    assert len(lines) == 1024 and len({tuple(l) for l in lines}) >= 1000
    for moves in lines:
        assert sorted(moves, key=_order_key) == moves
        for m in moves:  # every name parses and round-trips through move_index
            assert O.ptn(oracle, 5, O.from_ptn(oracle, 5, m)) == m


def test_generated_moves_follow_that_order(oracle):
    rng = np.random.default_rng(7)
    for n in (3, 4, 5, 6):
        for game in range(6):
            s = O.state_default(oracle, n, 4)
            for ply in range(60):
                if oracle.tzo_terminal(C.byref(s)) != -1:
                    break
                mv = O.possible_moves(oracle, s)
                names = [O.ptn(oracle, n, m) for m in mv]
                assert sorted(names, key=_order_key) == names
                assert len(set(mv)) == len(mv) > 0
                s = O.play(oracle, s, mv[int(rng.integers(len(mv)))])


def test_tps_round_trip_random_playouts(oracle):
    # target.rs:314-377 (text round trip over random 5x5 komi-2 playouts)
    rng = np.random.default_rng(123)
    for game in range(10):
        s = O.state_default(oracle, 5, 4)
        while oracle.tzo_terminal(C.byref(s)) == -1:
            tps = O.to_tps(oracle, s)
            s2 = O.state_from_tps(oracle, tps, 5, 4)
            s.reversible_plies = 0  # not representable in TPS (target.rs:322-326)
            assert bytes(s) == bytes(s2), tps
            mv = O.possible_moves(oracle, s)
            s = O.play(oracle, s, mv[int(rng.integers(len(mv)))])


def test_opening_rule_and_results(oracle):
    s = O.state_default(oracle, 5, 4)
    assert len(O.possible_moves(oracle, s)) == 25  # only flats on the first two plies
    s = O.play(oracle, s, O.from_ptn(oracle, 5, "a1"))
    assert s.top[0] == 1 and (s.colors[0] & 1) == 1  # white placed a black flat
    assert s.stones[1] == 20 and s.stones[0] == 21
    s = O.play(oracle, s, O.from_ptn(oracle, 5, "e5"))
    assert (s.colors[24] & 1) == 0 and s.stones[0] == 20
    # a road for white along rank 1 on 3x3: result from the side to move (black) is Loss
    s = O.state_from_tps(oracle, "x3/x3/1,1,x 1 3", 3, 0)
    s = O.play(oracle, s, O.from_ptn(oracle, 3, "c1"))
    assert oracle.tzo_result(C.byref(s)) == 1
    assert oracle.tzo_terminal(C.byref(s)) == 1
    # flat win on a full board with komi: 4 white flats vs 4 black flats + komi 2 -> black
    s = O.state_from_tps(oracle, "1,2,1/2,1S,2/1,2,x 1 5", 3, 4)
    s = O.play(oracle, s, O.from_ptn(oracle, 3, "c1"))
    assert oracle.tzo_result(C.byref(s)) == 2
    s = O.state_from_tps(oracle, "1,2,1/2,1S,2/1,2,x 1 5", 3, 0)
    s = O.play(oracle, s, O.from_ptn(oracle, 3, "c1"))
    assert oracle.tzo_result(C.byref(s)) == 3  # 4 v 4 without komi -> draw
