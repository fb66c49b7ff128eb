"""Target augmentation (target.rs:32-54) against the oracle's rules: the legal moves of a mirrored / rotated position
are the mirrored / rotated legal moves, and playing corresponding moves gives corresponding positions."""
import ctypes as C

import numpy as np

import oracle_lib as O
from gpu_util import random_positions


def test_symmetries_commute_with_the_rules():
    from takzero_amd import augment as AU

    oracle = O.load()
    for n in (4, 5, 6):
        perm, dirs = AU.tables(n)
        assert sorted(map(tuple, perm)) == sorted(set(map(tuple, perm))) and len(set(map(tuple, perm))) == 8
        for sym in range(8):
            assert sorted(perm[sym]) == list(range(n * n)) and sorted(dirs[sym]) == [0, 1, 2, 3]
        states = random_positions(oracle, O, n, 4, 6, 40 + n, min_ply=6, max_ply=30)
        arr = O.states_array(states)
        for i, s in enumerate(states):
            moves = np.array(O.possible_moves(oracle, s), np.int64)
            for sym in range(8):
                a = AU.augment_state(arr[i], sym, n)
                sa = O.TzState.from_buffer_copy(a.tobytes())
                am = AU.augment_moves(moves, sym, n)
                assert sorted(O.possible_moves(oracle, sa)) == sorted(int(m) for m in am), (n, i, sym)
                for j in (0, len(moves) // 2, len(moves) - 1):
                    after = O.states_array([O.play(oracle, s, int(moves[j]))])[0]
                    want = AU.augment_state(after, sym, n)
                    got = O.states_array([O.play(oracle, sa, int(am[j]))])[0]
                    assert O.to_tps(oracle, O.TzState.from_buffer_copy(got.tobytes())) == \
                        O.to_tps(oracle, O.TzState.from_buffer_copy(want.tobytes())), (n, i, sym, j)


def test_augment_target_keeps_probabilities_with_their_moves():
    from takzero_amd import augment as AU

    oracle = O.load()
    n = 5
    s = random_positions(oracle, O, n, 4, 1, 3, min_ply=8, max_ply=20)[0]
    moves = np.array(O.possible_moves(oracle, s), np.uint16)
    pol = np.linspace(0, 1, len(moves)).astype(np.float32)
    seen = set()
    rng = np.random.default_rng(0)
    for _ in range(64):
        st, mv, p, v, u = AU.augment_target((O.states_array([s])[0], moves, pol, 0.5, 1.0), rng, n)
        seen.add(tuple(int(m) for m in mv))
        assert np.array_equal(p, pol) and (v, u) == (0.5, 1.0) and len(set(mv.tolist())) == len(moves)
    assert len(seen) >= 4  # several different symmetries were drawn


def test_batch_augmentation_equals_the_per_target_one():
    from takzero_amd import augment as AU

    oracle = O.load()
    n, B = 5, 24
    states = O.states_array(random_positions(oracle, O, n, 4, B, 5, min_ply=4, max_ply=30))
    per = [np.array(O.possible_moves(oracle, O.TzState.from_buffer_copy(states[i].tobytes())), np.int64) for i in range(B)]
    rows = np.repeat(np.arange(B), [len(m) for m in per])

    class Fixed:  # hands out a chosen symmetry per target
        def __init__(self, sym):
            self.sym = sym

        def integers(self, lo, hi, size):
            return self.sym

    sym = np.arange(B) % 8
    out_states, out_moves = AU.augment_batch(states, np.concatenate(per), rows, Fixed(sym), n)
    off = 0
    for i in range(B):
        want_state = AU.augment_state(states[i], int(sym[i]), n)
        for f in ("colors", "height", "top", "stones", "caps", "to_move", "ply"):
            assert np.array_equal(out_states[i][f], want_state[f]), (i, f)
        assert np.array_equal(out_moves[off:off + len(per[i])], AU.augment_moves(per[i], int(sym[i]), n).astype(np.int64))
        off += len(per[i])
