"""Target augmentation (takzero/src/target.rs:32-54): one of the 8 board symmetries, drawn uniformly, applied to
the position and to every move of the policy.  The draw is uniform, so the order in which fast-tak lists the
symmetries does not matter here."""
import numpy as np

from . import api

_DIRS = ((0, 1), (1, 0), (0, -1), (-1, 0))  # move_index order: Up, Right, Down, Left (repr.rs:49-71)
_tables = {}


def _apply(n, sym, x, y):
    if sym & 1:
        x = n - 1 - x
    for _ in range((sym >> 1) & 3):
        x, y = n - 1 - y, x
    return x, y


def tables(n):
    """(square permutation [8][n*n], direction map [8][4])."""
    if n not in _tables:
        perm = np.zeros((8, n * n), np.int64)
        dirs = np.zeros((8, 4), np.int64)
        for sym in range(8):
            for y in range(n):
                for x in range(n):
                    ox, oy = _apply(n, sym, x, y)
                    perm[sym, y * n + x] = oy * n + ox
            for d, (dx, dy) in enumerate(_DIRS):
                # the symmetry is affine: the image of a unit step is the difference of the images of its end points
                sx, sy = (0 if dx >= 0 else n - 1), (0 if dy >= 0 else n - 1)
                a, b = _apply(n, sym, sx, sy), _apply(n, sym, sx + dx, sy + dy)
                dirs[sym, d] = _DIRS.index((b[0] - a[0], b[1] - a[1]))
        _tables[n] = (perm, dirs)
    return _tables[n]


def augment_state(state, sym, n):
    perm = tables(n)[0][sym]
    out = state.copy()
    nn = n * n
    for field in ("colors", "height", "top"):
        old = state[field]
        new = np.zeros_like(old)
        new[perm] = old[:nn]
        out[field] = new
    return out


def augment_moves(moves, sym, n):
    perm, dirs = tables(n)
    nn, patterns = n * n, (1 << n) - 2
    idx = np.asarray(moves, np.int64)
    ch, sq = idx // nn, idx % nn
    spread = ch >= 3
    d = np.where(spread, (ch - 3) // patterns, 0)
    pat = np.where(spread, (ch - 3) % patterns, 0)
    ch2 = np.where(spread, 3 + pat + patterns * dirs[sym][d], ch)
    return (ch2 * nn + perm[sym][sq]).astype(np.uint16)


def augment_target(target, rng, n):
    """Augment::augment for one (state, moves, policy, value, ube) target."""
    st, moves, pol, value, ube = target
    sym = int(rng.integers(0, 8))
    return augment_state(st, sym, n), augment_moves(moves, sym, n), pol, value, ube


def augment_batch(states, moves, rows, rng, n):
    """Augment::augment for a whole batch at once: states [B] records, moves = all policies' move indices
    concatenated, rows[i] = the target move i belongs to.  One symmetry per target, drawn uniformly."""
    perm, dirs = tables(n)
    B, nn, patterns = len(states), n * n, (1 << n) - 2
    sym = rng.integers(0, 8, B)
    out = states.copy()
    target_sq = perm[sym]                       # [B][nn]: where each square goes
    b_index = np.arange(B)[:, None]
    for field in ("colors", "height", "top"):
        old = states[field]
        new = np.zeros_like(old)
        new[b_index, target_sq] = old[:, :nn]
        out[field] = new
    idx = np.asarray(moves, np.int64)
    msym = sym[rows]
    ch, sq = idx // nn, idx % nn
    spread = ch >= 3
    d = np.where(spread, (ch - 3) // patterns, 0)
    pat = np.where(spread, (ch - 3) % patterns, 0)
    ch2 = np.where(spread, 3 + pat + patterns * dirs[msym, d], ch)
    return out, ch2 * nn + perm[msym, sq]
