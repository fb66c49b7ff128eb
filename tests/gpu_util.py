import os

import numpy as np
import pytest


def require_gpu():
    import takzero_amd.api as A

    if A._lib.load().tz_device_count() == 0:
        pytest.fail("test marked gpu but no HIP device is visible (no CPU fallback exists)")
    return A


def random_positions(oracle, O, n, half_komi, count, seed, min_ply=0, max_ply=40):
    """Positions from random playouts of the oracle rules (non-terminal)."""
    import ctypes as C

    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        s = O.state_default(oracle, n, half_komi)
        target = int(rng.integers(min_ply, max_ply + 1))
        ok = True
        for _ in range(target):
            if oracle.tzo_terminal(C.byref(s)) != -1:
                ok = False
                break
            mv = O.possible_moves(oracle, s)
            s = O.play(oracle, s, mv[int(rng.integers(len(mv)))])
        if ok and oracle.tzo_terminal(C.byref(s)) == -1:
            out.append(s)
    return out
