"""tz_math.h (shared by the device tree kernels and the oracle) against the host libm that the
Rust reference would call through f32::exp / f32::ln (policy.rs:10-19,140-156; noise.rs:23)."""
import ctypes as C
import math

import numpy as np


def test_expf_logf_agree_with_libm(oracle):
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    libm.logf.restype = C.c_float
    libm.logf.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    xs = np.concatenate([-rng.random(20000, dtype=np.float32) * 30, rng.random(2000, dtype=np.float32) * 80,
                         np.array([0.0, -0.0, -1e-30, -87.0, -103.0, 88.0], np.float32)])
    bad = 0
    for x in xs:
        a, b = oracle.tzo_expf(float(x)), libm.expf(float(x))
        if a != b:
            bad += 1
            assert abs(np.float32(a).view(np.int32).astype(np.int64) - np.float32(b).view(np.int32)) <= 1, x
    assert bad <= len(xs) // 1000, bad  # both round a double evaluation; disagreements are 1-ulp ties
    ys = np.concatenate([rng.random(20000, dtype=np.float32), rng.random(2000, dtype=np.float32) * 1e6 + 1,
                         (np.arange(1, 5000, dtype=np.float32) + 501) / 500, np.array([1.0, 1e-38, 3e38], np.float32)])
    bad = 0
    for y in ys:
        if y <= 0:
            continue
        a, b = oracle.tzo_logf(float(y)), libm.logf(float(y))
        if a != b:
            bad += 1
            assert abs(np.float32(a).view(np.int32).astype(np.int64) - np.float32(b).view(np.int32)) <= 1, y
    # glibc's logf documents 0.818 ulp, so 1-ulp disagreements are expected there; tz_logf itself must
    # be the correctly rounded value (checked against float64 below)
    assert bad <= len(ys) // 50, bad
    exact = np.log(ys.astype(np.float64)).astype(np.float32)
    mine = np.array([oracle.tzo_logf(float(y)) for y in ys], np.float32)
    assert np.count_nonzero(mine != exact) <= 2
    exact = np.exp(xs.astype(np.float64)).astype(np.float32)
    mine = np.array([oracle.tzo_expf(float(x)) for x in xs], np.float32)
    assert np.count_nonzero(mine != exact) <= 2
    assert oracle.tzo_expf(0.0) == 1.0 and oracle.tzo_logf(1.0) == 0.0
    assert oracle.tzo_expf(-200.0) == 0.0 and math.isinf(oracle.tzo_logf(0.0))


def test_exploration_rate_matches_formula(oracle):
    # policy.rs:143-145: ln((1 + N + 500) / 500) + 4
    for n in (1, 2, 10, 400, 801, 5000):
        want = np.float32(np.log(np.float64(np.float32(np.float32(np.float32(1.0) + np.float32(n)) + np.float32(500.0)) / np.float32(500.0)))) + np.float32(4.0)
        assert abs(oracle.tzo_exploration_rate(float(n)) - want) <= 5e-7
