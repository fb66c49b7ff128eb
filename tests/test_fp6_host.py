"""Host-side conversion of the TZ_PREC_F16C6 weights to OCP MX FP6 E2M3 (takzero_amd/csrc/tz_fp6.h) against a numpy restatement of
the format (1 sign, 2 exponent, 3 mantissa bits, bias 1, no infinities / NaN): every code point, every midpoint between
neighbouring codes and its two float neighbours (round to nearest even), saturation at 7.5, random values; the block scale rule
(the power of two s with amax / s in [3.75, 7.5)); and the 24-byte packing (element i in bits 6 i ..).  The hardware side of the
same facts - what v_cvt_scalef32_pk32_fp6_f16 emits and what v_mfma_scale_f32_16x16x128_f8f6f4 reads - is tools/mfma_f6_probe.hip
(profiles/r03_mfma_f6_probe.txt)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def e2m3_values():
    v = []
    for c in range(32):
        e, m = c >> 3, c & 7
        v.append(m / 8.0 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 1))
    return np.array(v, np.float64)   # ascending: 0 .. 7.5


def ref_code(x):
    """round to nearest even onto the grid, saturating; ties between codes c and c + 1 go to the even code"""
    vals = e2m3_values()
    a = abs(float(x))
    if a != a:
        c = 31
    elif a >= 7.75:
        c = 31
    else:
        i = int(np.searchsorted(vals, a, side="right")) - 1
        i = max(0, min(30, i))
        lo, hi = vals[i], vals[i + 1]
        if a - lo < hi - a:
            c = i
        elif a - lo > hi - a:
            c = i + 1
        else:
            c = i if i % 2 == 0 else i + 1
    sign = 32 if np.signbit(np.float32(x)) else 0
    return sign | c


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fp6") / "fp6_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "takzero_amd", "csrc"), os.path.join(ROOT, "tests", "fp6_harness.cpp"),
                    "-o", exe], check=True)
    return exe


def test_host_e2m3_codes(harness):
    rng = np.random.default_rng(0)
    vals = e2m3_values()
    mids = ((vals[:-1] + vals[1:]) / 2).astype(np.float32)
    x = np.concatenate([vals.astype(np.float32), -vals.astype(np.float32), mids, -mids, np.nextafter(mids, np.float32(100)),
                        np.nextafter(mids, np.float32(-100)), rng.uniform(-9, 9, 100000).astype(np.float32),
                        (rng.standard_normal(100000) * 0.3).astype(np.float32),
                        np.array([7.5, 7.74, 7.75, 7.76, 8, 100, 1e30, -1e30, 1e-30, -0.0, 0.0624, 0.0625, 0.0626], np.float32)]).astype(np.float32)
    out = subprocess.run([harness, "code"], input=x.tobytes(), capture_output=True, check=True).stdout
    got = np.frombuffer(out, np.uint8)
    want = np.array([ref_code(v) for v in x], np.uint8)
    # -0.0 and values that round to zero keep their sign bit in the code: both spell zero
    same = (got == want) | (((got & 31) == 0) & ((want & 31) == 0))
    bad = np.nonzero(~same)[0]
    assert len(bad) == 0, [(float(x[i]), int(got[i]), int(want[i])) for i in bad[:5]]


def test_block_scale_rule(harness):
    rng = np.random.default_rng(1)
    amax = np.concatenate([np.exp(rng.uniform(-40, 40, 20000)), [1.0, 7.5, 7.5 * 16 / 15, 3.75, 2.0 ** -20, 65504.0]]).astype(np.float32)
    out = subprocess.run([harness, "scale"], input=amax.tobytes(), capture_output=True, check=True).stdout
    b = np.frombuffer(out, np.uint8).astype(np.int64)
    s = np.ldexp(1.0, b - 127)
    r = amax.astype(np.float64) / s
    assert (r <= 7.5 * (1 + 1e-6)).all() and (r >= 3.75 * (1 - 1e-6)).all(), (r.min(), r.max())
    zero = subprocess.run([harness, "scale"], input=np.zeros(1, np.float32).tobytes(), capture_output=True, check=True).stdout
    assert zero[0] == 1        # a block of zeros: the smallest scale, never byte 0 or 255


def test_pack_32_codes_into_24_bytes(harness):
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 64, (500, 32)).astype(np.uint8)
    out = subprocess.run([harness, "pack"], input=codes.tobytes(), capture_output=True, check=True).stdout
    words = np.frombuffer(out, np.uint32).reshape(500, 6)
    for rec in range(500):
        big = 0
        for w in range(6):
            big |= int(words[rec, w]) << (32 * w)
        for i in range(32):
            assert (big >> (6 * i)) & 63 == codes[rec, i]
