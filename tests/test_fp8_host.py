"""Host-side conversion of the TZ_PREC_F16C8 weights to OCP FP8 E4M3 (takzero_amd/csrc/tz_fp8.h) against torch.float8_e4m3fn:
every code point, every midpoint between neighbouring codes and its two float neighbours (the rounding ties), a million random
values over eight decades, and saturation (torch turns overflow into NaN, the library saturates at 448: the weights are scaled
below 256 anyway, see build_layer)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_e4m3_matches_torch(tmp_path):
    torch = pytest.importorskip("torch")
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("this torch has no float8_e4m3fn")
    exe = str(tmp_path / "fp8_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "takzero_amd", "csrc"), os.path.join(ROOT, "tests", "fp8_harness.cpp"),
                    "-o", exe], check=True)
    rng = np.random.default_rng(0)
    codes = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    codes = np.sort(codes[np.isfinite(codes)])
    mids = ((codes[:-1].astype(np.float64) + codes[1:].astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(s) for s in (1e-4, 1e-3, 0.05, 1, 30, 300)] +
                       [codes, mids, np.nextafter(mids, np.float32(1e9)), np.nextafter(mids, np.float32(-1e9)),
                        np.array([0.0, -0.0, 448, 449, 463.9, 464, 465, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 2.0 ** -6, 0.0175], np.float32)]).astype(np.float32)
    out = subprocess.run([exe], input=x.tobytes(), capture_output=True, check=True).stdout
    got = np.frombuffer(out, np.uint8)
    assert len(got) == len(x)
    # torch rounds |x| < 464 to <= 448 and turns the rest into NaN; the library saturates
    want = torch.from_numpy(np.where(np.abs(x) >= 464, np.sign(x) * np.float32(448), x).astype(np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, [(float(x[i]), int(got[i]), int(want[i])) for i in bad[:5]]
    # and a NaN stays a NaN
    nan = subprocess.run([exe], input=np.array([np.nan], np.float32).tobytes(), capture_output=True, check=True).stdout
    assert nan[0] & 0x7f == 0x7f
