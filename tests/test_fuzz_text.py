"""AddressSanitizer / UBSan run of the native target-line parser and formatter (CPU build of takzero_amd/csrc/tz_text.cpp
only — GPU sanitizers are not available on the pool): 20 000 mutated inputs, no over-reads, no crashes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_text_entry_points_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_text")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "takzero_amd", "csrc"), os.path.join(ROOT, "tests", "fuzz_text.cpp"),
           os.path.join(ROOT, "takzero_amd", "csrc", "tz_text.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the sanitizer harness here: " + r.stderr[-300:])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok parsed="), (r.stdout[-300:], r.stderr[-1500:])
