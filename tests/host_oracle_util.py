"""Builds and runs tests/host_over_oracle.cpp: the native host drivers (csrc/tz_host.cpp) bound to the CPU oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out_dir, sanitize, with_net=False):
    exe = os.path.join(str(out_dir), "host_over_oracle" + ("_san" if sanitize else "") + ("_net" if with_net else ""))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "takzero_amd", "csrc"),
           os.path.join(ROOT, "tests", "host_over_oracle.cpp"), os.path.join(ROOT, "oracle", "capi.cpp"),
           os.path.join(ROOT, "takzero_amd", "csrc", "tz_text.cpp"), os.path.join(ROOT, "takzero_amd", "csrc", "tz_comm.cpp"),
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lpthread", "-o", exe]
    if sanitize:
        cmd[4:4] = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    if with_net:   # the oracle search's Agent = the HIP network through tz_net_eval; tz_text.cpp then comes from the library
        lib = os.path.join(ROOT, "takzero_amd")
        cmd.remove(os.path.join(ROOT, "takzero_amd", "csrc", "tz_text.cpp"))
        cmd.remove(os.path.join(ROOT, "takzero_amd", "csrc", "tz_comm.cpp"))
        cmd[4:4] = ["-DTZ_HARNESS_WITH_NET"]
        cmd += ["-L" + lib, "-ltakzero_hip", "-Wl,-rpath," + lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the harness here: " + r.stderr[-400:])
    return exe


def command(exe, prefix, n, half_komi, agent, batch, kind, sims, k, exploration, moves, seed, net_args=()):
    return [exe] + [str(x) for x in (n, half_komi, agent, batch, kind, sims, k, exploration, moves, seed)] + [str(prefix)] + [str(x) for x in net_args]


def run(exe, prefix, n, half_komi, agent, batch, kind, sims, k, exploration, moves, seed, net_args=(), env=None, parts=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", **(env or {}))
    r = subprocess.run(command(exe, prefix, n, half_komi, agent, batch, kind, sims, k, exploration, moves, seed, net_args),
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-2000:])
    out = {}
    for part in parts or ("targets", "replays", "exploration", "reanalyze", "consumers"):
        with open("%s.%s" % (prefix, part), "rb") as f:
            out[part] = f.read()
    fields = dict(zip(r.stdout.split()[::2], r.stdout.split()[1::2]))
    out["positions"] = int(fields["positions"])
    for key in ("host_checks", "host_mismatches", "sampled_games", "proven_selected_children"):
        out[key] = int(fields.get(key, -1))
    return out
