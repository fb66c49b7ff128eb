"""Builds takzero_amd/libtakzero_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtakzero_hip.so")
ARCH = "gfx950"
# (source, extra flags).  The tree kernels and the ABI glue do reference-order f32 arithmetic:
# no FMA contraction there.  The network kernels keep the default (contraction on).
UNITS = [
    ("tz_text.cpp", ["-ffp-contract=off"]),
    ("tz_ot.cpp", []),
    ("tz_host.cpp", ["-ffp-contract=off"]),
    ("tz_host_learn.cpp", ["-ffp-contract=off"]),
    ("tz_comm.cpp", []),
    ("tz_tree.hip", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]),
    ("tz_capi.hip", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]),
    ("tz_nn.hip", []),
    ("tz_nn_split.hip", []),          # includes tz_nn.hip with TZ_NN_SPLIT_TU: the split-precision kernels, compiled in parallel
    ("tz_nn_c6.hip", []),             # the same for TZ_PREC_F16C6 (FP6 block-scaled corrections): its own kernel on tz_nn.hip's templates,
    ("tz_nn_c6b.hip", []),            #   its workgroup forms spread over three units (5x5 full size / 5x5 small batches / 6x6)
    ("tz_nn_c6c.hip", []),
    ("tz_learn.hip", ["-ffp-contract=off"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-fast-math", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, ablations=None):
    """ablations (or TZ_BUILD_ABLATIONS=1): also compile the A/B twins and ablation variants of the network kernels that
    tz_debug_conv_bench / tz_debug_tower_bench time (tools/tower_bench.py, tools/conv_bench.py); the shipped object has none."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if ablations is None:
        ablations = os.environ.get("TZ_BUILD_ABLATIONS", "0") not in ("", "0")
    stamp = os.path.join(CSRC, ".ablations")
    if ablations != os.path.exists(stamp):        # switching the flavour rebuilds the network unit
        force_nn = True
        if ablations:
            open(stamp, "w").close()
        elif os.path.exists(stamp):
            os.remove(stamp)
    else:
        force_nn = False
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "takzero_hip.h"))
    objs = []
    procs = []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        nn = src in ("tz_nn.hip", "tz_nn_split.hip", "tz_nn_c6.hip", "tz_nn_c6b.hip", "tz_nn_c6c.hip")
        deps = [s] + headers + ([os.path.join(CSRC, "tz_nn.hip")] if nn else []) + ([os.path.join(CSRC, "tz_nn_c6.hip")] if src.startswith("tz_nn_c6") else [])
        if force or _stale(o, deps) or (force_nn and nn):
            cmd = [hipcc] + COMMON + extra + (["-DTZ_ABLATIONS"] if ablations and nn else []) + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("---- %s ----\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", OUT] + objs + ["-ldl"]
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablations=True if "--ablations" in sys.argv else None))
