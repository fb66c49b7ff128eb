"""The reference's model files.  `learn` writes `model_latest.ot` / `model_NNNNNNN.ot` with tch's VarStore::save
(takzero/src/network/mod.rs:16-18): a LibTorch `torch::serialize::OutputArchive` holding one named tensor per variable,
names = VarStore paths joined with '.'.

The library reads and writes that format natively (csrc/tz_ot.cpp: zip + the pickle subset LibTorch emits), so
`load_ot` / `save_ot` here go through `tz_weights_convert` and need neither torch nor a compiler.  The LibTorch-based
functions (`read_ot_libtorch` = torch.jit.load, `save_ot_libtorch` = the small C++ program over OutputArchive) are kept
as the genuine reader / writer the tests cross-check the native code against.

Name quirk (residual.rs:50-55): both SmallBlocks of a ResidualBlock are created under the same
path, so the second one's five variables collide with the first one's and tch renames them
`<path>__<number of variables registered so far>`.  Readers map   <name> -> `.a.`   and   <name>__K -> `.b.`   whatever K
is, which gives the names takzero_amd.weights / tz_net_load_weights use."""
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WRITER = os.path.join(HERE, "ot_writer")

_SUFFIX = re.compile(r"^(.*)__(\d+)$")
_BLOCK = re.compile(r"^(core\.res_block_\d+)\.(.+)$")


def read_ot_libtorch(path):
    """name -> fp32 ndarray for every tensor of a LibTorch archive, read by LibTorch itself (torch.jit.load): the
    cross-check of the native reader."""
    import torch

    m = torch.jit.load(str(path), map_location="cpu")
    out = {}
    for k, v in list(m.named_parameters()) + list(m.named_buffers()):
        out[k] = v.detach().to(torch.float32).numpy().copy()
    return out


def canonical_names(named):
    """tch VarStore names -> the `.a.` / `.b.` spelling of takzero_amd.weights."""
    out = {}
    for name, arr in named.items():
        m = _SUFFIX.match(name)
        base, dup = (m.group(1), True) if m else (name, False)
        b = _BLOCK.match(base)
        if b:
            out["%s.%s.%s" % (b.group(1), "b" if dup else "a", b.group(2))] = arr
        elif dup:
            raise ValueError("unexpected duplicated variable outside a residual block: %s" % name)
        else:
            out[base] = arr
    return out


def load_ot(path):
    """Canonical (`.a.` / `.b.`) name -> fp32 array, read by the library's own archive reader (tz_weights_convert)."""
    from . import _lib
    from .weights import load_tzw

    with tempfile.TemporaryDirectory() as tmp:
        flat = os.path.join(tmp, "model.tzw")
        _lib.check(_lib.load().tz_weights_convert(str(path).encode(), flat.encode()))
        return load_tzw(flat)


def load_ot_libtorch(path):
    return canonical_names(read_ot_libtorch(path))


def tch_names(tensors):
    """The inverse (what tch would name the variables when it builds the net: creation order of net5.rs /
    net6_simhash.rs, duplicates suffixed with the number of variables registered so far).  Used to write test
    archives; the reader above does not depend on the exact numbers."""
    out, count = [], 0
    seen = set()

    def add(name, key):
        nonlocal count
        final = name if name not in seen else "%s__%d" % (name, count)
        seen.add(final)
        out.append((final, tensors[key]))
        count += 1

    # tch nn::batch_norm creates the affine pair before the running statistics; TZ_TCH_BN_ORDER=stats_first gives the older
    # order (same switch as csrc/tz_ot.cpp ot_tch_names; not pinned by any file of the reference)
    order = (("running_mean", "running_var", "weight", "bias") if os.environ.get("TZ_TCH_BN_ORDER") == "stats_first"
             else ("weight", "bias", "running_mean", "running_var"))

    def bn(path, key):
        for v in order:
            add("%s.%s" % (path, v), "%s.%s" % (key, v))

    add("core.input_conv2d.weight", "core.input_conv2d.weight")
    bn("core.batch_norm", "core.batch_norm")
    b = 0
    while "core.res_block_%d.a.conv2d.weight" % b in tensors:
        for half in "ab":
            add("core.res_block_%d.conv2d.weight" % b, "core.res_block_%d.%s.conv2d.weight" % (b, half))
            bn("core.res_block_%d.batch_norm" % b, "core.res_block_%d.%s.batch_norm" % (b, half))
        b += 1
    rest = [k for k in tensors if not k.startswith("core.")]
    for k in rest:
        add(k, k)
    return out


def build_writer(force=False):
    """g++ takzero_amd/csrc/ot_writer.cpp against the torch wheel's LibTorch -> takzero_amd/ot_writer."""
    src = os.path.join(HERE, "csrc", "ot_writer.cpp")
    if not force and os.path.exists(WRITER) and os.path.getmtime(WRITER) >= os.path.getmtime(src):
        return WRITER
    import torch

    tdir = os.path.dirname(torch.__file__)
    cmd = ["g++", "-std=c++17", "-O1", src, "-o", WRITER, "-I" + os.path.join(tdir, "include"),
           "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-L" + os.path.join(tdir, "lib"),
           "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cannot build the LibTorch archive writer: " + r.stderr[-500:])
    return WRITER


def save_ot(path, tensors):
    """Network::save (network/mod.rs:16-18): `tensors` by canonical (`.a.` / `.b.`) name -> LibTorch archive with tch's
    variable names, written by the library's own writer to `<path>.part` and renamed (a reader never sees half a model)."""
    from . import _lib
    from .weights import save_tzw

    with tempfile.TemporaryDirectory() as tmp:
        flat = os.path.join(tmp, "model.tzw")
        save_tzw(flat, tensors)
        _lib.check(_lib.load().tz_weights_convert(flat.encode(), str(path).encode()))
    return str(path)


def save_ot_libtorch(path, tensors):
    """The same through LibTorch's own OutputArchive (the small C++ program below): cross-check of the native writer."""
    exe = build_writer()
    path = str(path)
    with tempfile.TemporaryDirectory() as tmp:
        manifest = os.path.join(tmp, "manifest.txt")
        with open(manifest, "w") as mf:
            for i, (name, arr) in enumerate(tch_names(tensors)):
                raw = os.path.join(tmp, "t%d.bin" % i)
                a = np.ascontiguousarray(arr, np.float32)
                a.tofile(raw)
                mf.write("%s %d %s %s\n" % (name, a.ndim, " ".join(str(d) for d in a.shape), raw))
        part = path + ".part"
        # the archive writer is a background helper: one thread, low priority, so that it never competes with the
        # threads that feed the GPU
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        cmd = [exe, manifest, part]
        if os.path.exists("/usr/bin/nice"):
            cmd = ["/usr/bin/nice", "-n", "10"] + cmd
        subprocess.check_call(cmd, env=env)
        os.replace(part, path)
    return path
