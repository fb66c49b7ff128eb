"""The reference's `.ot` model files (tch VarStore::save = LibTorch OutputArchive, network/mod.rs:16-28).

The library reads and writes them natively (csrc/tz_ot.cpp).  tch / LibTorch are not under /root/reference, so the format is
pinned against the LibTorch of this image, both ways: archives written by LibTorch's own OutputArchive (the small C++
program takzero_amd/csrc/ot_writer.cpp, i.e. what torch-sys' at_save_multi calls) are read by the native reader, archives
written by the native writer are read by LibTorch's own reader (torch.jit.load), and the pickle program + code file of the
two writers are compared byte for byte."""
import os
import zipfile

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ot_writer():
    from takzero_amd import ot

    try:
        return ot.build_writer()
    except RuntimeError as e:
        pytest.skip(str(e))


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert np.asarray(b[k]).shape == np.asarray(a[k]).shape and np.array_equal(a[k], b[k]), k


def test_native_reader_reads_what_libtorch_writes(ot_writer, tmp_path):
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=3, blocks=2, seed=11, trained_stats=True)
    names = [n for n, _ in ot.tch_names(w)]
    # the second SmallBlock of every ResidualBlock collides with the first one's path and gets `__K`
    assert "core.res_block_0.conv2d.weight" in names and any(n.startswith("core.res_block_0.conv2d.weight__") for n in names)
    assert len(set(names)) == len(names) == len(w)
    path = ot.save_ot_libtorch(tmp_path / "model_latest.ot", w)
    assert not (tmp_path / "model_latest.ot.part").exists()
    _same(w, ot.load_ot(path))            # native reader
    _same(w, ot.load_ot_libtorch(path))   # LibTorch's own reader agrees


def test_libtorch_reads_what_the_native_writer_writes(tmp_path):
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=4, blocks=2, seed=12, trained_stats=True)
    path = ot.save_ot(tmp_path / "model_0000100.ot", w)
    assert not (tmp_path / "model_0000100.ot.part").exists()
    named = ot.read_ot_libtorch(path)     # torch.jit.load: the reader tch's VarStore::load goes through
    assert list(named) == [n for n, _ in ot.tch_names(w)]      # tch's variable names, in creation order
    _same(w, ot.canonical_names(named))
    _same(w, ot.load_ot(path))


def test_native_writer_emits_libtorchs_bytes(ot_writer, tmp_path):
    """Same tensors through OutputArchive::save_to and through the native writer: identical data.pkl (opcodes, memo layout,
    integer widths), identical code/__torch__.py and constants, identical storages."""
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=4, blocks=1, seed=3)     # 4-D, 2-D, 1-D and [1] shapes
    big = W.init_weights(W.ARCH_NET5, seed=3)
    w.update({k: big[k] for k in ("rnd_learning.final_linear.weight", "rnd_learning.final_linear.bias", "min", "max")})
    w["simhash_matrix"] = W.init_weights(W.ARCH_NET4_SIMHASH, seed=3)["simhash_matrix"]
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    a = ot.save_ot_libtorch(tmp_path / "a" / "model.ot", w)
    b = ot.save_ot(tmp_path / "b" / "model.ot", w)
    za, zb = zipfile.ZipFile(a), zipfile.ZipFile(b)

    def entry(z, suffix):
        (name,) = [n for n in z.namelist() if n.endswith(suffix)]
        return z.read(name)

    for suffix in ("/data.pkl", "/code/__torch__.py", "/constants.pkl", "/version"):
        assert entry(za, suffix) == entry(zb, suffix), suffix
    for i in range(len(w)):
        assert entry(za, "/data/%d" % i) == entry(zb, "/data/%d" % i)
    assert zb.testzip() is None            # CRCs of the native zip


def test_long_memo_indices_and_wide_integers(tmp_path):
    """More than 256 memo slots (LONG_BINPUT / LONG_BINGET) and element counts beyond 16 bits: a full net5 store."""
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_NET5, seed=4)
    assert w["min"].tolist() == [0.0] and w["max"].tolist() == [1.0]   # net5.rs:166-168
    # net5.rs:327-347 `update_rnd_persistance`: the RND normalisation survives save / load, as do all 28.8 M parameters
    w["min"] = np.float32([0.25])
    w["max"] = np.float32([7.5])
    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    back = ot.load_ot(path)
    assert back["min"].tolist() == [0.25] and back["max"].tolist() == [7.5]
    _same(w, back)
    named = ot.read_ot_libtorch(path)
    assert len(named) == len(w) and np.array_equal(named["max"], w["max"])
    assert np.array_equal(named["core.res_block_19.conv2d.weight__%d" % (5 + 19 * 10 + 5)], w["core.res_block_19.b.conv2d.weight"])


def test_reader_accepts_other_pickle_dialects_and_strided_tensors(tmp_path):
    """torch.jit.save of a Python-built module (named buffers, a transposed = non-contiguous parameter, fp64 and fp16
    storages) and torch.save of a plain state dict (protocol-2 pickle from Python's pickler)."""
    import torch

    from takzero_amd import _lib
    from takzero_amd.weights import load_tzw

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(1)
            self.w = torch.nn.Parameter(torch.randn(5, 7, generator=g).t())          # strides (1, 7)
            self.register_buffer("d", torch.randn(3, 2, generator=g, dtype=torch.float64))
            self.register_buffer("h", torch.randn(4, generator=g).to(torch.float16))
            self.register_buffer("v", torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)[:, 1:, ::2])

        def forward(self, x):
            return x

    m = M()
    torch.jit.script(m).save(str(tmp_path / "scripted.pt"))
    torch.save({k: v for k, v in m.state_dict().items()}, str(tmp_path / "state.pt"))
    want = {k: v.detach().to(torch.float32).numpy() for k, v in m.state_dict().items()}
    for src in ("scripted.pt", "state.pt"):
        _lib.check(_lib.load().tz_weights_convert(str(tmp_path / src).encode(), str(tmp_path / "out.tzw").encode()))
        got = load_tzw(tmp_path / "out.tzw")
        assert set(got) == set(want), src
        for k in want:
            assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), (src, k)


def test_damaged_archives_are_parse_errors(tmp_path):
    """A torn model file (learn writing while selfplay reads: selfplay/src/main.rs:112-119 keeps the old net) is TZ_EPARSE,
    never a crash: truncations at every structural boundary and a flipped byte in the pickle."""
    from takzero_amd import _lib, ot
    from takzero_amd import weights as W

    lib = _lib.load()
    w = W.init_weights(W.ARCH_TEST, n=3, blocks=1, seed=2)
    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    blob = open(path, "rb").read()
    out = str(tmp_path / "x.tzw").encode()
    for cut in (0, 3, 30, 100, len(blob) // 2, len(blob) - 300, len(blob) - 23, len(blob) - 1):
        bad = tmp_path / "cut.ot"
        bad.write_bytes(blob[:cut])
        assert lib.tz_weights_convert(str(bad).encode(), out) == -2, cut
    z = zipfile.ZipFile(path)
    info = [i for i in z.infolist() if i.filename.endswith("/data.pkl")][0]
    start = blob.index(b"\x80\x02c__torch__", info.header_offset)
    for at, byte in ((start, b"\x01"), (start + 2, b"\x7f")):     # an unknown opcode in place of PROTO / GLOBAL
        bad = tmp_path / "flip.ot"
        bad.write_bytes(blob[:at] + byte + blob[at + 1:])
        assert lib.tz_weights_convert(str(bad).encode(), out) == -2
    assert lib.tz_weights_convert(str(tmp_path / "missing.ot").encode(), out) == -2


def test_canonical_names_do_not_depend_on_the_suffix_number(tmp_path):
    from takzero_amd import ot

    a = np.zeros(1, np.float32)
    named = {"core.res_block_3.conv2d.weight": a, "core.res_block_3.conv2d.weight__999": a + 1,
             "core.res_block_3.batch_norm.bias": a, "core.res_block_3.batch_norm.bias__41": a + 1, "policy.conv2d.bias": a}
    c = ot.canonical_names(named)
    assert set(c) == {"core.res_block_3.a.conv2d.weight", "core.res_block_3.b.conv2d.weight",
                      "core.res_block_3.a.batch_norm.bias", "core.res_block_3.b.batch_norm.bias", "policy.conv2d.bias"}
    assert c["core.res_block_3.b.conv2d.weight"][0] == 1
    with pytest.raises(ValueError):
        ot.canonical_names({"policy.conv2d.bias__3": a})
    # the native reader applies the same rule: archives whose duplicates carry other numbers (the two creation orders of
    # nn::batch_norm, TZ_TCH_BN_ORDER) map to the same `.a.` / `.b.` names
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=3, blocks=1, seed=9, trained_stats=True)
    os.environ["TZ_TCH_BN_ORDER"] = "stats_first"
    try:
        p1 = ot.save_ot(tmp_path / "stats_first.ot", w)
        n1 = [n for n, _ in ot.tch_names(w)]
    finally:
        del os.environ["TZ_TCH_BN_ORDER"]
    p2 = ot.save_ot(tmp_path / "affine_first.ot", w)
    n2 = [n for n, _ in ot.tch_names(w)]
    assert n1 != n2 and "core.res_block_0.batch_norm.running_mean__11" in n1 and "core.res_block_0.batch_norm.weight__11" in n2
    assert list(ot.read_ot_libtorch(p1)) == n1 and list(ot.read_ot_libtorch(p2)) == n2
    _same(w, ot.load_ot(p1))
    _same(w, ot.load_ot(p2))


@pytest.mark.gpu
def test_net_load_ot_equals_load_tensors(tmp_path):
    """Network::load (network/mod.rs:24-28) on a LibTorch archive gives the same network as the flat container."""
    import takzero_amd.api as tz
    from takzero_amd import weights as W

    import oracle_lib as O
    from gpu_util import random_positions, require_gpu

    require_gpu()
    oracle = O.load()
    w = W.init_weights(W.ARCH_NET6_SIMHASH, blocks=2, seed=5, trained_stats=True)
    from takzero_amd import ot

    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    states = O.states_array(random_positions(oracle, O, 6, 4, 24, 3, max_ply=30))
    a = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load_tensors(w)
    b = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load(path)
    ra, rb = a.forward_raw(states), b.forward_raw(states)
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(a.hash_indices(states), b.hash_indices(states))


def test_native_reader_under_sanitizers_on_damaged_archives(tmp_path):
    """csrc/tz_ot.cpp compiled alone with -fsanitize=address,undefined and fed 400 damaged copies of a real archive: random byte
    flips in the zip records, the pickle program and the storages, truncations at random points, spliced garbage.  Every file must
    come back as loaded or TZ_EPARSE; the sanitizers must stay silent (a reader of files that another process is still writing)."""
    import subprocess

    from takzero_amd import ot
    from takzero_amd import weights as W

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ot_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "tests", "ot_fuzz_harness.cpp"), os.path.join(root, "takzero_amd", "csrc", "tz_ot.cpp"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the harness here: " + r.stderr[-400:])
    w = W.init_weights(W.ARCH_TEST, n=3, blocks=1, seed=2)
    w = {k: v for k, v in w.items() if v.size < 5000}          # keep the file small: the structure is what is attacked
    good = ot.save_ot(tmp_path / "good.ot", w)
    blob = bytearray(open(good, "rb").read())
    rng = np.random.default_rng(0)
    pkl_at = bytes(blob).index(b"\x80\x02c__torch__")
    eocd_at = bytes(blob).rindex(b"PK\x05\x06")
    cd_at = bytes(blob).index(b"PK\x01\x02")
    files = [good]
    for i in range(400):
        b = bytearray(blob)
        kind = i % 5
        if kind == 0:      # flips anywhere
            for at in rng.integers(0, len(b), int(rng.integers(1, 8))):
                b[at] = int(rng.integers(0, 256))
        elif kind == 1:    # flips in the pickle program
            for at in rng.integers(pkl_at, pkl_at + 1500, int(rng.integers(1, 6))):
                b[at] = int(rng.integers(0, 256))
        elif kind == 2:    # flips in the central directory / end records (sizes, offsets, counts)
            for at in rng.integers(cd_at, len(b), int(rng.integers(1, 6))):
                b[at] = int(rng.integers(0, 256))
        elif kind == 3:    # truncation
            b = b[:int(rng.integers(0, len(b)))]
        else:              # a spliced run of garbage
            at = int(rng.integers(0, len(b) - 64))
            b[at:at + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        path = tmp_path / ("bad%03d.ot" % i)
        path.write_bytes(bytes(b))
        files.append(str(path))
    # targeted: 64-bit sizes and offsets chosen so that a bounds check written as a sum wraps around - the zip64 records'
    # directory offset / size / count, a zip64 locator pointing near 2^64, an 8-byte string length in the pickle, huge tensor
    # dimensions in the flat container
    import struct

    count, cd_size, cd_off = struct.unpack("<HII", bytes(blob[eocd_at + 10:eocd_at + 20]))
    z64_at = eocd_at                                                   # a zip64 end record + locator in front of the end record,
    z64 = (b"PK\x06\x06" + struct.pack("<QHHIIQQQQ", 44, 45, 45, 0, 0, count, count, cd_size, cd_off) +   # as LibTorch writes them
           b"PK\x06\x07" + struct.pack("<IQI", 0, z64_at, 1))
    blob64 = bytearray(blob[:eocd_at] + z64 + blob[eocd_at:])
    loc_at = z64_at + 56
    targeted = [bytes(blob64)]                                          # (the intact zip64 form must still load)
    for off, val in ((48, 2 ** 64 - 10), (40, 2 ** 64 - 1), (32, 2 ** 63), (48, len(blob64) - 5), (40, 2 ** 40)):
        b = bytearray(blob64)
        b[z64_at + off:z64_at + off + 8] = struct.pack("<Q", val)      # count (32), directory size (40), directory offset (48)
        targeted.append(bytes(b))
    for val in (2 ** 64 - 56, 2 ** 64 - 1, len(blob64)):
        b = bytearray(blob64)
        b[loc_at + 8:loc_at + 16] = struct.pack("<Q", val)             # where the zip64 end record is said to be
        targeted.append(bytes(b))
    b = bytearray(blob)
    x_at = bytes(b).index(b"X", pkl_at)                                # a BINUNICODE: turn it into BINUNICODE8 with a length of 2^64 - 3
    b[x_at:x_at + 5] = b"\x8d" + struct.pack("<I", 0xFFFFFFFD)
    b[x_at + 5:x_at + 9] = b"\xff\xff\xff\xff"
    targeted.append(bytes(b))
    for dims in ((2 ** 31, 2 ** 31, 4), (2 ** 32 - 1, 2 ** 32 - 1), (2 ** 30, 16)):
        t = b"TZW1" + struct.pack("<I", 1) + struct.pack("<H", 1) + b"w" + bytes([len(dims)]) + b"".join(struct.pack("<I", d) for d in dims) + b"\0" * 64
        targeted.append(t)
    for i, data in enumerate(targeted):
        path = tmp_path / ("wrap%02d.ot" % i)
        path.write_bytes(data)
        files.append(str(path))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1")
    out = subprocess.run([exe] + files, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, (out.stdout[-300:], out.stderr[-3000:])
    fields = dict(zip(out.stdout.split()[::2], out.stdout.split()[1::2]))
    assert int(fields["ok"]) >= 2 and int(fields["other"]) == 0 and int(fields["ok"]) + int(fields["parse_errors"]) == len(files)
    assert int(fields["parse_errors"]) > 150 and eocd_at > cd_at
