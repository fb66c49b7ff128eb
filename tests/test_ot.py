"""The reference's `.ot` model files (tch VarStore::save = LibTorch OutputArchive): a real archive is written with
LibTorch's C++ API under tch's variable names (incl. the `__K` suffix of the second SmallBlock) and read back."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ot_writer(tmp_path_factory):
    import torch

    tdir = os.path.dirname(torch.__file__)
    exe = str(tmp_path_factory.mktemp("ot") / "tzw_to_ot")
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "tools", "tzw_to_ot.cpp"), "-o", exe,
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the LibTorch archive writer here: " + r.stderr[-300:])
    return exe


def test_ot_round_trip_through_libtorch(ot_writer, tmp_path):
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=3, blocks=2, seed=11, trained_stats=True)
    named = ot.tch_names(w)
    names = [n for n, _ in named]
    # the second SmallBlock of every ResidualBlock collides with the first one's path and gets `__K`
    assert "core.res_block_0.conv2d.weight" in names and any(n.startswith("core.res_block_0.conv2d.weight__") for n in names)
    assert len(set(names)) == len(names) == len(w)
    manifest = tmp_path / "manifest.txt"
    with open(manifest, "w") as mf:
        for i, (name, arr) in enumerate(named):
            raw = tmp_path / ("t%d.bin" % i)
            np.ascontiguousarray(arr, np.float32).tofile(raw)
            mf.write("%s %d %s %s\n" % (name, arr.ndim, " ".join(str(d) for d in arr.shape), raw))
    path = tmp_path / "model_latest.ot"
    subprocess.check_call([ot_writer, str(manifest), str(path)])
    back = ot.load_ot(path)
    assert set(back) == set(w)
    for k in w:
        assert back[k].shape == w[k].shape and np.array_equal(back[k], w[k]), k


def test_canonical_names_do_not_depend_on_the_suffix_number():
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


def _write_ot(ot_writer, tmp_path, w):
    from takzero_amd import ot

    manifest = tmp_path / "manifest.txt"
    with open(manifest, "w") as mf:
        for i, (name, arr) in enumerate(ot.tch_names(w)):
            raw = tmp_path / ("t%d.bin" % i)
            np.ascontiguousarray(arr, np.float32).tofile(raw)
            mf.write("%s %d %s %s\n" % (name, arr.ndim, " ".join(str(d) for d in arr.shape), raw))
    path = tmp_path / "model_latest.ot"
    subprocess.check_call([ot_writer, str(manifest), str(path)])
    return path


@pytest.mark.gpu
def test_net_load_ot_equals_load_tensors(ot_writer, tmp_path):
    """Network::load (network/mod.rs:24-28) on a LibTorch archive gives the same network as the flat container."""
    import takzero_amd.api as tz
    from takzero_amd import weights as W

    import oracle_lib as O
    from gpu_util import random_positions, require_gpu

    require_gpu()
    oracle = O.load()
    w = W.init_weights(W.ARCH_NET6_SIMHASH, blocks=2, seed=5, trained_stats=True)
    path = _write_ot(ot_writer, tmp_path, w)
    states = O.states_array(random_positions(oracle, O, 6, 4, 24, 3, max_ply=30))
    a = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load_tensors(w)
    b = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load(path)
    ra, rb = a.forward_raw(states), b.forward_raw(states)
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(a.hash_indices(states), b.hash_indices(states))
