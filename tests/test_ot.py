"""The reference's `.ot` model files (tch VarStore::save = LibTorch OutputArchive): a real archive is written with
LibTorch's C++ API under tch's variable names (incl. the `__K` suffix of the second SmallBlock) and read back."""
import numpy as np
import pytest



@pytest.fixture(scope="module")
def ot_writer():
    from takzero_amd import ot

    try:
        return ot.build_writer()
    except RuntimeError as e:
        pytest.skip(str(e))


def test_ot_round_trip_through_libtorch(ot_writer, tmp_path):
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=3, blocks=2, seed=11, trained_stats=True)
    names = [n for n, _ in ot.tch_names(w)]
    # the second SmallBlock of every ResidualBlock collides with the first one's path and gets `__K`
    assert "core.res_block_0.conv2d.weight" in names and any(n.startswith("core.res_block_0.conv2d.weight__") for n in names)
    assert len(set(names)) == len(names) == len(w)
    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    assert not (tmp_path / "model_latest.ot.part").exists()
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
    from takzero_amd import ot

    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    states = O.states_array(random_positions(oracle, O, 6, 4, 24, 3, max_ply=30))
    a = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load_tensors(w)
    b = tz.Net(arch=tz.ARCH_NET6_SIMHASH, blocks=2).load(path)
    ra, rb = a.forward_raw(states), b.forward_raw(states)
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(a.hash_indices(states), b.hash_indices(states))


def test_update_rnd_persistance(ot_writer, tmp_path):
    """net5.rs:327-347 `update_rnd_persistance`: the RND normalisation (min / max variables of the VarStore) survives a
    save / load of the full net5 archive, as do all 28.8 M parameters."""
    from takzero_amd import ot
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_NET5, seed=4)
    assert w["min"].tolist() == [0.0] and w["max"].tolist() == [1.0]   # net5.rs:166-168
    w["min"] = np.float32([0.25])
    w["max"] = np.float32([7.5])
    path = ot.save_ot(tmp_path / "model_latest.ot", w)
    back = ot.load_ot(path)
    assert back["min"].tolist() == [0.25] and back["max"].tolist() == [7.5]
    assert set(back) == set(w) and sum(v.size for v in back.values()) == sum(v.size for v in w.values())
    for k in ("rnd_target.final_linear.weight", "core.res_block_19.b.batch_norm.running_var", "policy.conv2d.bias"):
        assert np.array_equal(back[k], w[k]), k
