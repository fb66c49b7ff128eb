"""GPU parity of the HIP network forward against a plain PyTorch fp32 reference of the same graph
(oracle/nets_torch.py) and of the fused input encoder against the oracle's game_repr."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

# north_star: policy/value logits within 1e-3 (absolute) of the fp32 LibTorch path.  A random-init net5 emits logits of
# magnitude ~0.2, a trained one 5-10, so the bar is demonstrated on weights brought to that scale (`_trained_scale`).
#   TZ_PREC_F32   plain-FMA validation path: 1e-3 at any scale (measured ~1e-6 relative)
#   TZ_PREC_F16X2 split precision (hi/lo fp16 operands, 3 MFMAs per product): 1e-3 at trained scale - the mode that meets
#                 the north star's tolerance on the MFMA path
#   TZ_PREC_F16C8 the fp16 product plus FP8 (E4M3) correction products: 1e-3 at trained scale as well (measured 1.4e-4), at 2.3x the
#                 fp16 kernel's time instead of 3x
#   TZ_PREC_F16   throughput default (fp16 storage, fp32 accumulate): ~2e-4 relative; 1e-3 absolute only while |logit| <~ 1,
#                 held to a relative bound at trained scale (F16_REL_TOL) and reported
#   TZ_PREC_BF16  same kernels, 5 % faster: explicit absolute bounds at random-init scale
F32_TOL = 1e-3
BF16_LOGIT_TOL, BF16_VALUE_TOL = 1.2e-2, 5e-3
F16_REL_TOL = 2e-3          # max |delta logit| / max |logit| of the fp16 default at trained scale (measured 1.0e-3)
TRAINED_LOGIT = 8.0         # target max |logit| of the rescaled nets (a trained net5: 5-10)


def _planes(oracle, states):
    return np.stack([O.game_repr(oracle, s) for s in states])


@pytest.mark.parametrize("n", [3, 4, 5, 6])
def test_encoder_matches_game_repr(oracle, n):
    A = require_gpu()
    from takzero_amd import weights as W

    states = random_positions(oracle, O, n, 4, 40, 10 + n, max_ply=50)
    net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_F32, blocks=1)
    net.load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=1))
    got = net.encode(O.states_array(states))
    want = _planes(oracle, states)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _trained_scale(w, planes, blocks):
    """Heads rescaled so that the fp32 torch graph emits max |logit| = TRAINED_LOGIT, a value pre-activation of +-1.5
    (tanh still sensitive) and |ube| up to 2 on these positions: the output scale of a trained net."""
    import nets_torch as T
    import torch
    from takzero_amd import weights as W

    pol, val, ube = T.forward(w, planes, blocks)
    pre = torch.atanh(val.clamp(-0.999999, 0.999999))
    return W.rescale_heads(w, TRAINED_LOGIT / float(pol.abs().max()), 1.5 / max(1e-6, float(pre.abs().max())),
                           2.0 / max(1e-6, float(ube.abs().max())))


def _compare(A, oracle, arch, n, blocks, prec, batch, seed, trained, trained_scale=False):
    import nets_torch as T
    from takzero_amd import weights as W

    w = W.init_weights(arch, n=n, blocks=blocks, seed=seed, trained_stats=trained)
    n = W.arch_board(arch, n)
    blocks = W.arch_blocks(arch, blocks)
    states = random_positions(oracle, O, n, 4, batch, seed, max_ply=40)
    planes = _planes(oracle, states).reshape(batch, -1, n, n)
    if trained_scale:
        w = _trained_scale(w, planes, blocks)
    pol_t, val_t, ube_t = T.forward(w, planes, blocks)
    var_t = T.variance(w, planes, ube_t, arch if arch != W.ARCH_TEST else 0).numpy()
    pol_t = pol_t.reshape(batch, -1).numpy()
    net = A.Net(arch=arch, n=n, precision=prec, blocks=blocks)
    net.load_tensors(w)
    arr = O.states_array(states)
    pol, val, ube = net.forward_raw(arr)
    acts = [O.possible_moves(oracle, s) for s in states]
    logits, val2, var = net.policy_value_uncertainty(arr, acts)
    # the Agent surface returns exactly the legal logits in order, and the same values
    for i, a in enumerate(acts):
        assert np.array_equal(logits[i], pol[i, a])
    assert np.array_equal(val, val2)
    err = dict(policy=np.abs(pol - pol_t).max(), value=np.abs(val - val_t.numpy()).max(),
               ube=np.abs(ube - ube_t.numpy()).max(), variance=np.abs(var - var_t).max(),
               scale=np.abs(pol_t).max(), value_scale=np.abs(val_t.numpy()).max(), ube_scale=np.abs(ube_t.numpy()).max())
    net.close()
    return err


@pytest.mark.parametrize("arch,n,blocks,batch", [(100, 5, 3, 19), (100, 3, 2, 33), (100, 4, 2, 13), (100, 6, 2, 9),
                                                 (5, 5, 20, 10)])
def test_f32_path_within_1e_3_of_torch(oracle, arch, n, blocks, batch):
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F32, batch, 42, True)
    print("f32 errors", err)
    assert err["policy"] < F32_TOL and err["value"] < F32_TOL and err["ube"] < F32_TOL and err["variance"] < 4e-3


@pytest.mark.parametrize("arch,n,blocks,batch", [(100, 5, 3, 19), (100, 3, 2, 33), (100, 4, 2, 13), (100, 6, 2, 9),
                                                 (5, 5, 20, 21), (6, 6, 16, 7), (4, 4, 16, 15)])
def test_bf16_mfma_path_close_to_torch(oracle, arch, n, blocks, batch):
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_BF16, batch, 43, True)
    print("bf16 errors", err)
    assert err["policy"] < BF16_LOGIT_TOL and err["value"] < BF16_VALUE_TOL and err["ube"] < BF16_VALUE_TOL


NETS = [(5, 5, 20, 21), (100, 5, 3, 19), (6, 6, 16, 7), (4, 4, 16, 15), (100, 3, 2, 33)]


@pytest.mark.parametrize("arch,n,blocks,batch", NETS)
def test_f16_mfma_path_within_1e_3_of_torch_at_random_init_scale(oracle, arch, n, blocks, batch):
    """TZ_PREC_F16 on random-init weights (|logit| ~ 0.2, the weights bench.py runs): within 1e-3 absolute.  This says
    ~1e-3 *relative*; the trained-scale cases below are the ones that carry the north star's tolerance."""
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F16, batch, 44, False)
    print("f16 errors (random-init scale)", err)
    assert err["policy"] < F32_TOL and err["value"] < F32_TOL and err["ube"] < 2 * F32_TOL


@pytest.mark.parametrize("arch,n,blocks,batch", NETS)
def test_f16x2_split_precision_within_1e_3_of_torch_at_trained_logit_scale(oracle, arch, n, blocks, batch):
    """TZ_PREC_F16X2 with BatchNorm statistics of a trained net and heads at a trained net's output scale (max |logit| = 8,
    value pre-activation 1.5, |ube| 2): policy, value and UBE within the north star's absolute 1e-3 of the fp32 LibTorch
    graph (net5.rs:184-191,237-238) on the full nets of every board size."""
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F16X2, batch, 45, True, trained_scale=True)
    print("f16x2 errors (trained scale)", err)
    assert 7.9 < err["scale"] < 8.1
    assert err["policy"] < F32_TOL and err["value"] < F32_TOL and err["ube"] < F32_TOL


@pytest.mark.parametrize("arch,n,blocks,batch", NETS)
def test_f16c8_fp8_corrections_within_1e_3_of_torch_at_trained_logit_scale(oracle, arch, n, blocks, batch):
    """TZ_PREC_F16C8 on the same weights: the fp16 product wh*xh plus the correction products wl*xh + wh*xl on FP8 (E4M3) copies
    of the operands (v_mfma_f32_16x16x128_f8f6f4), block inputs carried as hi + two FP8 bytes.  Same absolute 1e-3 on policy,
    value and UBE against the fp32 LibTorch graph, every board size; measured 1.4e-4 on net5 (60x closer than TZ_PREC_F16)."""
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F16C8, batch, 45, True, trained_scale=True)
    print("f16c8 errors (trained scale)", err)
    assert 7.9 < err["scale"] < 8.1
    assert err["policy"] < F32_TOL and err["value"] < F32_TOL and err["ube"] < F32_TOL
    assert err["policy"] < 5e-4   # measured 1.4e-4 (5x5) .. : a regression of the corrections would show here long before 1e-3


def test_f16c8_saturating_activations_stay_finite(oracle):
    """The FP8 copies saturate (hi copy at 112, lo part at its largest code) instead of becoming NaN - the conversion instruction
    itself turns overflow into NaN: a net whose first conv emits activations in the thousands still gives finite outputs, and
    they stay close to the fp32 path's (the corrections of the saturated elements are lost, the fp16 products are not)."""
    A = require_gpu()
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=5, blocks=2, seed=3, trained_stats=True)
    w = dict(w)
    w["core.batch_norm.weight"] = (np.asarray(w["core.batch_norm.weight"], np.float32) * np.float32(3000.0)).astype(np.float32)
    states = O.states_array(random_positions(oracle, O, 5, 4, 16, 5))
    outs = {}
    for prec in (A.PREC_F32, A.PREC_F16C8):
        net = A.Net(arch=A.ARCH_TEST, n=5, precision=prec, blocks=2)
        net.load_tensors(w)
        outs[prec] = net.forward_raw(states)
        net.close()
    pol, val, ube = outs[A.PREC_F16C8]
    assert np.isfinite(pol).all() and np.isfinite(val).all() and np.isfinite(ube).all()
    scale = float(np.abs(outs[A.PREC_F32][0]).max())
    print("saturating net: |logit| max", scale, "max error", float(np.abs(pol - outs[A.PREC_F32][0]).max()))
    assert float(np.abs(pol - outs[A.PREC_F32][0]).max()) < 2e-3 * scale


C6_NETS = [(100, 5, 3, 19), (100, 6, 2, 9), (5, 5, 20, 10), (6, 6, 16, 6)]


@pytest.mark.parametrize("arch,n,blocks,batch", C6_NETS)
def test_f16c6_fp6_block_scaled_corrections_within_1e_3_of_torch_at_trained_logit_scale(oracle, arch, n, blocks, batch):
    """TZ_PREC_F16C6 on the same weights: the fp16 product wh*xh plus the correction products wl*xh + wh*xl on FP6 (E2M3) copies of the
    operands with one power-of-two scale per block of 32 input channels (v_mfma_scale_f32_16x16x128_f8f6f4), the blocks' inputs
    carried in fp32 between the two convs of a block, value / UBE heads on the fp32 tower output.  Same absolute 1e-3 on policy,
    value and UBE against the fp32 LibTorch graph (net5.rs:184-191,237-238); measured 1.4e-4 on net5 like TZ_PREC_F16C8."""
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F16C6, batch, 45, True, trained_scale=True)
    print("f16c6 errors (trained scale)", err)
    assert 7.9 < err["scale"] < 8.1
    assert err["policy"] < F32_TOL and err["value"] < F32_TOL and err["ube"] < F32_TOL
    assert err["policy"] < 5e-4 and err["value"] < 2e-4 and err["ube"] < 2e-4


def test_f16c6_is_much_closer_than_fp16_at_random_init_scale_too(oracle):
    """At random-init scale (|logit| ~ 0.2) the fp16 default is 1.5e-4 off the fp32 graph; the corrections bring that to a few 1e-6."""
    A = require_gpu()
    err = _compare(A, oracle, 5, 5, 20, A.PREC_F16C6, 10, 123, False)
    print("f16c6 errors (random init)", err)
    assert err["policy"] < 2e-5 and err["value"] < 2e-5 and err["ube"] < 2e-5


def test_f16c6_large_activations_stay_finite(oracle):
    """Activations in the thousands: the E2M3 copies are block-scaled (no fixed range to leave), stored halves saturate at 65504."""
    A = require_gpu()
    from takzero_amd import weights as W

    w = dict(W.init_weights(W.ARCH_TEST, n=5, blocks=2, seed=3, trained_stats=True))
    w["core.batch_norm.weight"] = (np.asarray(w["core.batch_norm.weight"], np.float32) * np.float32(3000.0)).astype(np.float32)
    states = O.states_array(random_positions(oracle, O, 5, 4, 16, 5))
    outs = {}
    for prec in (A.PREC_F32, A.PREC_F16C6):
        net = A.Net(arch=A.ARCH_TEST, n=5, precision=prec, blocks=2)
        net.load_tensors(w)
        outs[prec] = net.forward_raw(states)
        net.close()
    pol, val, ube = outs[A.PREC_F16C6]
    assert np.isfinite(pol).all() and np.isfinite(val).all() and np.isfinite(ube).all()
    scale = float(np.abs(outs[A.PREC_F32][0]).max())
    print("large activations: |logit| max", scale, "max error", float(np.abs(pol - outs[A.PREC_F32][0]).max()))
    assert float(np.abs(pol - outs[A.PREC_F32][0]).max()) < 1e-4 * scale


def test_f16c6_is_refused_on_board_sizes_it_is_not_built_for():
    A = require_gpu()
    with pytest.raises(Exception):
        A.Net(arch=A.ARCH_TEST, n=4, precision=A.PREC_F16C6, blocks=2)


@pytest.mark.parametrize("arch,n,blocks,batch", NETS)
def test_f16_default_at_trained_logit_scale_is_relative(oracle, arch, n, blocks, batch):
    """The fp16 throughput default on the same trained-scale weights: its error is relative (~2e-4 of the logit scale
    per 41 convs), so at |logit| = 8 it does NOT meet 1e-3 absolute (measured 8e-3 on net5); it is held to 2e-3 of the
    logit scale and the figure is printed (Precision; profiles/r02_precision.json)."""
    A = require_gpu()
    err = _compare(A, oracle, arch, n, blocks, A.PREC_F16, batch, 45, True, trained_scale=True)
    print("f16 errors (trained scale)", err)
    assert err["policy"] < F16_REL_TOL * err["scale"] and err["value"] < 5e-3 and err["ube"] < 2 * F16_REL_TOL * max(2.0, err["ube_scale"])


def test_f32_path_at_trained_logit_scale(oracle):
    A = require_gpu()
    err = _compare(A, oracle, 5, 5, 20, A.PREC_F32, 10, 45, True, trained_scale=True)
    print("f32 errors (trained scale)", err)
    assert err["policy"] < 1e-4 and err["value"] < 1e-4 and err["ube"] < 1e-4


@pytest.mark.parametrize("prec,n", [(0, 5), (2, 5), (3, 5), (3, 6), (3, 4), (3, 3), (4, 5), (4, 6), (4, 4), (4, 3), (5, 5), (5, 6)])
def test_forward_is_batch_composition_independent(oracle, prec, n):
    """A position's outputs do not depend on its slot or its neighbours (needed so that the oracle can
    replay the engine's network calls one position at a time): bf16, fp16, the split-precision kernel and the FP8-correction
    kernel (whose workgroups hold half the boards: 4 / 2 / 6 / 8 on 5x5 / 6x6 / 4x4 / 3x3), ragged batch sizes incl. a single position."""
    A = require_gpu()
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=n, blocks=2, seed=5)
    net = A.Net(arch=A.ARCH_TEST, n=n, precision=prec, blocks=2)
    net.load_tensors(w)
    states = random_positions(oracle, O, n, 4, 37, 9)
    arr = O.states_array(states)
    pol, val, ube = net.forward_raw(arr)
    rng = np.random.default_rng(0)
    for count in (11, 1, 5, 2):
        perm = rng.permutation(37)[:count]
        pol2, val2, ube2 = net.forward_raw(arr[perm])
        assert np.array_equal(pol[perm], pol2) and np.array_equal(val[perm], val2) and np.array_equal(ube[perm], ube2), count


@pytest.mark.parametrize("prec", [3, 4, 5])
def test_workgroup_forms_of_the_split_precisions_on_6x6_give_the_same_bits(oracle, prec):
    """6x6 in a split precision: 2 boards per workgroup (board-major rows) below 1024 positions, 4 boards (square-major rows, 12 of
    81 (tap, tile) pairs skipped, tap table of 8-bit rows) from there on - the same bits for a position either way."""
    A = require_gpu()
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_NET6_SIMHASH, precision=prec)
    net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH, seed=9))
    base = O.states_array(random_positions(oracle, O, 6, 4, 64, 23))
    states = np.concatenate([base] * 18)[:1100]
    big = net.forward_raw(states)
    for count in (1027, 300, 5):
        part = net.forward_raw(states[:count])
        for x, y in zip(big, part):
            assert np.array_equal(x[:count], y), (prec, count)
    net.close()


@pytest.mark.parametrize("prec", [2, 4, 5])
def test_workgroup_forms_give_the_same_bits(oracle, prec):
    """Small batches run on 1- and 2-board workgroups (board-major rows), large ones on 8 (fp16) or 4 (fp16 + FP8 corrections)
    boards in square-major order with the all-padding (tap, tile) pairs left out: a position's outputs are the same bits in
    every form (same k order, what is skipped adds exact zeros)."""
    A = require_gpu()
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_NET5, precision=prec)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=9))
    base = O.states_array(random_positions(oracle, O, 5, 4, 64, 21))
    states = np.concatenate([base] * 18)[:1100]
    big = net.forward_raw(states)                      # > 1024 positions: the full-size workgroups
    for count in (1030, 600, 300, 100):                # full size again (ragged), 4, 2 and 1 boards per workgroup
        part = net.forward_raw(states[:count])
        for x, y in zip(big, part):
            assert np.array_equal(x[:count], y), (prec, count)
    net.close()


def test_failed_load_keeps_old_weights(oracle, tmp_path):
    A = require_gpu()
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=5, blocks=1, seed=5)
    net = A.Net(arch=A.ARCH_TEST, n=5, precision=A.PREC_BF16, blocks=1)
    path = tmp_path / "w.tzw"
    W.save_tzw(str(path), w)
    net.load(path)
    states = O.states_array(random_positions(oracle, O, 5, 4, 4, 1))
    before = net.forward_raw(states)[0]
    bad = dict(w)
    del bad["policy.conv2d.bias"]
    with pytest.raises(A.TakzeroError):
        net.load_tensors(bad)
    assert np.array_equal(before, net.forward_raw(states)[0])


def test_simhash_indices_counts_and_bitvec_file(oracle, tmp_path):
    """net4_simhash.rs:370-430 (`counts_work`, `saving_works`) on the HIP SimHash path, indices against torch."""
    A = require_gpu()
    import nets_torch as T
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_NET4_SIMHASH, seed=3)
    net = A.Net(arch=A.ARCH_NET4_SIMHASH, precision=A.PREC_BF16)
    net.load_tensors(w)
    states = random_positions(oracle, O, 4, 4, 24, 5, max_ply=20)
    arr = O.states_array(states)
    planes = _planes(oracle, states).reshape(24, -1, 4, 4)
    want, dots = T.simhash_indices(w, planes, planes.shape[1], return_dots=True)
    want = want.astype(np.uint32)
    got = net.hash_indices(arr)
    # every bit is the sign of a 448-term fp32 dot product (net6_simhash.rs:224-233): a bit may differ from torch's only where
    # that projection is within summation-order rounding of zero
    differ = ((got[:, None] ^ want[:, None]) >> np.arange(32, dtype=np.uint32)[None, :]) & 1
    assert np.all(np.abs(dots[differ == 1]) < 1e-4), (np.abs(dots[differ == 1]).max(), int(differ.sum()))
    assert (got == want).mean() >= 0.9 and np.abs(dots).min() < 1.0
    acts = [O.possible_moves(oracle, s) for s in states]
    var0 = net.policy_value_uncertainty(arr, acts)[2]
    assert np.all(var0 == 4.0)           # nothing seen yet: maximum variance (net6_simhash.rs:246-255)
    net.hash_indices(arr[:12], update=True)
    var1 = net.policy_value_uncertainty(arr, acts)[2]
    seen = np.isin(got, got[:12])
    assert np.all(var1[seen] < 4.0) and np.all(var1[~seen] == 4.0)
    path = tmp_path / "bitvec.bin"
    net.save_bitset(path)
    assert path.stat().st_size == 1 << 29
    net2 = A.Net(arch=A.ARCH_NET4_SIMHASH, precision=A.PREC_BF16)
    net2.load_tensors(w)
    net2.load_bitset(path)
    assert np.array_equal(net2.policy_value_uncertainty(arr, acts)[2], var1)
    with pytest.raises(A.TakzeroError):
        net2.load_bitset(tmp_path / "missing.bin")


def test_network_lifecycle_new_save_load_partial_clone(oracle, tmp_path):
    """Network::{new, save, load, load_partial, clone} (network/mod.rs:10-45) at the C ABI (tz_net_init_random, tz_net_save,
    tz_net_load_weights, tz_net_load_partial, tz_net_clone): no Python-side copy of the variables is involved."""
    A = require_gpu()
    from takzero_amd import ot
    from takzero_amd import weights as W

    n, blocks = 4, 1
    a = A.Net.new(arch=A.ARCH_TEST, seed=5, n=n, blocks=blocks)
    assert set(a.tensors()) == set(W.init_weights(W.ARCH_TEST, n=n, blocks=blocks))
    b5 = A.Net.new(arch=A.ARCH_TEST, seed=5, n=n, blocks=blocks)
    b6 = A.Net.new(arch=A.ARCH_TEST, seed=6, n=n, blocks=blocks)
    states = O.states_array(random_positions(oracle, O, n, 4, 6, 3))
    want = a.forward_raw(states)
    assert all(np.array_equal(x, y) for x, y in zip(b5.forward_raw(states), want))        # new(seed) is a function of the seed
    assert not np.array_equal(b6.forward_raw(states)[0], want[0])
    t = a.tensors()
    bound = 1.0 / np.sqrt(256 * 9)      # tch's defaults: Kaiming-uniform(a = sqrt 5) = U(+-1/sqrt(fan_in)); BN weight U(0,1), stats 0 / 1
    cw = t["core.res_block_0.a.conv2d.weight"]
    assert np.abs(cw).max() <= bound and np.abs(cw).max() > 0.98 * bound and abs(float(cw.mean())) < 1e-3
    assert 0 <= t["core.batch_norm.weight"].min() and t["core.batch_norm.weight"].max() <= 1 and np.all(t["core.batch_norm.running_var"] == 1)
    a.save(tmp_path / "model_0000000.ot")
    a.save(tmp_path / "model.tzw")
    assert sorted(os.listdir(tmp_path)) == ["model.tzw", "model_0000000.ot"]            # no .part left behind
    for path in ("model_0000000.ot", "model.tzw"):
        b = A.Net(arch=A.ARCH_TEST, n=n, blocks=blocks).load(tmp_path / path)
        assert all(np.array_equal(x, y) for x, y in zip(b.forward_raw(states), want)), path
    # what tz_net_save wrote is a LibTorch archive under tch's variable names: LibTorch's own reader agrees
    named = ot.read_ot_libtorch(tmp_path / "model_0000000.ot")
    assert "core.res_block_0.conv2d.weight__10" in named and np.array_equal(named["policy.conv2d.bias"], t["policy.conv2d.bias"])
    c = a.clone(0)
    assert all(np.array_equal(x, y) for x, y in zip(c.forward_raw(states), want))
    # load_partial: a file without the policy head leaves that head as it was and reports it
    other = b6.tensors()
    other = {k: v.reshape(np.shape(W.init_weights(W.ARCH_TEST, n=n, blocks=blocks)[k])) for k, v in other.items()}
    partial = {k: v for k, v in other.items() if not k.startswith("policy.")}
    W.save_tzw(tmp_path / "partial.tzw", partial)
    missing = a.load_partial(tmp_path / "partial.tzw")
    assert sorted(missing) == ["policy.conv2d.bias", "policy.conv2d.weight"]
    mixed = dict(other)
    mixed.update({k: v.reshape(np.shape(other[k])) for k, v in t.items() if k.startswith("policy.")})
    d = A.Net(arch=A.ARCH_TEST, n=n, blocks=blocks).load_tensors(mixed)
    assert all(np.array_equal(x, y) for x, y in zip(a.forward_raw(states), d.forward_raw(states)))
    assert all(np.array_equal(x, y) for x, y in zip(c.forward_raw(states), want))        # the clone has its own variables
    # a failed load (truncated archive) leaves the weights alone
    blob = open(tmp_path / "model_0000000.ot", "rb").read()
    open(tmp_path / "torn.ot", "wb").write(blob[:len(blob) // 3])
    with pytest.raises(A.TakzeroError):
        c.load(tmp_path / "torn.ot")
    assert all(np.array_equal(x, y) for x, y in zip(c.forward_raw(states), want))


def test_load_in_two_halves_while_the_net_is_evaluating(oracle, tmp_path):
    """tz_net_load_prepare on another thread while the network keeps evaluating (the self-play process does not stop for a new
    model_latest.ot), tz_net_load_commit between two forwards: outputs before the commit are the old model's, after it the new
    model's, bit for bit what a plain load gives; a prepare of a torn file fails and leaves nothing behind."""
    import threading

    A = require_gpu()
    n, blocks = 5, 3
    old = A.Net.new(arch=A.ARCH_TEST, seed=5, n=n, blocks=blocks)
    new = A.Net.new(arch=A.ARCH_TEST, seed=6, n=n, blocks=blocks)
    new.save(tmp_path / "model_latest.ot")
    states = O.states_array(random_positions(oracle, O, n, 4, 64, 3))
    want_old, want_new = old.forward_raw(states), new.forward_raw(states)
    box = {}

    def prepare():
        box["pending"] = old.load_prepare(tmp_path / "model_latest.ot")

    th = threading.Thread(target=prepare)
    th.start()
    during = [old.forward_raw(states) for _ in range(20)]          # evaluating while the other thread parses, converts and uploads
    th.join()
    for out in during + [old.forward_raw(states)]:
        assert all(np.array_equal(x, y) for x, y in zip(out, want_old))
    old.load_commit(box["pending"])
    assert all(np.array_equal(x, y) for x, y in zip(old.forward_raw(states), want_new))
    blob = open(tmp_path / "model_latest.ot", "rb").read()
    open(tmp_path / "torn.ot", "wb").write(blob[:len(blob) // 2])
    with pytest.raises(A.TakzeroError):
        old.load_prepare(tmp_path / "torn.ot")
    assert all(np.array_equal(x, y) for x, y in zip(old.forward_raw(states), want_new))


def test_simhash_net_saves_and_loads_its_set_beside_the_model(oracle, tmp_path):
    """net6_simhash.rs:152-190: Network::save writes bitvec.bin beside the .ot, Network::load reads it back; clone copies it."""
    A = require_gpu()

    a = A.Net.new(arch=A.ARCH_NET4_SIMHASH, seed=2)
    positions = random_positions(oracle, O, 4, 4, 16, 5, max_ply=20)
    states = O.states_array(positions)
    acts = [O.possible_moves(oracle, s) for s in positions]
    a.hash_indices(states[:8], update=True)
    var = a.policy_value_uncertainty(states, acts)[2]
    assert np.any(var < 4.0) and np.any(var == 4.0)
    a.save(tmp_path / "model_latest.ot")
    assert (tmp_path / "bitvec.bin").stat().st_size == 1 << 29
    b = A.Net(arch=A.ARCH_NET4_SIMHASH).load(tmp_path / "model_latest.ot")
    assert np.array_equal(b.policy_value_uncertainty(states, acts)[2], var)
    assert np.array_equal(a.clone(0).policy_value_uncertainty(states, acts)[2], var)


@pytest.mark.gpu
def test_net_launch_forms_are_bit_identical():
    """The net kernel's square-major row order leaves out the (tap, row tile) pairs that only multiply zero padding
    (csrc/tz_nn.hip RowMap), and small batches run on 1-, 2- or 4-board workgroups: policy, value and UBE must equal, byte
    for byte, what the board-major kernel with full-size workgroups that issues every pair produces (TZ_NET_ROWS=board
    TZ_NET_P=full) - 3x3 / 5x5 / 6x6 nets, bf16 and f16, batch sizes 1 .. 1030 that leave partial workgroups."""
    require_gpu()
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "net_rows_ab.py"), "3", "5", "6"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "BIT-IDENTICAL" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n", [4, 5, 6])
def test_agent_surface_with_four_cus_per_board_group_gives_the_same_bits(oracle, n):
    """tz_net_eval on 5x5 at batches up to 256 (the reference's batch is 128, selfplay/src/main.rs:37) runs net_mfma_kernel's SPLIT form: four
    workgroups per group of one, two or four boards, each computing a quarter of every conv's output channels and handing its planes to the
    others after every layer through a buffer that the four meet in by hand inside one XCD's L2 (csrc/tz_nn.hip).  The k-loop of an
    output is the one-CU form's, so logits, value and variance must equal — bit for bit — what the same positions give through
    tz_net_forward_raw (one CU per group) and through tz_net_eval with TZ_NET_SPLIT=0, at every batch size incl. partial groups and
    octets, in both 16-bit storage types, and call after call (a stale read of a partner's planes would differ from run to run)."""
    import subprocess
    import sys

    A = require_gpu()
    from takzero_amd import weights as W

    arch, wname = {4: (A.ARCH_NET4_SIMHASH, "ARCH_NET4_SIMHASH"), 5: (A.ARCH_NET5, "ARCH_NET5"), 6: (A.ARCH_NET6_SIMHASH, "ARCH_NET6_SIMHASH")}[n]   # 4x4, 6x6: groups of one or two boards, up to 128 positions
    states = random_positions(oracle, O, n, 4, 512 if n == 5 else 200, 17, max_ply=30)
    arr = O.states_array(states)
    acts = [O.possible_moves(oracle, s) for s in states]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sizes = (1, 7, 64, 65, 100, 128, 129, 256, 300, 512) if n == 5 else (1, 7, 64, 65, 128, 200)   # 1, 2 and 4 boards per group, partial groups and octets of groups; past 256 the one-CU forms
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import takzero_amd.api as A; from takzero_amd import weights as W\n"
            "d = np.load(sys.argv[1], allow_pickle=True); arr = d['arr'].view(A._lib.STATE_DTYPE).reshape(-1); acts = list(d['acts'])\n"
            "net = A.Net(arch=A.%s, precision=int(sys.argv[2])).load_tensors(W.init_weights(W.%s, seed=5))\n"
            "out = {}\n"
            "for B in %r:\n"
            "    l, v, u = net.policy_value_uncertainty(arr[:B], acts[:B]); out['l%%d' %% B] = np.concatenate(l); out['v%%d' %% B] = v; out['u%%d' %% B] = u\n"
            "np.savez(sys.argv[3], **out)\n") % (root, os.path.join(root, "tests"), wname, wname, sizes)
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "in.npz"), arr=np.frombuffer(arr.tobytes(), np.uint8), acts=np.array(acts, dtype=object))
        # 5x5 also in the hi / lo split precision (TZ_PREC_F16X2, the arithmetic that holds the 1e-3 tolerance on trained nets)
        for prec in (A.PREC_F16, A.PREC_BF16) + ((A.PREC_F16X2,) if n == 5 else ()):
            r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "in.npz"), str(prec), os.path.join(d, "off.npz")],
                               env=dict(os.environ, TZ_NET_SPLIT="0"), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-1500:]
            off = np.load(os.path.join(d, "off.npz"))
            net = A.Net(arch=arch, precision=prec).load_tensors(W.init_weights(getattr(W, wname), seed=5))
            pol, val, _ = net.forward_raw(arr)
            for B in sizes:
                for rep in range(6 if B == 128 else 2):
                    l, v, u = net.policy_value_uncertainty(arr[:B], acts[:B])
                    assert np.array_equal(np.concatenate(l), off["l%d" % B]) and np.array_equal(v, off["v%d" % B]) and np.array_equal(u, off["u%d" % B]), (prec, B, rep)
                    assert all(np.array_equal(l[i], pol[i, np.asarray(acts[i], np.int64)]) for i in range(B)) and np.array_equal(v, val[:B])
            net.close()


def test_two_processes_on_one_gpu_both_on_the_several_cu_form_write_the_same_bytes(tmp_path):
    """The reference's deployment puts selfplay and reanalyze of 128 games each on one GPU (README.md:130).  At that width both run
    the net kernel's several-CU form, whose members wait for each other inside the kernel: two such processes at once must neither
    stall each other (a member whose partners are not resident yet waits; complete groups keep finishing and free their CUs) nor
    change a result — each process writes, byte for byte, the targets and replays it writes when it has the GPU to itself."""
    import subprocess
    import sys

    require_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import takzero_amd.api as A\n"
            "from takzero_amd import selfplay as SP, weights as W\n"
            "net = A.Net(arch=A.ARCH_NET5).load_tensors(W.init_weights(W.ARCH_NET5, seed=int(sys.argv[1])))\n"
            "m = A.BatchedMCTS(128, 5, 4, agent=net, node_capacity=1 << 16)\n"
            "sp = SP.NativeSelfPlay(m, 48, seed=int(sys.argv[1]), shard=0, search='puct')\n"
            "for _ in range(25): sp.play_move()\n"
            "open(sys.argv[2], 'wb').write(sp.take_text(0) + b'#' + sp.take_text(1))\n") % root
    solo = []
    for seed in (1, 2):
        out = str(tmp_path / ("solo%d.bin" % seed))
        r = subprocess.run([sys.executable, "-c", code, str(seed), out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        solo.append(open(out, "rb").read())
    assert len(solo[0]) > 1000 and solo[0] != solo[1]
    procs = [subprocess.Popen([sys.executable, "-c", code, str(seed), str(tmp_path / ("pair%d.bin" % seed))], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for seed in (1, 2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    for seed, want in zip((1, 2), solo):
        assert open(tmp_path / ("pair%d.bin" % seed), "rb").read() == want
