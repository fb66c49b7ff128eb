"""CPU study behind TZ_PREC_F16C6 (no GPU): net5 at trained logit scale through torch in float64, every conv's operands
rounded the way a precision would round them.  The fp16 product stays; the two correction products wl*xh + wh*xl run on
**FP6 E2M3** copies of the four operands with one power-of-two scale per block of 32 input channels (the operand form of
v_mfma_scale_f32_16x16x128_f8f6f4: FP6 x FP6 issues in 16 cycles where E4M3 takes 32).  E2M3 has E4M3's three mantissa bits
but two exponent bits: the block scale has to supply the range that E4M3 carries per element.

modes:  f16         fp16 operands only
        fp8         TZ_PREC_F16C8's arithmetic (per-tensor power-of-two scales, E4M3)
        fp6         E2M3 corrections, block of 32 channels per pixel / per (cout, tap), residual stream carried exactly
        fp6carry    + the block input carried as hi + its FP6 lo part only (what a 14-plane image holds)
        fp6carry8   + the block input carried as hi + FP6 lo + one more E4M3 byte of remainder
    python tools/fp6_correction_study.py [positions=48]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
sys.path.insert(0, os.path.join(_root, "tests"))
import oracle_lib as O  # noqa: E402
from takzero_amd import weights as W  # noqa: E402

torch.set_num_threads(8)
lib = O.load()
rng = np.random.default_rng(1)
BLOCK = 32


def positions(count, plies=12):
    out = []
    while len(out) < count:
        s = O.state_default(lib, 5, 4)
        ok = True
        for _ in range(plies):
            mv = O.possible_moves(lib, s)
            if len(mv) == 0 or lib.tzo_terminal(O.C.byref(s)) != -1:
                ok = False
                break
            s = O.play(lib, s, int(mv[rng.integers(len(mv))]))
        if ok and lib.tzo_terminal(O.C.byref(s)) == -1:
            out.append(s)
    return out


B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
st = positions(B)
planes = np.stack([np.asarray(O.game_repr(lib, s), np.float32).reshape(-1, 5, 5) for s in st])
w = W.init_weights(W.ARCH_NET5, seed=123, trained_stats=True)
T = lambda n: torch.from_numpy(np.ascontiguousarray(w[n])).double()


def fold(p, conv="conv2d", bn="batch_norm"):
    g, b, m, v = T(p + "." + bn + ".weight"), T(p + "." + bn + ".bias"), T(p + "." + bn + ".running_mean"), T(p + "." + bn + ".running_var")
    s = g / torch.sqrt(v + 1e-5)
    return (T(p + "." + conv + ".weight") * s[:, None, None, None]).float(), (b - m * s).float()


def fold_in():
    g, b, m, v = T("core.batch_norm.weight"), T("core.batch_norm.bias"), T("core.batch_norm.running_mean"), T("core.batch_norm.running_var")
    s = g / torch.sqrt(v + 1e-5)
    return (T("core.input_conv2d.weight") * s[:, None, None, None]).float(), (b - m * s).float()


h16 = lambda t: t.half().float()


def q8(t, scale):
    return (t * scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() / scale


def p2(v):
    return 2.0 ** np.floor(np.log2(256.0 / max(float(v), 1e-30)))


def e2m3(t):
    """round-to-nearest-even onto the E2M3 grid, saturating at 7.5 (t in units of the block scale)"""
    a = t.abs().double()
    e = torch.floor(torch.log2(torch.clamp(a, min=1e-30)))
    e = torch.clamp(e, min=0.0, max=2.0)          # subnormals share the exponent of [1, 2)
    step = torch.pow(2.0, e - 3.0)
    q = torch.round(a / step) * step              # torch.round is half-to-even
    q = torch.clamp(q, max=7.5)
    return (torch.sign(t).double() * q).float()


def block_scale(amax):
    """power of two s with amax / s <= 7.5:  s = 2^(floor(log2(amax * 16/15)) - 2)"""
    k = torch.floor(torch.log2(torch.clamp(amax.double() * (16.0 / 15.0), min=2.0 ** -60))) - 2.0
    return torch.pow(2.0, k).float()


def q6_act(x):
    """x [B, C, H, W]: one scale per (pixel, block of 32 channels)"""
    b, c, h, wd = x.shape
    xb = x.reshape(b, c // BLOCK, BLOCK, h, wd)
    s = block_scale(xb.abs().amax(dim=2, keepdim=True))
    return (e2m3(xb / s) * s).reshape(b, c, h, wd)


def q6_w(wf):
    """wf [Cout, Cin, 3, 3]: one scale per (cout, tap, block of 32 cin)"""
    co, ci, kh, kw = wf.shape
    if ci % BLOCK:
        pad = BLOCK - ci % BLOCK
        wf = torch.cat([wf, torch.zeros(co, pad, kh, kw)], 1)
    wb = wf.reshape(co, wf.shape[1] // BLOCK, BLOCK, kh, kw)
    s = block_scale(wb.abs().amax(dim=2, keepdim=True))
    return (e2m3(wb / s) * s).reshape(co, -1, kh, kw)[:, :ci]


def conv(x, wf, mode, pad=1):
    if mode == "f32":
        return F.conv2d(x.double(), wf.double(), padding=pad).float()
    xh, wh = h16(x), h16(wf)
    main = F.conv2d(xh.double(), wh.double(), padding=pad)
    if mode == "f16":
        return main.float()
    xl, wl = x - xh, wf - wh
    if mode.startswith("fp8"):
        sx, sw = 4.0, p2(wf.abs().max())
        xh8, xl8 = q8(xh, sx), q8(xl, sx * 2048)
        wh8, wl8 = q8(wh, sw), q8(wl, sw * 2048)
        return (main + F.conv2d(xh8.double(), wl8.double(), padding=pad) + F.conv2d(xl8.double(), wh8.double(), padding=pad)).float()
    if mode.startswith("fp6"):
        if x.shape[1] % BLOCK:      # the first conv (32 planes already; generic guard)
            return (main + F.conv2d(xh.double(), wl.double(), padding=pad) + F.conv2d(xl.double(), wh.double(), padding=pad)).float()
        xh6, xl6 = q6_act(xh), q6_act(xl)
        wh6, wl6 = q6_w(wh), q6_w(wl)
        return (main + F.conv2d(xh6.double(), wl6.double(), padding=pad) + F.conv2d(xl6.double(), wh6.double(), padding=pad)).float()
    raise ValueError(mode)


def carry(x, mode):
    """what the image keeps of a block input (the residual connection reads it back)"""
    if "carry" not in mode:
        return x
    xh = h16(x)
    if mode.startswith("fp8"):
        return xh + q8(x - xh, 4.0 * 2048)
    lo = q6_act(x - xh)
    if mode.endswith("carry8"):
        r = x - xh - lo
        return xh + lo + q8(r, p2(r.abs().max()))
    return xh + lo


def forward(mode, gains=(1, 1, 1)):
    x = torch.from_numpy(planes)
    wf, b = fold_in()
    # the first conv runs the split form (three fp16 products) in the correction precisions: as good as exact here
    x = F.relu(conv(x, wf, "f32" if mode.startswith("fp") else mode) + b[None, :, None, None])
    amax = float(x.max())
    x = carry(x, mode)
    for blk in range(20):
        p = "core.res_block_%d" % blk
        wa, ba = fold(p + ".a")
        wb, bb = fold(p + ".b")
        y = F.relu(conv(x, wa, mode) + ba[None, :, None, None])
        y = conv(y, wb, mode) + bb[None, :, None, None]
        x = F.relu(y + x)
        x = carry(x, mode)
        amax = max(amax, float(x.max()), float(y.abs().max()))
    pol = conv(x, T("policy.conv2d.weight").float() * gains[0], mode) + (T("policy.conv2d.bias").float() * gains[0])[None, :, None, None]
    return pol, amax


if __name__ == "__main__":
    print("planes", planes.shape)
    pol0, amax = forward("f32")
    g = 8.0 / float(pol0.abs().max())
    print("act max", amax, "policy gain", g)
    pol0, _ = forward("f32", (g, 1, 1))
    for mode in ("f16", "fp8", "fp8carry", "fp6", "fp6carry", "fp6carry8"):
        pol, _ = forward(mode, (g, 1, 1))
        print("%-12s max abs logit err %.3g   rms %.3g" % (mode, float((pol - pol0).abs().max()), float((pol - pol0).pow(2).mean().sqrt())), flush=True)
