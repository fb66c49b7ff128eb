"""Weight container (.tzw) for libtakzero_hip.so and random initialisation of the reference's
network architectures.

The reference stores weights as a LibTorch archive written by tch's VarStore::save
(takzero/src/network/mod.rs:16-18); reading that format is a "next" row.  This
module defines the flat container tz_net_load_weights() consumes: named fp32 tensors, with the
VarStore's parameter paths as names, except that the two SmallBlocks of a ResidualBlock (which the
reference creates under one path, residual.rs:50-55) are spelled `.a.` and `.b.`.

    magic  b"TZW1", u32 tensor count, then per tensor:
    u16 name length, name (utf-8), u8 ndim, u32 dims[ndim], f32 data (little endian, C order)

Architectures: net5 (takzero/src/network/net5.rs:44-148), net4_simhash /
net6_simhash (net6_simhash.rs:43-141).  `init_weights` follows tch's defaults as the reference
relies on them: conv/linear weights Kaiming-uniform(a=sqrt 5), biases U(+-1/sqrt(fan_in)),
BatchNorm weight U(0,1), bias 0, running stats 0/1, RND min 0 / max 1 (net5.rs:166-168).
"""
import struct

import numpy as np

ARCH_NET4_SIMHASH = 4
ARCH_NET5 = 5
ARCH_NET6_SIMHASH = 6
ARCH_TEST = 100
FILTERS = 256
HASH_BITS = 32
RND_HIDDEN = 1024
RND_OUT = 512


def input_channels(n):
    return 2 * ((3 + (n - 1) + (n + 1)) + 2) + 2


def output_channels(n):
    return 3 + 4 * (2 ** n - 2)


def arch_blocks(arch, blocks=0):
    return {ARCH_NET5: 20, ARCH_NET4_SIMHASH: 16, ARCH_NET6_SIMHASH: 16}.get(arch, blocks)


def arch_board(arch, n=0):
    return {ARCH_NET5: 5, ARCH_NET4_SIMHASH: 4, ARCH_NET6_SIMHASH: 6}.get(arch, n)


def save_tzw(path, tensors):
    with open(path, "wb") as f:
        f.write(dumps_tzw(tensors))


def dumps_tzw(tensors):
    out = [b"TZW1", struct.pack("<I", len(tensors))]
    for name, arr in tensors.items():
        a = np.ascontiguousarray(arr, dtype="<f4")
        nb = name.encode()
        out.append(struct.pack("<H", len(nb)))
        out.append(nb)
        out.append(struct.pack("<B", a.ndim))
        out.append(struct.pack("<%dI" % a.ndim, *a.shape))
        out.append(a.tobytes())
    return b"".join(out)


def load_tzw(path):
    data = open(path, "rb").read()
    assert data[:4] == b"TZW1"
    (count,) = struct.unpack_from("<I", data, 4)
    off, out = 8, {}
    for _ in range(count):
        (ln,) = struct.unpack_from("<H", data, off)
        off += 2
        name = data[off:off + ln].decode()
        off += ln
        nd = data[off]
        off += 1
        dims = struct.unpack_from("<%dI" % nd, data, off)
        off += 4 * nd
        size = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(data, "<f4", size, off).reshape(dims).copy()
        off += 4 * size
    return out


def _kaiming_uniform(rng, shape, fan_in):
    # torch.nn.init.kaiming_uniform_(a=sqrt(5)) -> bound = sqrt(6 / ((1 + 5) * fan_in)) = 1/sqrt(fan_in)
    bound = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _conv(rng, t, name, cout, cin, k, bias):
    fan_in = cin * k * k
    t[name + ".weight"] = _kaiming_uniform(rng, (cout, cin, k, k), fan_in)
    if bias:
        t[name + ".bias"] = _kaiming_uniform(rng, (cout,), fan_in)


def _bn(rng, t, name, c, trained):
    t[name + ".weight"] = rng.uniform(0.0, 1.0, size=c).astype(np.float32)
    t[name + ".bias"] = np.zeros(c, np.float32)
    t[name + ".running_mean"] = np.zeros(c, np.float32)
    t[name + ".running_var"] = np.ones(c, np.float32)
    if trained:  # non-trivial statistics so that BN folding is exercised by the tests
        t[name + ".bias"] = rng.normal(0, 0.1, size=c).astype(np.float32)
        t[name + ".running_mean"] = rng.normal(0, 0.1, size=c).astype(np.float32)
        t[name + ".running_var"] = rng.uniform(0.5, 1.5, size=c).astype(np.float32)


def _linear(rng, t, name, cout, cin):
    t[name + ".weight"] = _kaiming_uniform(rng, (cout, cin), cin)
    t[name + ".bias"] = _kaiming_uniform(rng, (cout,), cin)


def init_weights(arch, n=0, blocks=0, seed=123, trained_stats=False):
    """Random-init weights of `arch` as a dict name -> fp32 ndarray (deterministic in `seed`)."""
    n = arch_board(arch, n)
    blocks = arch_blocks(arch, blocks)
    rng = np.random.default_rng(seed)
    t = {}
    cin, nn = input_channels(n), n * n
    _conv(rng, t, "core.input_conv2d", FILTERS, cin, 3, False)
    _bn(rng, t, "core.batch_norm", FILTERS, trained_stats)
    for b in range(blocks):
        for half in "ab":
            p = "core.res_block_%d.%s" % (b, half)
            _conv(rng, t, p + ".conv2d", FILTERS, FILTERS, 3, False)
            _bn(rng, t, p + ".batch_norm", FILTERS, trained_stats)
    _conv(rng, t, "policy.conv2d", output_channels(n), FILTERS, 3, True)
    for head in ("value", "ube"):
        _conv(rng, t, head + ".conv2d", 1, FILTERS, 1, True)
        _linear(rng, t, head + ".linear", 1, nn)
    if arch == ARCH_NET5:
        for net in ("rnd_learning", "rnd_target"):
            _linear(rng, t, net + ".input_linear", RND_HIDDEN, cin * nn)
            _linear(rng, t, net + ".hidden_linear", RND_HIDDEN, RND_HIDDEN)
            _linear(rng, t, net + ".final_linear", RND_OUT, RND_HIDDEN)
        t["min"] = np.zeros(1, np.float32)
        t["max"] = np.ones(1, np.float32)
    if arch in (ARCH_NET4_SIMHASH, ARCH_NET6_SIMHASH):
        # simhash_matrix: root.var("simhash_matrix", [in_size, 32], Init::Randn{0,1}) net6_simhash.rs:133-137
        t["simhash_matrix"] = rng.normal(0.0, 1.0, size=(cin * nn, HASH_BITS)).astype(np.float32)
    return t


def rescale_heads(tensors, policy_gain=1.0, value_gain=1.0, ube_gain=1.0):
    """A copy of `tensors` whose policy logits, value pre-activation (before tanh) and UBE output are multiplied by the
    given gains (the heads are linear in their last layer).  Random-init logits are ~0.2 in magnitude; a trained
    net5 emits logits of order 5-10, which is where an absolute 1e-3 tolerance has to be demonstrated: the tests and
    bench.py bring a random-init net to that scale with this function."""
    out = dict(tensors)
    for name, gain in (("policy.conv2d", policy_gain), ("value.linear", value_gain), ("ube.linear", ube_gain)):
        for part in (".weight", ".bias"):
            out[name + part] = (np.asarray(tensors[name + part], np.float32) * np.float32(gain)).astype(np.float32)
    return out
