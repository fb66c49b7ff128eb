"""Plain PyTorch fp32 restatement of the reference's network graphs (test infrastructure only).

    net5          takzero/src/network/net5.rs:44-218      (20 blocks, RND uncertainty)
    net4/6 simhash takzero/src/network/net6_simhash.rs:43-256 (16 blocks, SimHash uncertainty)
    residual      takzero/src/network/residual.rs:13-63

The reference executes these graphs through LibTorch (tch 0.22); torch in this image is the same
ATen code on CPU, so this module is the numeric oracle for the HIP forward.
Parity unpinned for the forward's *values*: the reference holds no input/output vector for its nets (its
tests check shapes), so tests/golden/net_forward.json is this module's output, not the reference's.
Weights come from takzero_amd.weights (name -> ndarray)."""
import numpy as np
import torch
import torch.nn.functional as F

MAXIMUM_VARIANCE = 4.0


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name]))


def _bn(x, w, p):
    return F.batch_norm(x, _t(w, p + ".running_mean"), _t(w, p + ".running_var"), _t(w, p + ".weight"),
                        _t(w, p + ".bias"), training=False, eps=1e-5)


def _small_block(x, w, p):  # residual.rs:13-37
    return _bn(F.conv2d(x, _t(w, p + ".conv2d.weight"), padding=1), w, p + ".batch_norm")


def forward(w, planes, blocks):
    """planes [B,C,N,N] fp32 -> (policy [B,OUT,N,N], value [B], ube [B]); net5.rs:184-191."""
    x = torch.from_numpy(planes) if isinstance(planes, np.ndarray) else planes
    with torch.no_grad():
        x = F.relu(_bn(F.conv2d(x, _t(w, "core.input_conv2d.weight"), padding=1), w, "core.batch_norm"))
        for b in range(blocks):
            p = "core.res_block_%d" % b
            y = _small_block(x, w, p + ".a")
            y = _small_block(F.relu(y), w, p + ".b")
            x = F.relu(y + x)  # residual.rs:58-62
        policy = F.conv2d(x, _t(w, "policy.conv2d.weight"), _t(w, "policy.conv2d.bias"), padding=1)
        heads = []
        for head in ("value", "ube"):
            h = F.relu(F.conv2d(x, _t(w, head + ".conv2d.weight"), _t(w, head + ".conv2d.bias")))
            h = h.view(h.shape[0], -1)
            heads.append(F.linear(h, _t(w, head + ".linear.weight"), _t(w, head + ".linear.bias")).view(-1))
        return policy, torch.tanh(heads[0]), heads[1]


def rnd(w, planes):
    """normalized_rnd, net5.rs:193-211."""
    x = torch.from_numpy(planes) if isinstance(planes, np.ndarray) else planes
    with torch.no_grad():
        x = x.reshape(x.shape[0], -1)
        x = x / x.square().sum(dim=1, keepdim=True)
        outs = []
        for net in ("rnd_learning", "rnd_target"):
            h = F.relu(F.linear(x, _t(w, net + ".input_linear.weight"), _t(w, net + ".input_linear.bias")))
            h = F.relu(F.linear(h, _t(w, net + ".hidden_linear.weight"), _t(w, net + ".hidden_linear.bias")))
            outs.append(F.linear(h, _t(w, net + ".final_linear.weight"), _t(w, net + ".final_linear.bias")))
        raw = (outs[0] - outs[1]).square().sum(dim=1)
        mn, mx = _t(w, "min"), _t(w, "max")
        return ((raw - mn) / (mx - mn)).clamp(0.0, 1.0) * MAXIMUM_VARIANCE


def simhash_indices(w, planes, cin, return_dots=False):
    """get_indices, net6_simhash.rs:202-236 (with return_dots also the 32 projections whose signs are the bits)."""
    x = torch.from_numpy(planes.copy())
    with torch.no_grad():
        x[:, cin - 2] = 0.0
        dots = x.reshape(x.shape[0], -1) @ _t(w, "simhash_matrix")
        bits = (~(dots < 0.0)).to(torch.int64)
        idx = (bits * (2 ** torch.arange(32, dtype=torch.int64))).sum(dim=1).numpy()
        return (idx, dots.numpy()) if return_dots else idx


def variance(w, planes, ube, arch, seen=None):
    """net5.rs:271-278 / net6_simhash.rs:311-318."""
    with torch.no_grad():
        if arch == 5:
            local = rnd(w, planes)
        elif arch in (4, 6):
            idx = simhash_indices(w, planes, planes.shape[1])
            local = torch.tensor([0.0 if (seen is not None and int(i) in seen) else MAXIMUM_VARIANCE for i in idx])
        else:
            local = torch.zeros_like(ube)
        return torch.maximum(ube.exp(), local).clamp(0.0, MAXIMUM_VARIANCE)
