"""TEST INFRASTRUCTURE: the reference's training step restated in plain PyTorch fp32 with autograd — the same ATen
kernels tch calls (learn/src/main.rs:376-423: forward_t(xs, true), masked log-softmax cross entropy, value MSE, UBE
MSE, nn::Adam::default()).  Graph: net5.rs:44-191 / net6_simhash.rs:43-141, residual.rs:13-63."""
import numpy as np
import torch
import torch.nn.functional as F

MINIMUM_UBE_TARGET = -10.0  # learn/src/main.rs:47
MAXIMUM_VARIANCE = 4.0      # net5.rs:23
TRAINED_PREFIXES = ("core.", "policy.", "value.", "ube.")


def make_params(weights):
    """name -> torch tensor; trainable ones require grad, BatchNorm running statistics are plain buffers."""
    out = {}
    for name, arr in weights.items():
        if not name.startswith(TRAINED_PREFIXES):
            continue
        t = torch.from_numpy(np.array(arr, dtype=np.float32, copy=True))
        if "running_" not in name:
            t.requires_grad_(True)
        out[name] = t
    return out


def _bn(x, p, prefix, train):
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"],
                        p[prefix + ".bias"], training=train, momentum=0.1, eps=1e-5)


def forward_t(p, planes, blocks, train=True, relu_masks=None):
    """RndNetwork::forward_t (net5.rs:184-191): (policy [B, OUT*N*N], value [B, 1], ube [B, 1]).
    relu_masks (a list of bool tensors, one per trunk ReLU in order): where a ReLU's input is within 1e-5 of zero — closer than two
    fp32 implementations agree on it — the mask says which side of the kink to take (the side the implementation under test took,
    read from its stored activations); everywhere else, and in the heads, it is the input's own sign."""
    taken = [0]

    def relu(t):
        i = taken[0]
        taken[0] += 1
        if relu_masks is None or i >= len(relu_masks):
            return F.relu(t)
        return t * torch.where(t.detach().abs() < 1e-5, relu_masks[i], t.detach() > 0).to(t.dtype)

    x = relu(_bn(F.conv2d(planes, p["core.input_conv2d.weight"], padding=1), p, "core.batch_norm", train))
    for b in range(blocks):
        q = "core.res_block_%d" % b
        y = _bn(F.conv2d(x, p[q + ".a.conv2d.weight"], padding=1), p, q + ".a.batch_norm", train)
        y = _bn(F.conv2d(relu(y), p[q + ".b.conv2d.weight"], padding=1), p, q + ".b.batch_norm", train)
        x = relu(y + x)
    policy = F.conv2d(x, p["policy.conv2d.weight"], p["policy.conv2d.bias"], padding=1)
    heads = []
    for head, core in (("value", x), ("ube", x.detach())):  # "Detached UBE so it does not mess with baseline"
        h = relu(F.conv2d(core, p[head + ".conv2d.weight"], p[head + ".conv2d.bias"]))
        h = h.view(h.shape[0], -1)
        heads.append(F.linear(h, p[head + ".linear.weight"], p[head + ".linear.bias"]))
    return policy, torch.tanh(heads[0]), heads[1]


def losses(p, planes, mask, target_policy, target_value, target_ube, blocks, train_ube=True, relu_masks=None):
    """compute_loss_and_take_step up to the loss (learn/src/main.rs:384-402) -> (policy, value, ube) losses + outputs."""
    B = planes.shape[0]
    policy, value, ube = forward_t(p, planes, blocks, True, relu_masks)
    logp = policy.masked_fill(mask.view_as(policy), float(np.finfo(np.float32).min)).view(B, -1).log_softmax(1)
    loss_policy = -(logp * target_policy).sum() / B
    loss_value = (target_value.unsqueeze(1) - value).square().mean()
    tu = target_ube.unsqueeze(1).log().clamp(MINIMUM_UBE_TARGET, float(np.log(MAXIMUM_VARIANCE)))
    loss_ube = (tu - ube).square().mean() if train_ube else torch.zeros_like(loss_value)
    return (loss_policy, loss_value, loss_ube), (policy.view(B, -1), value.view(-1), ube.view(-1))


def adam(p, lr):
    return torch.optim.Adam([t for t in p.values() if t.requires_grad], lr=lr)
