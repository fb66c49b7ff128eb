"""Measured error of the MFMA precisions against the library's own fp32 path (TZ_PREC_F32, plain FMA kernels, itself held to
1e-6 of the PyTorch graph by tests/test_gpu_net.py) at a trained net's output scale.

The reference computes the forward in fp32 (net5.rs:184-191,237-238) and the north star asks for logits within 1e-3 of
it.  A random-init net emits |logit| ~ 0.2, a trained net5 5-10, so an absolute tolerance has to be shown on weights at
that scale: `trained_scale_weights` brings a random-init net there (BatchNorm statistics of a trained net, heads rescaled
until the fp32 path emits max |logit| = 8, value pre-activation 1.5, |ube| 2 on the sample positions).
Used by bench.py (its line carries the figures next to each precision's throughput) and tools/precision_report.py."""
import numpy as np

from . import api as A
from . import weights as W

TRAINED_LOGIT = 8.0


def sample_positions(n, half_komi, count, seed=0, plies=12):
    """`count` positions `plies` uniformly random moves after the openings (played on the GPU through the Dummy agent)."""
    from .selfplay import NativeSelfPlay

    mcts = A.BatchedMCTS(count, n, half_komi, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    sp = NativeSelfPlay(mcts, 0, seed=seed, search="random")
    for _ in range(plies):
        sp.play_move()
    states = mcts.get_positions().copy()
    sp.close()
    mcts.close()
    return states


def trained_scale_weights(arch, states, seed=123, n=0, blocks=0, device=0):
    """Random-init weights with trained-like BatchNorm statistics and the heads rescaled to a trained net's output scale,
    measured on `states` with the fp32 path."""
    w = W.init_weights(arch, n=n, blocks=blocks, seed=seed, trained_stats=True)
    net = A.Net(arch=arch, n=n, device=device, precision=A.PREC_F32, blocks=blocks)
    net.load_tensors(w)
    pol, val, ube = net.forward_raw(states)
    net.close()
    pre = np.arctanh(np.clip(val.astype(np.float64), -0.999999, 0.999999))
    return W.rescale_heads(w, TRAINED_LOGIT / float(np.abs(pol).max()), 1.5 / max(1e-6, float(np.abs(pre).max())),
                           2.0 / max(1e-6, float(np.abs(ube).max())))


def legal_mask(states, n, half_komi=4):
    """[len(states)][policy_size] bool: the legal moves of every position (the children a root gets from the engine's own move
    generation) - the logits a search ever reads (net5.rs:239-267 gathers exactly these)."""
    m = A.BatchedMCTS(len(states), n, half_komi, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    m.set_positions(np.arange(len(states)), states)
    m.simulate(np.zeros(len(states), np.float32), 1)
    info = m.root_info()
    ch = m.root_children(int(max(1, info["n_children"].max())))
    mask = np.zeros((len(states), A.policy_size(n)), bool)
    for g in range(len(states)):
        mask[g, ch["move_idx"][g, :int(info["n_children"][g])]] = True
    m.close()
    return mask


def errors_against_f32(arch, weights, states, precisions=("f16", "f16c8", "f16x2", "bf16"), n=0, blocks=0, device=0, legal=None):
    """{precision: {max_abs_logit_err, max_abs_value_err, max_abs_ube_err}} + the fp32 path's output scale.  With `legal` (a mask
    from legal_mask) also the same over the legal-move logits only: a trained net's illegal-move logits are masked out of its loss
    and drift far from the scale of the ones a search reads."""
    ref = A.Net(arch=arch, n=n, device=device, precision=A.PREC_F32, blocks=blocks)
    ref.load_tensors(weights)
    pol0, val0, ube0 = ref.forward_raw(states)
    ref.close()
    out = {"reference": "TZ_PREC_F32 (fp32 FMA kernels of this library)", "positions": int(len(states)),
           "logit_scale": float(np.abs(pol0).max()), "value_scale": float(np.abs(val0).max()), "ube_scale": float(np.abs(ube0).max())}
    if legal is not None:
        out["legal_logit_scale"] = float(np.abs(pol0[legal]).max())
        out["legal_logit_std"] = float(pol0[legal].std())
    for name in precisions:
        net = A.Net(arch=arch, n=n, device=device, precision=A.PREC_NAMES[name], blocks=blocks)
        net.load_tensors(weights)
        pol, val, ube = net.forward_raw(states)
        net.close()
        out[name] = {"max_abs_logit_err": float(np.abs(pol - pol0).max()), "max_abs_value_err": float(np.abs(val - val0).max()),
                     "max_abs_ube_err": float(np.abs(ube - ube0).max())}
        if legal is not None:
            out[name]["max_abs_legal_logit_err"] = float(np.abs(pol - pol0)[legal].max())
    return out
