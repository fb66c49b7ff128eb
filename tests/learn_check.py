"""Checker script (test infrastructure): prints how far the learn step is from PyTorch autograd (outputs, losses,
per-tensor gradient error).  `python tests/learn_check.py [n] [blocks]` on a GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import learn_torch as LT  # noqa: E402
import oracle_lib as O  # noqa: E402
import takzero_amd.api as A  # noqa: E402
from takzero_amd import learn as L  # noqa: E402
from takzero_amd import weights as W  # noqa: E402
from test_gpu_learn import _batch  # noqa: E402

n, blocks, B = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 2, 64
oracle = O.load()
w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=3 + n, trained_stats=True)
tr = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B).load_tensors(w)
p = LT.make_params(w)
states, planes, policy, mask, value, ube = _batch(oracle, n, B, 100)
got = tr.step(states, policy, mask, value, ube, train_ube=True, apply=False)
want, wouts = LT.losses(p, torch.from_numpy(planes), torch.from_numpy(mask.astype(bool)), torch.from_numpy(policy),
                        torch.from_numpy(value), torch.from_numpy(ube), blocks, True)
(want[0] + want[1] + want[2]).backward()
for name, g, t_ in zip(("policy", "value", "ube"), tr.outputs(), wouts):
    t_ = t_.detach().numpy()
    print("out %-8s max|x| %.3f  max diff %.3e" % (name, np.abs(t_).max(), np.abs(g - t_).max()))
print("losses", got, [float(x.detach()) for x in want])
# the same graph in float64: which of the two fp32 computations is closer to the exact gradient?
p64 = {k: (v.detach().double().requires_grad_(v.requires_grad)) for k, v in LT.make_params(w).items()}
w64, _ = LT.losses(p64, torch.from_numpy(planes).double(), torch.from_numpy(mask.astype(bool)), torch.from_numpy(policy).double(),
                   torch.from_numpy(value).double(), torch.from_numpy(ube).double(), blocks, True)
(w64[0] + w64[1] + w64[2]).backward()
worst = []
for k in tr.names:
    if "running_" in k:
        continue
    g, tg, g64 = tr.tensor(k, L.GRAD), p[k].grad.numpy(), p64[k].grad.numpy()
    scale = np.abs(g64).max() + 1e-30
    worst.append((float(np.abs(g - tg.reshape(g.shape)).max() / scale), float(np.abs(g - g64.reshape(g.shape)).max() / scale),
                  float(np.abs(tg - g64).max() / scale), k))
print("relative to the float64 gradient's largest entry:  hip-vs-torch32   hip-vs-f64   torch32-vs-f64")
for a, b, c, k in sorted(worst, reverse=True)[:8]:
    print("  %.3e   %.3e   %.3e   %s" % (a, b, c, k))

# where does the worst tensor differ?  A ReLU whose input is ~0 can land on either side of zero in two fp32
# computations; that changes the gradient of exactly one output channel of the layer below it.
a, b, c, k = sorted(worst, reverse=True)[0]
g, g64 = tr.tensor(k, L.GRAD), p64[k].grad.numpy()
bad = np.abs(g - g64.reshape(g.shape)) > 1e-3 * np.abs(g64).max()
if g.ndim == 4:
    print("%s: %d entries off by more than 1e-3 of the largest, in output channels %s" % (k, int(bad.sum()), sorted(set(np.nonzero(bad)[0].tolist()))[:10]))
