import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import takzero_amd.api as A
from takzero_amd import precision as P, weights as W
states = P.sample_positions(5, 4, 64, seed=7)
for blocks, arch, trained in ((1, A.ARCH_TEST, False), (2, A.ARCH_TEST, False), (0, A.ARCH_NET5, False)):
    w = W.init_weights(arch if arch != A.ARCH_TEST else W.ARCH_TEST, n=5, blocks=blocks, seed=123, trained_stats=trained)
    outs = {}
    for prec in ("f32", "f16c6"):
        net = A.Net(arch=arch, n=5, precision=A.PREC_NAMES[prec], blocks=blocks)
        net.load_tensors(w)
        outs[prec] = net.forward_raw(states)
        net.close()
    p0, v0, u0 = outs["f32"]; p1, v1, u1 = outs["f16c6"]
    d = np.abs(p1 - p0)
    i = np.unravel_index(np.argmax(d), d.shape)
    print("blocks", blocks, "arch", arch, "pol shape", p0.shape, "max err", d.max(), "at", i, "ref", p0[i], "got", p1[i], "scale", np.abs(p0).max(),
          "val err", np.abs(v1 - v0).max(), "ube err", np.abs(u1 - u0).max(), "nan", np.isnan(p1).sum(), flush=True)
    bad = np.argwhere(d > 1e-2)
    print(" bad entries", len(bad), "of", d.size, "positions", np.unique(bad[:, 0])[:20] if len(bad) else "", "channels", np.unique(bad[:, -1])[:40] if len(bad) else "", flush=True)
