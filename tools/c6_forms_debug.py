import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import precision as P, weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prec = A.PREC_NAMES[sys.argv[2]] if len(sys.argv) > 2 else A.PREC_F16C6
arch = A.ARCH_NET6_SIMHASH if n == 6 else A.ARCH_NET5
net = A.Net(arch=arch, precision=prec)
net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH if n == 6 else W.ARCH_NET5, seed=9))
base = P.sample_positions(n, 4, 64, seed=3)
states = np.concatenate([base] * 18)[:1100]
big = net.forward_raw(states)
for count in ((1100, 1027, 1026, 1025, 1028) if n == 6 else (1024, 1023, 1022, 1021, 511, 510)):
    ref = net.forward_raw(states[:((count + 7) // 8) * 8 if count < 1024 or n == 5 else 1100])
    part = net.forward_raw(states[:count])
    for name, x, y in zip(("pol", "val", "ube"), ref, part):
        d = np.abs(x[:count].astype(np.float64) - y.astype(np.float64))
        bad = np.argwhere(d.reshape(count, -1).max(axis=1) > 0).ravel()
        if name == "pol" and len(bad):
            j = np.argwhere(d[bad[0]].ravel() > 0).ravel()
            print("   first bad position", bad[0], "entries", len(j), j[:10], "of", d[bad[0]].size)
        print(count, name, "max diff", d.max(), "bad positions", len(bad), bad[:12], flush=True)
