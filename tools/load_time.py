"""Time of a model (re)load per arithmetic: net5 variables in host memory -> BatchNorm folding, fragment order, fp16 / FP8 conversion,
upload (what a hot reload of model_latest.ot costs the self-play process after the archive is parsed).   python tools/load_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import weights as W  # noqa: E402

w = W.init_weights(W.ARCH_NET5, seed=1)
for p in ("f16", "f16x2", "f16c8"):
    net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[p])
    net.load_tensors(w)
    t0 = time.perf_counter()
    for _ in range(3):
        net.load_tensors(w)
    print(p, "load_tensors %.3f s" % ((time.perf_counter() - t0) / 3), flush=True)
    net.close()
