"""In-process A/B of fused-tower kernel variants (diagnostic, GPU box only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
variants = [int(v) for v in sys.argv[1:]] or [0, 1]
flop = 40 * 2 * 25 * 256 * 2304 * 4096
for rnd in range(4):
    for v in variants:
        ms = C.c_float()
        A.check(net.lib.tz_debug_tower_bench(net.h, v, 4096, 10, C.byref(ms)))
        print("round %d variant %d: %.3f ms  %.0f TFLOP/s" % (rnd, v, ms.value, flop / ms.value / 1e9), flush=True)
