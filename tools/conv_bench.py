"""Times the residual-tower conv kernel and its ablation variants (diagnostic, GPU box only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

net = A.Net(arch=A.ARCH_TEST, n=5, blocks=1)
net.load_tensors(W.init_weights(W.ARCH_TEST, n=5, blocks=1, seed=1))
pos = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flop = 2 * 25 * 256 * 2304 * pos
names = {0: "8w x 32ch, 8 boards", 1: "  no LDS frag reads", 2: "  no weight loads", 3: "  MFMA only", 4: "  1 tap of 9",
         20: "8w, padded row-major LDS", 21: "  no LDS frag reads", 23: "  MFMA only",
         30: "4w x 64ch, 8 boards, 1 wave/SIMD", 31: "  no LDS frag reads", 33: "  MFMA only",
         10: "4w x 64ch, 4 boards", 11: "  no LDS frag reads", 12: "  no weight loads", 13: "  MFMA only", 14: "  1 tap of 9"}
for rnd in range(3):
    for v in (0, 30, 31, 33, 0, 30, 3):
        ms = C.c_float()
        A.check(net.lib.tz_debug_conv_bench(net.h, v, pos, 50, C.byref(ms)))
        print("round %d variant %d (%s): %.1f us  %.0f TFLOP/s" % (rnd, v, names[v], ms.value * 1e3, flop / ms.value / 1e9))
