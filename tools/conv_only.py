"""Runs only the shipped residual-tower conv kernel (for rocprofv3 --pmc passes)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

net = A.Net(arch=A.ARCH_TEST, n=5, blocks=1)
net.load_tensors(W.init_weights(W.ARCH_TEST, n=5, blocks=1, seed=1))
ms = C.c_float()
A.check(net.lib.tz_debug_conv_bench(net.h, 0, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 20, C.byref(ms)))
print("avg ms", ms.value)
