"""Average duration of the fused net kernel per 4096 positions (HIP events around every launch of a 40-simulation search on
4096 games, net5) for the precisions named on the command line.
    python tools/net_kernel_time.py f16 f16c8 f16x2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from precision_report import kernel_ms  # noqa: E402

from takzero_amd import weights as W  # noqa: E402

w0 = W.init_weights(W.ARCH_NET5, seed=123)
for p in sys.argv[1:]:
    print(p, round(kernel_ms(p, w0, 4096, 40), 3), flush=True)
