"""Net kernel time per 4096 positions for the precisions in C6_PRECS (default f16c6), under whatever TZ_C6_FLAGS says."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from precision_report import kernel_ms
from takzero_amd import weights as W
w0 = W.init_weights(W.ARCH_NET5, seed=123)
for p in os.environ.get("C6_PRECS", "f16c6").split(","):
    print("flags", os.environ.get("TZ_C6_FLAGS", "0"), p, round(kernel_ms(p, w0, 4096, 30), 3), flush=True)
