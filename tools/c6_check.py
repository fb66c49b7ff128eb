"""TZ_PREC_F16C6 against the library's fp32 path at trained logit scale (net5), next to TZ_PREC_F16C8 and TZ_PREC_F16, and the net
kernel's time per 4096 positions.     python tools/c6_check.py [positions=64] [time=1]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import precision as P  # noqa: E402
from takzero_amd import weights as W  # noqa: E402
from precision_report import kernel_ms  # noqa: E402

npos = int(sys.argv[1]) if len(sys.argv) > 1 else 64
do_time = int(sys.argv[2]) if len(sys.argv) > 2 else 1
precs = tuple(os.environ.get("C6_PRECS", "f16c6,f16c8,f16").split(","))
states = P.sample_positions(5, 4, npos, seed=7)
w0 = W.init_weights(W.ARCH_NET5, seed=123)
w1 = P.trained_scale_weights(A.ARCH_NET5, states, seed=123)
out = {"positions": npos}
out["random_init_scale"] = P.errors_against_f32(A.ARCH_NET5, w0, states, precisions=precs)
out["trained_scale"] = P.errors_against_f32(A.ARCH_NET5, w1, states, precisions=precs)
print(json.dumps(out, indent=1), flush=True)
if do_time:
    print(json.dumps({p: round(kernel_ms(p, w0, 4096, 40), 3) for p in precs}), flush=True)
