"""A/B of the four-wave form of the 5x5 net kernel (diagnostic build, TZ_NET_W4=1: 4 waves x 64 output channels, one wave per SIMD)
against the shipped eight-wave form: same bits out, kernel time per 4096 positions.   python tools/net_w4_ab.py [f16|bf16]"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def child(prec):
    import takzero_amd.api as A
    from precision_report import kernel_ms
    from takzero_amd import precision as P
    from takzero_amd import weights as W

    w0 = W.init_weights(W.ARCH_NET5, seed=123)
    states = P.sample_positions(5, 4, 2000, seed=3)
    net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[prec]).load_tensors(w0)
    out = net.forward_raw(states)
    net.close()
    h = hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in out)).hexdigest()[:16]
    print("W4=%s %s outputs %s kernel %.3f ms" % (os.environ.get("TZ_NET_W4", "0"), prec, h, kernel_ms(prec, w0, 4096, 40)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
        for w4 in ("0", "1", "0", "1"):
            env = dict(os.environ, TZ_NET_W4=w4)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", prec], env=env, check=True)
