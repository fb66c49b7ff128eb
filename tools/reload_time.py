"""Time of tz_net_load_weights on a net5 LibTorch archive (what a hot reload of model_latest.ot costs the self-play process): archive
parse + weight preparation + swap.   python tools/reload_time.py"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    src = A.Net.new(arch=A.ARCH_NET5, seed=3)
    path = os.path.join(d, "model_latest.ot")
    src.save(path)
    print("archive", os.path.getsize(path) >> 20, "MiB")
    for p in ("f16", "f16c8"):
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[p]).load(path)
        t0 = time.perf_counter()
        for _ in range(3):
            net.load(path)
        print(p, "tz_net_load_weights %.3f s" % ((time.perf_counter() - t0) / 3), flush=True)
        net.close()
