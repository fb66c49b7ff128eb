"""Where a tower layer's time goes inside the net kernel (diagnostic build: `python -m takzero_amd.build --ablations`): wave 0 of
every workgroup stamps s_memtime after the barrier that opens the middle layer, after its k-loop, after the barrier that
follows, and after the epilogue.  Medians over the workgroups, in shader-clock cycles of s_memtime.
    TZ_PRECISION=f16c8 python tools/net_phases.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TZ_NET_ABL"] = os.environ.get("TZ_NET_ABL", "16")
import takzero_amd.api as A  # noqa: E402
from takzero_amd import weights as W  # noqa: E402


def main():
    out = {}
    for prec in sys.argv[1:] or ["f16", "f16c8"]:
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[prec])
        net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
        mcts = A.BatchedMCTS(4096, 5, 4, agent=net, node_capacity=2048)
        mcts.new_openings(np.arange(4096) % 16)
        mcts.simulate(np.zeros(4096, np.float32), 6)
        mcts.sync()
        buf = (C.c_uint64 * (4 * 2048))()
        groups = C.c_int()
        A.check(net.lib.tz_debug_net_stamps(net.h, buf, 2048, C.byref(groups)))
        st = np.frombuffer(buf, np.uint64).reshape(-1, 4)[:groups.value].astype(np.int64)
        st = st[(st[:, 3] > st[:, 0])]
        d = {"k_loop": np.median(st[:, 1] - st[:, 0]), "barrier_after_loop": np.median(st[:, 2] - st[:, 1]),
             "epilogue_and_barrier": np.median(st[:, 3] - st[:, 2]), "layer": np.median(st[:, 3] - st[:, 0]), "workgroups": int(len(st))}
        out[prec] = {k: float(v) for k, v in d.items()}
        mcts.close()
        net.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
