"""Where the shipped 5x5 net kernel's time goes: its ablated twins (build with `python -m takzero_amd.build --ablations`) timed
in one process on the live search, and the in-kernel shader clock (s_memtime / s_memrealtime stamps around the tower).

    python tools/net_ablation.py [precision=f16] [games=4096] [sims=80]

TZ_NET_ABL bits: 1 no activation ds_read stream, 2 no weight stream from L2, 4 operands stream but no MFMA, 8 stamps."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import weights as W  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    sims = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[prec])
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(games, 5, 4, agent=net, node_capacity=1 << 15)
    betas = np.zeros(games, np.float32)
    out = {"precision": prec, "games": games, "ms_per_launch": {}}
    names = {0: "shipped", 1: "no activation ds_reads", 2: "no weight stream", 3: "neither stream (MFMAs, barriers, epilogues)",
             4: "both streams, no MFMA", 8: "shipped + stamps"}
    for rnd in range(2):                      # two interleaved rounds in one process (rule 24)
        for abl in (0, 1, 2, 3, 4, 8):
            os.environ["TZ_NET_ABL"] = str(abl)
            mcts.new_openings(np.arange(games) % 16)      # fresh trees: the ablated kernels write garbage logits
            mcts.simulate(betas, 6)
            mcts.sync()
            mcts.profile(reset=1)
            mcts.simulate(betas, sims)
            mcts.sync()
            p = mcts.profile(reset=2)
            out["ms_per_launch"].setdefault(names[abl], []).append(round(p["conv_ms"] / max(1, p["conv_launches"]), 4))
            if abl == 8:
                mhz, us = C.c_double(), C.c_double()
                A.check(net.lib.tz_debug_net_clock(net.h, C.byref(mhz), C.byref(us)))
                out.setdefault("in_kernel_clock_mhz", []).append(round(mhz.value, 1))
                out.setdefault("tower_us_median_workgroup", []).append(round(us.value, 1))
    os.environ.pop("TZ_NET_ABL", None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
