"""Errors of the MFMA precisions against the fp32 path at random-init and at trained logit scale (net5, 5x5), and the net
kernel's time per 4096 positions in each precision.  Prints one JSON object (kept as profiles/r02_precision.json).

    python tools/precision_report.py [positions=64] [games=4096] [sims=60]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import precision as P  # noqa: E402
from takzero_amd import weights as W  # noqa: E402


def kernel_ms(prec, weights, games, sims):
    net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[prec])
    net.load_tensors(weights)
    mcts = A.BatchedMCTS(games, 5, 4, agent=net)
    mcts.new_openings(np.arange(games) % 16)
    betas = np.zeros(games, np.float32)
    mcts.simulate(betas, 10)
    mcts.sync()
    mcts.profile(reset=1)
    mcts.simulate(betas, sims)
    mcts.sync()
    prof = mcts.profile(reset=2)
    mcts.close()
    net.close()
    return prof["conv_ms"] / max(1, prof["conv_launches"])


def main():
    npos = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    sims = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    states = P.sample_positions(5, 4, npos, seed=7)
    out = {"net": "net5 (5x5, 20 blocks)", "positions": npos}
    w0 = W.init_weights(W.ARCH_NET5, seed=123)
    out["random_init_scale"] = P.errors_against_f32(A.ARCH_NET5, w0, states)
    w1 = P.trained_scale_weights(A.ARCH_NET5, states, seed=123)
    out["trained_scale"] = P.errors_against_f32(A.ARCH_NET5, w1, states)
    out["net_kernel_ms_per_%d_positions" % games] = {p: kernel_ms(p, w0, games, sims) for p in ("f16", "bf16", "f16c8", "f16x2")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
