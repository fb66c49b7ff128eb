"""Golden vectors for the network forward ("pinned instead by fixtures generated here").

The reference executes its nets through LibTorch; the PyTorch in this image is the same ATen code.  This script
evaluates oracle/nets_torch.py (fp32, CPU) on fixed positions with weights that takzero_amd.weights regenerates from a
seed, and commits inputs + outputs:  python tests/golden/make_net_fixture.py  ->  tests/golden/net_forward.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import nets_torch as T  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_util import random_positions  # noqa: E402
from takzero_amd import weights as W  # noqa: E402

oracle = O.load()
cases = []
for arch, n, blocks, seed, count in ((W.ARCH_TEST, 5, 2, 11, 6), (W.ARCH_TEST, 4, 1, 12, 5), (W.ARCH_TEST, 6, 1, 13, 4),
                                     (W.ARCH_TEST, 3, 1, 14, 4)):
    w = W.init_weights(arch, n=n, blocks=blocks, seed=seed, trained_stats=True)
    states = random_positions(oracle, O, n, 4, count, seed, min_ply=2, max_ply=36)
    planes = np.stack([O.game_repr(oracle, s) for s in states]).reshape(count, -1, n, n)
    pol, val, ube = T.forward(w, planes, blocks)
    pol = pol.reshape(count, -1).numpy()
    positions = []
    for i, s in enumerate(states):
        legal = O.possible_moves(oracle, s)
        positions.append({"tps": O.to_tps(oracle, s), "legal": [int(m) for m in legal],
                          "logits": [float(np.float32(x)) for x in pol[i, legal]],
                          "value": float(val[i]), "ube": float(ube[i])})
    cases.append({"arch": int(arch), "n": n, "blocks": blocks, "seed": seed, "half_komi": 4, "trained_stats": True,
                  "weights_checksum": float(sum(float(np.abs(v).sum(dtype=np.float64)) for v in w.values())),
                  "positions": positions})
with open(os.path.join(HERE, "net_forward.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_net_fixture.py", "cases": cases}, f, indent=0)
print("wrote", sum(len(c["positions"]) for c in cases), "positions")
