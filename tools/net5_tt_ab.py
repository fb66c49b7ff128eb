"""A/B (ablation build): the shipped 5x5 net kernel against the same kernel on the compact tap table + fragment ring (TZ_NET_TT=1),
interleaved rounds in one process, outputs compared."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import precision as P  # noqa: E402
from takzero_amd import weights as W  # noqa: E402

net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
states = P.sample_positions(5, 4, 4096, seed=3)
os.environ["TZ_NET_TT"] = "0"
ref = net.forward_raw(states)
os.environ["TZ_NET_TT"] = "1"
alt = net.forward_raw(states)
same = all(np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)) for x, y in zip(ref, alt))
mcts = A.BatchedMCTS(4096, 5, 4, agent=net, node_capacity=1 << 15)
betas = np.zeros(4096, np.float32)
out = {"bit_identical": bool(same), "ms_per_launch": {"shipped": [], "ring": []}}
for rnd in range(3):
    for name, v in (("shipped", "0"), ("ring", "1")):
        os.environ["TZ_NET_TT"] = v
        mcts.new_openings(np.arange(4096) % 16)
        mcts.simulate(betas, 6)
        mcts.sync()
        mcts.profile(reset=1)
        mcts.simulate(betas, 80)
        mcts.sync()
        p = mcts.profile(reset=2)
        out["ms_per_launch"][name].append(round(p["conv_ms"] / max(1, p["conv_launches"]), 4))
print(json.dumps(out))
