"""Fixed cost of the fused net kernel outside the residual tower (game_repr, first conv, heads, policy conv, launch):
time per launch at 4096 positions for 1, 3, 5, 9 residual blocks of the 256-filter test architecture; the intercept of
the straight line is what does not scale with depth."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

B = 4096
pts = []
for blocks in (1, 3, 5, 9):
    net = A.Net(arch=A.ARCH_TEST, n=5, blocks=blocks).load_tensors(W.init_weights(W.ARCH_TEST, n=5, blocks=blocks, seed=1))
    mcts = A.BatchedMCTS(B, 5, 4, agent=net, node_capacity=1 << 11)
    mcts.new_openings(np.arange(B) % 16)
    mcts.simulate(np.zeros(B, np.float32), 8)
    mcts.profile(reset=1)
    mcts.simulate(np.zeros(B, np.float32), 40)
    p = mcts.profile()
    ms = p["conv_ms"] / max(1, p["conv_launches"])
    pts.append((blocks, ms))
    print("blocks %d: %.4f ms per launch (%d launches timed)" % (blocks, ms, p["conv_launches"]), flush=True)
    mcts.close()
    net.close()
x = np.array([2 * b for b, _ in pts], float)
y = np.array([m for _, m in pts])
slope, icpt = np.polyfit(x, y, 1)
print("per tower conv %.4f ms, fixed %.4f ms per launch (%.1f %% of a 40-conv launch)" % (slope, icpt, 100 * icpt / (icpt + 40 * slope)))
