"""Runs the net5 forward (fused tower) a few times on 4096 positions (for rocprofv3 --pmc passes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

net = A.Net(arch=A.ARCH_NET5)  # default TZ_TOWER=2: fused net kernel
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
mcts = A.BatchedMCTS(4096, 5, 4, agent=net, node_capacity=2048)
mcts.new_openings(np.arange(4096) % 16)
mcts.simulate(np.zeros(4096, np.float32), int(sys.argv[1]) if len(sys.argv) > 1 else 6)
print(mcts.counters())
