"""A few tz_net_eval calls at one batch size (for rocprofv3 --kernel-trace): python tools/agent_trace.py <batch>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
mcts = A.BatchedMCTS(B, 5, 4, agent_kind=A.AGENT_DUMMY, node_capacity=256)
mcts.new_openings(np.arange(B) % 16)
mcts.simulate(np.zeros(B, np.float32), 1)
states = mcts.get_positions()
ch = mcts.root_children()
info = mcts.root_info()
acts = [ch["move_idx"][g, :info["n_children"][g]] for g in range(B)]
for _ in range(20):
    net.policy_value_uncertainty(states, acts)
