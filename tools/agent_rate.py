"""PCIe-inclusive rate of the Agent surface (tz_net_eval: host states in, host logits/value/variance out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import takzero_amd.api as A
from takzero_amd import weights as W

net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
mcts = A.BatchedMCTS(4096, 5, 4, agent=net, node_capacity=1024)
mcts.new_openings(np.arange(4096) % 16)
mcts.simulate(np.zeros(4096, np.float32), 1)
states = mcts.get_positions()
ch = mcts.root_children()
info = mcts.root_info()
acts = [ch["move_idx"][g, :info["n_children"][g]] for g in range(4096)]
for B in (1, 8, 32, 128, 256, 512, 1024, 4096):
    net.policy_value_uncertainty(states[:B], acts[:B])
    t0 = time.perf_counter()
    n = 30 if B <= 1024 else 10
    for _ in range(n):
        net.policy_value_uncertainty(states[:B], acts[:B])
    dt = (time.perf_counter() - t0) / n
    print("tz_net_eval batch %d: %.2f ms per call, %.0f positions/s (host buffers in and out, PCIe included)" % (B, dt * 1e3, B / dt))
    # the same call without the Python wrapper's packing of the legal-move lists (what a C / Rust host pays): arrays built once
    st = A._states(states[:B])
    amax = max(1, max(len(a) for a in acts[:B]))
    idx = np.zeros((B, amax), np.uint16)
    cnt = np.zeros(B, np.int32)
    for i, a in enumerate(acts[:B]):
        cnt[i] = len(a)
        idx[i, :len(a)] = a
    logits, value, var = np.zeros((B, amax), np.float32), np.zeros(B, np.float32), np.zeros(B, np.float32)
    call = lambda: A.check(net.lib.tz_net_eval(net.h, B, st.ctypes.data, idx.ctypes.data, cnt.ctypes.data, amax, logits.ctypes.data, value.ctypes.data, var.ctypes.data))
    call()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    dt = (time.perf_counter() - t0) / n
    print("    the C ABI call alone (arrays packed once): %.3f ms per call, %.0f positions/s" % (dt * 1e3, B / dt))
