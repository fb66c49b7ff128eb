"""Simulations/s of a small lock-step search (PUCT, random-init net) at a given board size and width: the widths the reference itself runs
(128 games per process, selfplay/src/main.rs:37).  python tools/small_search_rate.py <n> <games> [sims]   (TZ_NET_SPLIT=0 for A/B)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import weights as W

n, games = int(sys.argv[1]), int(sys.argv[2])
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 400
arch, wa = {4: (A.ARCH_NET4_SIMHASH, W.ARCH_NET4_SIMHASH), 5: (A.ARCH_NET5, W.ARCH_NET5), 6: (A.ARCH_NET6_SIMHASH, W.ARCH_NET6_SIMHASH)}[n]
net = A.Net(arch=arch).load_tensors(W.init_weights(wa, seed=123))
m = A.BatchedMCTS(games, n, 4, agent=net)
m.new_openings(np.arange(games) % 16)
betas = np.zeros(games, np.float32)
m.simulate(betas, 20)
m.sync()
m.profile(reset=1)
s0, _ = m.counters()
t0 = time.perf_counter()
m.simulate(betas, sims)
m.sync()
dt = time.perf_counter() - t0
s1, _ = m.counters()
p = m.profile(reset=2)
print("%dx%d, %d games, %d simulations: %.0f simulations/s, net kernel %.3f ms per simulation (TZ_NET_SPLIT=%s)"
      % (n, n, games, sims, (s1 - s0) / dt, p["conv_ms"] / max(1, p["conv_launches"]), os.environ.get("TZ_NET_SPLIT", "default")))
