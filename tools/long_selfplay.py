"""Long self-play run to watch node-pool usage under tree reuse (diagnostic)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import selfplay as SP
from takzero_amd import weights as W

games, sims, moves = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
mcts = A.BatchedMCTS(games, 5, 4, agent=net)
sp = SP.SelfPlay(mcts, sims, seed=0)
peak, finished, ntargets = 0, 0, 0
t0 = time.time()
for mv in range(moves):
    t, r = sp.play_move()
    finished += len(r)
    ntargets += len(t)
    used, cap = mcts.pool_usage()
    peak = max(peak, used)
    if mv % 10 == 9:
        info = mcts.root_info()
        print("move %d: pool %d/%d (peak %d), finished games %d, targets %d, max ply %d, %.1fs  host phases %s" % (
            mv + 1, used, cap, peak, finished, ntargets, int(info["ply"].max()), time.time() - t0,
            {k: round(v, 2) for k, v in sp.host_s.items()}), flush=True)
