"""Soak run of the reference's current self-play configuration at full width: Gumbel sequential halving (k = 64, budget
768), the exploration feature (half the games at beta = 0.25), 4096 games, files written through the appender."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import formats as F
from takzero_amd import runner as R
from takzero_amd import weights as W

moves = int(sys.argv[1]) if len(sys.argv) > 1 else 60
net = A.Net(arch=A.ARCH_NET5)
net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
mcts = A.BatchedMCTS(4096, 5, 4, agent=net)
d = tempfile.mkdtemp()
open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
t0 = time.perf_counter()
sp = R.run_selfplay(d, mcts, 768, moves=moves, seed=0, search="gumbel", sampled_actions=64, watch_model=False,
                    exploration=True, max_wait=60)
dt = time.perf_counter() - t0
sims, evals = mcts.counters()
used, cap = mcts.pool_usage()
print(json.dumps({"moves": moves, "s_per_move": dt / moves, "sims_per_s": sims / dt, "nn_evals": evals, "pool_used": used,
                  "pool_cap": cap, "files": {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}}))
