"""The README quick-start snippet at a small size (checks that the documented calls still exist and run)."""
import sys, tempfile
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, takzero_amd.api as tz
from takzero_amd import weights, selfplay
net = tz.Net(arch=tz.ARCH_NET5).load_tensors(weights.init_weights(weights.ARCH_NET5, seed=123))
mcts = tz.BatchedMCTS(256, 5, 4, agent=net)
sp = selfplay.NativeSelfPlay(mcts, 64, search="puct")
sp.play_move(); lines = sp.take_text(0)
sp = selfplay.SelfPlay(mcts, sims_per_move=32)
targets, replays = sp.play_move()
print("readme snippet ok", net.precision, len(lines), len(targets))
