"""How large do the trees get when the policy is sharp?  (Subtree reuse keeps the chosen child's subtree; with a sharp
prior that is most of the tree, move after move.)  Random net5 with its policy logits scaled by `sharpness`, Gumbel 768 /
k 64 on `games` games: prints the fullest pool, and how many expansions were skipped because a pool was full."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import selfplay as SP
from takzero_amd import weights as W

sharp = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
games = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 120
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w = W.init_weights(W.ARCH_NET5, seed=123)
w["policy.conv2d.weight"] = w["policy.conv2d.weight"] * sharp
w["policy.conv2d.bias"] = w["policy.conv2d.bias"] * sharp
net = A.Net(arch=A.ARCH_NET5).load_tensors(w)
mcts = A.BatchedMCTS(games, 5, 4, agent=net, node_capacity=cap)
sp = SP.SelfPlay(mcts, 768, seed=1, search="gumbel", collect_targets=False)
peak = before = 0
real_step = mcts.step


def step_and_measure(actions):   # the pools are fullest right before `step` compacts the kept subtrees
    global before
    before = mcts.pool_usage()[0]
    real_step(actions)


mcts.step = step_and_measure
for mv in range(moves):
    sp.play_move()
    after, capacity = mcts.pool_usage()
    peak = max(peak, before)
    if mv % 10 == 9:
        print("move %d: fullest pool before step %d, after %d, of %d (peak %d), skipped expansions %d" % (
            mv + 1, before, after, capacity, peak, mcts.pool_overflows()), flush=True)
