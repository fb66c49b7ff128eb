"""Experiment for the gap between a simulation's wall time and its net kernel (tree kernels, RND launches, launch gaps and the
tail of the net kernel's last workgroup round): the 4096 games as G searches of 4096 / G games, each with its own net object, stream
and driver thread, so that one group's tree / RND kernels run while another group's net kernel holds the CUs.
    python tools/two_group_overlap.py [precision] [moves] [groups ...]
Prints simulations/s of the whole job for each group count (same total games, same simulations per move)."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import selfplay as SP
from takzero_amd import weights as W

GAMES, SIMS = 4096, 1600


def run(precision, groups, moves, warmup=1):
    per = GAMES // groups
    w = W.init_weights(W.ARCH_NET5, seed=123)
    nets, searches, players = [], [], []
    for g in range(groups):
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[precision])
        net.load_tensors(w)
        m = A.BatchedMCTS(per, 5, 4, agent=net)
        nets.append(net)
        searches.append(m)
        players.append(SP.NativeSelfPlay(m, SIMS, seed=g, shard=g, search="puct", sampled_actions=64))

    def play(g, n):
        for _ in range(n):
            players[g].play_move()
        searches[g].sync()

    def all_groups(n):
        ts = [threading.Thread(target=play, args=(g, n)) for g in range(groups)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    all_groups(warmup)
    for m in searches:
        m.profile(reset=1)
    s0 = sum(m.counters()[0] for m in searches)
    t0 = time.perf_counter()
    all_groups(moves)
    dt = time.perf_counter() - t0
    s1 = sum(m.counters()[0] for m in searches)
    profs = [m.profile(reset=2) for m in searches]
    out = {"groups": groups, "games_per_group": per, "moves": moves, "sims_per_s": (s1 - s0) / dt,
           "net_kernel_ms_per_launch": [round(p["conv_ms"] / max(1, p["conv_launches"]), 4) for p in profs],
           "wall_ms_per_simulation_of_all_games": 1000.0 * dt / ((s1 - s0) / GAMES)}
    for p in players:
        p.close()
    for m in searches:
        m.close()
    for n in nets:
        n.close()
    return out


if __name__ == "__main__":
    precision = sys.argv[1] if len(sys.argv) > 1 else "f16"
    moves = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    counts = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
    for g in counts:
        print(json.dumps(run(precision, g, moves)), flush=True)
