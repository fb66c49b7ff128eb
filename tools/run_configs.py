"""Runs BASELINE.json configs 4 and 5 at (near) full size on one MI355X and prints one JSON line each.
(config 2 is bench.py; configs 1/4/5 at test size are parity cases in tests/.)"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import formats as F
from takzero_amd import reanalyze as RA
from takzero_amd import selfplay as SP
from takzero_amd import weights as W


def config4(moves=2, games=2048, sims=800):
    net = A.Net(arch=A.ARCH_NET6_SIMHASH)
    net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH, seed=123))
    mcts = A.BatchedMCTS(games, 6, 4, agent=net)
    sp = SP.SelfPlay(mcts, sims, seed=0)
    sp.play_move()  # warm-up
    mcts.sync()
    s0, e0 = mcts.counters()
    t0 = time.perf_counter()
    for _ in range(moves):
        sp.play_move()
    mcts.sync()
    dt = time.perf_counter() - t0
    s1, e1 = mcts.counters()
    out = {"config": "6x6 Tak, %d games, %d sims/move, net6_simhash (BASELINE configs[3])" % (games, sims),
           "sims_per_s": (s1 - s0) / dt, "nn_leaf_evals_per_s": (e1 - e0) / dt, "positions_per_s": games * moves / dt,
           "net_flops_frac_of_bf16_peak": (e1 - e0) / dt * 1.4067e9 / 2.5e15}
    mcts.close()
    net.close()
    return out


def config5(games=4096, sims=1600, selfplay_moves=40, iterations=1):
    net = A.Net(arch=A.ARCH_NET5)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(games, 5, 4, agent=net)
    # synthetic replays.txt from the engine's own quick self-play (8 sims/move), SURVEY.md §8d config 5
    sp = SP.SelfPlay(mcts, 8, seed=1)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "replays.txt")
    with open(path, "w") as f:
        for _ in range(selfplay_moves):
            _t, replays = sp.play_move()
            for start, acts, result in replays:
                f.write(F.format_replay(5, start, acts, result))
    re = RA.Reanalyze(mcts, sims, seed=0)
    t0 = time.perf_counter()
    npos = re.buffer.read_new(path)
    t_feed = time.perf_counter() - t0
    if npos < games:
        return {"config": "reanalyze", "error": "only %d positions in the synthetic replay file" % npos}
    mcts.sync()
    s0, e0 = mcts.counters()
    t0 = time.perf_counter()
    ntargets = 0
    for _ in range(iterations):
        ntargets += len(re.iterate())
    mcts.sync()
    dt = time.perf_counter() - t0
    s1, e1 = mcts.counters()
    out = {"config": "5x5 reanalyze, %d positions/iteration, %d sims/position, net5 (BASELINE configs[4], one GPU's shard)" % (games, sims),
           "replay_positions_in_buffer": npos, "feed_positions_per_s": npos / t_feed, "sims_per_s": (s1 - s0) / dt,
           "targets_per_s": ntargets / dt}
    mcts.close()
    net.close()
    return out


def config5_native(games=4096, sims=1600, selfplay_moves=40, iterations=1):
    """config 5 through the native drivers (csrc/tz_host.cpp): self-play writes replays.txt, reanalyze tails it."""
    net = A.Net(arch=A.ARCH_NET5)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(games, 5, 4, agent=net)
    sp = SP.NativeSelfPlay(mcts, 8, seed=1, search="puct")
    d = tempfile.mkdtemp()
    path = os.path.join(d, "replays.txt")
    with open(path, "wb") as f:
        for _ in range(selfplay_moves):
            sp.play_move()
            f.write(sp.take_text(1))
    re = RA.NativeReanalyze(mcts, sims, seed=0, search="puct")
    t0 = time.perf_counter()
    npos = re.feed(path)
    t_feed = time.perf_counter() - t0
    if npos < games:
        return {"config": "reanalyze (native)", "error": "only %d positions in the synthetic replay file" % npos}
    mcts.sync()
    s0, _ = mcts.counters()
    t0 = time.perf_counter()
    nt = 0
    for _ in range(iterations):
        re.iterate()
        nt += re.take_text().count(b"\n")
    mcts.sync()
    dt = time.perf_counter() - t0
    s1, _ = mcts.counters()
    return {"config": "5x5 reanalyze through the native driver, %d positions/iteration, %d sims/position, net5" % (games, sims),
            "replay_positions_in_buffer": npos, "feed_positions_per_s": npos / t_feed, "sims_per_s": (s1 - s0) / dt,
            "targets_per_s": nt / dt}


def gumbel(moves=3, games=4096, budget=768, k=64):
    """What the reference's selfplay binary runs today (selfplay/src/main.rs:138-153): Gumbel sequential halving,
    64 sampled actions, budget 768, on 5x5 / net5."""
    net = A.Net(arch=A.ARCH_NET5)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(games, 5, 4, agent=net)
    sp = SP.SelfPlay(mcts, budget, seed=0, search="gumbel", sampled_actions=k)
    sp.play_move()
    mcts.sync()
    s0, e0 = mcts.counters()
    t0 = time.perf_counter()
    for _ in range(moves):
        sp.play_move()
    mcts.sync()
    dt = time.perf_counter() - t0
    s1, e1 = mcts.counters()
    out = {"config": "5x5 Tak, %d games, Gumbel sequential halving k=%d budget=%d, net5" % (games, k, budget),
           "sims_per_s": (s1 - s0) / dt, "nn_leaf_evals_per_s": (e1 - e0) / dt, "positions_per_s": games * moves / dt,
           "s_per_move": dt / moves}
    mcts.close()
    net.close()
    return out


def directory_loop(moves=100, games=4096, sims=400):
    """selfplay::main end to end (runner.run_selfplay: back-pressure file, target / replay files written by the
    appender thread) against the bare search loop of bench.py, over enough moves that games finish and targets flow."""
    from takzero_amd import runner as R

    net = A.Net(arch=A.ARCH_NET5)
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(games, 5, 4, agent=net)
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "buffer_lengths.txt"), "w") as f:
        f.write(F.format_buffer_lengths(0, 0))
    t0 = time.perf_counter()
    sp = R.run_selfplay(d, mcts, sims, moves=moves, seed=0, search="puct", watch_model=False, max_wait=60)
    dt = time.perf_counter() - t0
    sims_total, _ = mcts.counters()
    tpath = os.path.join(d, "targets-selfplay.txt")
    out = {"config": "run_selfplay on a directory: 5x5, %d games, %d sims/move, %d moves" % (games, sims, moves),
           "s_per_move": dt / moves, "sims_per_s": sims_total / dt, "search_s": sp.host_s["search"],
           "record_s": sp.host_s["record"], "complete_s": sp.host_s["complete"], "wall_s": dt,
           "targets_written": sum(1 for _ in open(tpath)) if os.path.exists(tpath) else 0,
           "targets_file_mb": os.path.getsize(tpath) / 1e6 if os.path.exists(tpath) else 0}
    mcts.close()
    net.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["4", "5"]
    if "5n" in which:
        print(json.dumps(config5_native()), flush=True)
    if "dir" in which:
        print(json.dumps(directory_loop()), flush=True)
    if "gumbel" in which:
        print(json.dumps(gumbel()), flush=True)
    if "4" in which:
        print(json.dumps(config4()), flush=True)
    if "5" in which:
        print(json.dumps(config5()), flush=True)
