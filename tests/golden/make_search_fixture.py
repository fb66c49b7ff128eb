"""Golden vectors for the search: root statistics after a fixed number of lock-step simulations with the reference's
two network-free agents (Dummy: uniform logits, value 0; Simple: piece-type prior, flat-count value; agent.rs:16-87),
computed by the CPU oracle (oracle/mcts.hpp) and committed:  python tests/golden/make_search_fixture.py
-> tests/golden/search_roots.json.  Everything is integer or f32 bit pattern, so the comparison is exact."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
from gpu_util import random_positions  # noqa: E402

oracle = O.load()
cases = []
for n, hk, agent, sims, beta, count, seed in ((3, 0, 1, 300, 1.0, 6, 1), (4, 4, 2, 200, 0.5, 6, 2), (5, 4, 2, 160, 0.0, 6, 3),
                                              (5, 4, 1, 120, 0.25, 4, 4), (6, 4, 2, 80, 0.0, 4, 5)):
    states = random_positions(oracle, O, n, hk, count, seed, min_ply=2, max_ply=24)
    s = O.OracleSearch(oracle, count, n, hk, agent_kind=agent)
    s.set_positions(np.arange(count), states)
    s.simulate(np.full(count, beta, np.float32), sims)
    info = s.root_info()
    ch = s.root_children(int(info["n_children"].max()))
    best = s.select_best_actions()
    games = []
    for g in range(count):
        k = int(info["n_children"][g])
        games.append({"tps": O.to_tps(oracle, states[g]), "root_visits": int(info["visit_count"][g]),
                      "root_eval": [int(info["eval_tag"][g]), int(info["eval_bits"][g])],
                      "root_std_bits": int(info["std_dev"][g].view(np.uint32)),
                      "moves": [int(x) for x in ch["move_idx"][g, :k]], "visits": [int(x) for x in ch["visits"][g, :k]],
                      "eval_tag": [int(x) for x in ch["eval_tag"][g, :k]], "eval_bits": [int(x) for x in ch["eval_bits"][g, :k]],
                      "std_bits": [int(x) for x in ch["std_dev"][g, :k].view(np.uint32)], "best": int(best[g])})
    cases.append({"n": n, "half_komi": hk, "agent": agent, "sims": sims, "beta": beta, "games": games})
with open(os.path.join(HERE, "search_roots.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_search_fixture.py", "cases": cases}, f, separators=(",", ":"))
print("wrote", sum(len(c["games"]) for c in cases), "games")
