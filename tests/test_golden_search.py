"""Committed golden vectors of the search (tests/golden/search_roots.json, from the CPU oracle with the reference's
Dummy / Simple agents): the oracle must keep reproducing them (CPU), and the HIP tree kernels must match them bit for
bit (GPU) — visit counts, proven / valued evaluations, std-dev bit patterns, chosen moves."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "search_roots.json")))["cases"]
IDS = ["%dx%d-agent%d" % (c["n"], c["n"], c["agent"]) for c in CASES]


def _check(search, case):
    games = case["games"]
    info = search.root_info()
    ch = search.root_children(max(len(g["moves"]) for g in games))
    best = search.select_best_actions()
    for i, g in enumerate(games):
        k = len(g["moves"])
        assert int(info["n_children"][i]) == k and int(info["visit_count"][i]) == g["root_visits"], i
        assert [int(info["eval_tag"][i]), int(info["eval_bits"][i])] == g["root_eval"], i
        assert int(info["std_dev"][i].view(np.uint32)) == g["root_std_bits"], i
        for field, key in (("move_idx", "moves"), ("visits", "visits"), ("eval_tag", "eval_tag"), ("eval_bits", "eval_bits")):
            assert [int(x) for x in ch[field][i, :k]] == g[key], (i, field)
        assert [int(x) for x in ch["std_dev"][i, :k].view(np.uint32)] == g["std_bits"], i
        assert int(best[i]) == g["best"], i


def _positions(lib, case):
    return O.states_array([O.state_from_tps(lib, g["tps"], case["n"], case["half_komi"]) for g in case["games"]])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_the_golden_roots(case):
    lib = O.load()
    B = len(case["games"])
    s = O.OracleSearch(lib, B, case["n"], case["half_komi"], agent_kind=case["agent"])
    s.set_positions(np.arange(B), _positions(lib, case))
    s.simulate(np.full(B, case["beta"], np.float32), case["sims"])
    _check(s, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_search_reproduces_the_golden_roots(case):
    from gpu_util import require_gpu

    A = require_gpu()
    lib = O.load()
    B = len(case["games"])
    s = A.BatchedMCTS(B, case["n"], case["half_komi"], agent_kind=case["agent"], node_capacity=1 << 15)
    s.set_positions(np.arange(B), _positions(lib, case))
    s.simulate(np.full(B, case["beta"], np.float32), case["sims"])
    _check(s, case)
