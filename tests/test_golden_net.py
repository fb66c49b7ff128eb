"""Committed golden vectors of the network forward (tests/golden/net_forward.json, made by make_net_fixture.py from
the PyTorch fp32 graph): on CPU they pin the weight generator and the torch restatement; on the GPU the HIP forward
is held to them — fp32 path 1e-4, f16 MFMA path 1e-3 (the north star's tolerance), bf16 MFMA path 2e-2."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
CASES = json.load(open(os.path.join(HERE, "golden", "net_forward.json")))["cases"]


def _weights(case):
    from takzero_amd import weights as W

    w = W.init_weights(case["arch"], n=case["n"], blocks=case["blocks"], seed=case["seed"], trained_stats=case["trained_stats"])
    checksum = sum(float(np.abs(v).sum(dtype=np.float64)) for v in w.values())
    assert abs(checksum - case["weights_checksum"]) <= 1e-6 * case["weights_checksum"], "weight generator drifted"
    return w


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d" % (c["n"], c["n"]))
def test_torch_restatement_reproduces_the_golden_vectors(case):
    import nets_torch as T

    oracle = O.load()
    w = _weights(case)
    n = case["n"]
    for pos in case["positions"]:
        s = O.state_from_tps(oracle, pos["tps"], n, case["half_komi"])
        assert list(O.possible_moves(oracle, s)) == pos["legal"]  # the rules give the same legal moves in the same order
        planes = O.game_repr(oracle, s).reshape(1, -1, n, n)
        pol, val, ube = T.forward(w, planes, case["blocks"])
        got = pol.reshape(-1).numpy()[pos["legal"]]
        assert np.allclose(got, np.float32(pos["logits"]), atol=2e-5) and abs(float(val[0]) - pos["value"]) < 2e-5
        assert abs(float(ube[0]) - pos["ube"]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d" % (c["n"], c["n"]))
def test_hip_forward_matches_the_golden_vectors(case):
    from gpu_util import require_gpu

    A = require_gpu()
    w = _weights(case)
    n = case["n"]
    states = np.array([A.state_from_tps(p["tps"], n, case["half_komi"]) for p in case["positions"]], dtype=A.STATE_DTYPE)
    acts = [p["legal"] for p in case["positions"]]
    for prec, tol in ((A.PREC_F32, 1e-4), (A.PREC_F16X2, 1e-4), (A.PREC_F16C8, 2e-4), (A.PREC_F16, 1e-3), (A.PREC_BF16, 2e-2)):
        net = A.Net(arch=case["arch"], n=n, precision=prec, blocks=case["blocks"]).load_tensors(w)
        logits, value, _var = net.policy_value_uncertainty(states, acts)
        _pol, _val, ube = net.forward_raw(states)
        for i, p in enumerate(case["positions"]):
            assert np.abs(logits[i] - np.float32(p["logits"])).max() <= tol, (prec, i)
            assert abs(float(value[i]) - p["value"]) <= tol and abs(float(ube[i]) - p["ube"]) <= tol, (prec, i)
        net.close()
