"""The native host drivers (csrc/tz_host.cpp: selfplay::main, reanalyze::main) run over the CPU oracle's search under
AddressSanitizer + UBSan (GPU sanitizers are not available on the pool; this is the same driver code).  What they
write is checked against the rules: every target lists exactly the legal moves of its position, replays re-validate."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from host_oracle_util import build, run


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    return build(tmp_path_factory.mktemp("hoo"), sanitize=True)


@pytest.mark.parametrize("kind,sims,k,exploration,agent", [(0, 20, 64, 1, 2), (1, 16, 4, 0, 2), (2, 0, 64, 0, 1)])
def test_native_drivers_over_the_oracle_under_sanitizers(harness, tmp_path, kind, sims, k, exploration, agent):
    from takzero_amd import formats as F

    oracle = O.load()
    n = 4
    out = run(harness, tmp_path / "out", n, 4, agent, 24, kind, sims, k, exploration, 60, 5)
    assert out["targets"] and out["replays"] and out["positions"] > 0
    for text in (out["targets"], out["reanalyze"]):
        for line in text.decode().splitlines(keepends=True)[:400]:
            st, mv, pol, value, ube = F.parse_target(line, n, 4)
            legal = O.possible_moves(oracle, O.TzState.from_buffer_copy(np.array([st]).tobytes()))
            assert [int(m) for m in mv] == list(legal) and -1.0 <= value <= 1.0
    total = 0
    for line in out["replays"].decode().splitlines():
        start, moves = F.parse_replay(line, n, 4)
        s = O.TzState.from_buffer_copy(np.array([start]).tobytes())
        for m in moves:
            s = O.play(oracle, s, int(m))
        assert oracle.tzo_terminal(C.byref(s)) != -1
        total += len(moves)
    assert out["positions"] == total           # the reanalyze buffer holds every pre-move state of every replay
    if exploration:
        assert out["exploration"] and all(len(line.split()) <= 15 for line in out["exploration"].decode().splitlines())
    if kind != 2 and out["positions"] >= 24:
        assert out["reanalyze"].count(b"\n") == 48   # two iterations of one target per position
