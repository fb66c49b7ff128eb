"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/takzero_hip.h
declares, host text helpers agree with the oracle, and the HIP path fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import takzero_amd.api as A

    lib = A._lib.load()
    header = open(os.path.join(ROOT, "include", "takzero_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tz_[a-z0-9_]+)\s*\(", header))
    assert declared == set(A._lib.SYMBOLS), declared ^ set(A._lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tz_version() >= 1
    assert C.sizeof(O.TzState) == A._lib.STATE_DTYPE.itemsize == 376


def test_text_helpers_agree_with_oracle(oracle):
    import takzero_amd.api as A

    rng = np.random.default_rng(3)
    for n in (3, 4, 5, 6):
        for idx in range(A.policy_size(n)):
            name = O.ptn(oracle, n, idx)
            assert A.move_to_ptn(n, idx) == name
            if idx % 7 == 0:
                assert A.move_from_ptn(n, name) == idx
        s = O.state_default(oracle, n, 4)
        for ply in range(40):
            if oracle.tzo_terminal(C.byref(s)) != -1:
                break
            tps = O.to_tps(oracle, s)
            mine = A.state_from_tps(tps, n, 4)
            s.reversible_plies = 0
            assert mine.tobytes() == bytes(s), tps
            assert A.state_to_tps(mine) == tps
            mv = O.possible_moves(oracle, s)
            s = O.play(oracle, s, mv[int(rng.integers(len(mv)))])
    for bad in ("", "x5/x5/x5/x5 1 1", "x5/x5/x5/x5/x5 3 1", "x5/x5/x5/x5/x4,7 1 1"):
        with pytest.raises(A.TakzeroError) as e:
            A.state_from_tps(bad, 5, 4)
        assert e.value.code == -2
    for bad in ("z1", "6a1+", "a1+6", "Ca1+", "3a1+11"):
        with pytest.raises(A.TakzeroError):
            A.move_from_ptn(5, bad)


def test_no_silent_cpu_fallback():
    import takzero_amd.api as A

    if A._lib.load().tz_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(A.TakzeroError) as e:
        A.Net(arch=A.ARCH_NET5)
    assert e.value.code == -3
    with pytest.raises(A.TakzeroError) as e:
        A.BatchedMCTS(4, 5, 4, agent_kind=A.AGENT_DUMMY)
    assert e.value.code == -3


def test_product_does_not_reference_oracle():
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "takzero_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "liboracle" not in text and "nets_torch" not in text, f


def test_weight_container_round_trip(tmp_path):
    from takzero_amd import weights as W

    w = W.init_weights(W.ARCH_TEST, n=4, blocks=1, seed=9)
    p = tmp_path / "w.tzw"
    W.save_tzw(str(p), w)
    r = W.load_tzw(str(p))
    assert set(r) == set(w) and all(np.array_equal(r[k], w[k]) for k in w)
