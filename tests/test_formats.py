"""Text formats (target.rs:56-73,215-232) — CPU only."""
import numpy as np

from takzero_amd import formats as F
import takzero_amd.api as A


def test_rust_f32_display():
    # Rust's `impl Display for f32`: shortest round-trip decimal, no exponent, no trailing ".0"
    cases = {1.0: "1", 0.5: "0.5", -0.0: "-0", 0.1: "0.1", 1e-7: "0.0000001", 1e20: "100000000000000000000",
             0.997: "0.997", -0.994009: "-0.994009", 16777216.0: "16777216", 0.011656231: "0.011656231",
             3.4028235e38: "340282350000000000000000000000000000000"}
    for v, want in cases.items():
        assert F.rust_f32(v) == want, (v, F.rust_f32(v))
    assert F.rust_f32(float("nan")) == "NaN" and F.rust_f32(float("inf")) == "inf"
    rng = np.random.default_rng(0)
    for x in rng.standard_normal(2000).astype(np.float32):
        s = F.rust_f32(x)
        assert "e" not in s and np.float32(s) == x


def test_target_line_round_trip():
    st = A.state_from_tps("x2,1221,x,1S/2,2C,2,1,x/x,212,21C,2S,2/2211S,2,21,1,1/x2,221S,2,x 2 23", 5, 4)
    moves = np.array([A.move_from_ptn(5, m) for m in ("a1", "Sa1", "3b3-21", "e3+")], np.uint16)
    pol = np.array([0.5, 0.25, 0.125, 0.125], np.float32)
    line = F.format_target(5, st, moves, pol, np.float32(-0.994009), np.float32(0))
    assert line == "x2,1221,x,1S/2,2C,2,1,x/x,212,21C,2S,2/2211S,2,21,1,1/x2,221S,2,x 2 23;-0.994009;0;a1:0.5,Sa1:0.25,3b3-21:0.125,e3+:0.125\n"
    st2, mv2, p2, v2, u2 = F.parse_target(line, 5, 4)
    assert st2.tobytes() == st.tobytes() and np.array_equal(mv2, moves) and np.array_equal(p2, pol)
    assert v2 == np.float32(-0.994009) and u2 == 0


def test_replay_line_round_trip():
    st = A.state_from_tps("x5/x5/x5/x5/2,x3,1 1 2", 5, 4)
    moves = [A.move_from_ptn(5, m) for m in ("c3", "Sd4", "c3>", "Cb2")]
    line = F.format_replay(5, st, moves, F.result_string(1, 0))
    assert line == '[TPS "x5/x5/x5/x5/2,x3,1 1 2"] c3 Sd4 c3> Cb2 R-0\n'
    st2, mv2 = F.parse_replay(line, 5, 4)
    assert st2.tobytes() == st.tobytes() and mv2 == moves
    assert F.format_replay(5, st, moves).endswith("Cb2\n")
    assert F.result_string(2, 1) == "0-F" and F.result_string(3, 2) == "1/2-1/2"


def test_buffer_lengths():
    assert F.format_buffer_lengths(12, 30) == "12,30,42"
    assert F.parse_buffer_lengths("12,30,42") == (12, 30)
    import pytest
    with pytest.raises(ValueError):
        F.parse_buffer_lengths("12,30,43")


def _fields(state):
    """bytes of a tz_state record with its padding zeroed (numpy leaves the padding of a void scalar undefined)"""
    from takzero_amd._lib import STATE_DTYPE

    arr = np.zeros(1, STATE_DTYPE)
    arr[0] = state
    return arr.tobytes()


def test_bulk_target_lines_equal_the_line_by_line_ones():
    """tz_format_targets / tz_parse_targets (native, thousands of lines per move) against format_target / parse_target."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_learn_host import _targets
    from takzero_amd import formats as F

    for n in (4, 5, 6):
        targets = _targets(n, 40, 10 + n)
        # awkward floats: tiny, huge, negative zero, exact integers, denormal, many digits
        odd = [1e-10, 3.4e38, -0.0, 1.0, 0.0, 1e-45, 0.1, 123456.79, 16777216.0, 2.5e-7, 0.30000001]
        st, mv, pol, v, u = targets[0]
        pol = pol.copy()
        pol[:min(len(pol), len(odd))] = np.float32(odd[:len(pol)])
        targets[0] = (st, mv, pol, np.float32(-0.0), np.float32(1e-10))
        want = "".join(F.format_target(n, *t) for t in targets)
        got = F.format_targets(n, targets)
        assert got == want
        back, consumed, skipped = F.parse_targets(got.encode(), n, 4)
        assert consumed == len(got.encode()) and skipped == 0 and len(back) == len(targets)
        for a, b in zip(back, targets):
            ref = F.parse_target(F.format_target(n, *b), n, 4)
            assert _fields(a[0]) == _fields(ref[0]) and np.array_equal(a[1], ref[1])
            assert np.array_equal(a[2].view(np.uint32), ref[2].view(np.uint32))
            assert np.float32(a[3]).tobytes() == np.float32(ref[3]).tobytes() and np.float32(a[4]) == np.float32(ref[4])
    # skipping and the half-written tail (learn tails the file while selfplay appends)
    lines = F.format_targets(5, _targets(5, 3, 1)).splitlines(keepends=True)
    data = (lines[0] + "not a target\n" + "x5/x5/x5/x5/x5 1 1;0.5;1;a1:0.5,\n" + ";;;\n" + lines[1] + lines[2][:25]).encode()
    back, consumed, skipped = F.parse_targets(data, 5, 4)
    assert len(back) == 2 and skipped == 3 and consumed == len(data) - 25
    assert F.parse_targets(b"", 5, 4) == ([], 0, 0) and F.format_targets(5, []) == ""
