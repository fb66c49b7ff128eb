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
