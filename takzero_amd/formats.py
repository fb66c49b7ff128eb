"""Text formats of the reference's data files, byte-compatible with what `learn`, `reanalyze` and the
plotting scripts parse (takzero/src/target.rs:56-73,99-143,215-268):

    target line : "{tps};{value};{ube};{move}:{p},{move}:{p},...\\n"
    replay line : "[TPS \\"{tps}\\"] m1 m2 ... {result}\\n"   (result omitted while the game is ongoing)
    buffer_lengths.txt : "{selfplay},{reanalyze},{sum}"       (learn/src/main.rs:201-206)

Floats are written as Rust's `Display for f32` writes them: shortest decimal that round-trips, never an
exponent, `1` rather than `1.0`, `-0`, `NaN`, `inf`."""
import numpy as np

from . import api

RESULTS = {(1, 0): "R-0", (1, 1): "0-R", (2, 0): "F-0", (2, 1): "0-F"}  # (reason, winner) -> takparse GameResult


def rust_f32(x):
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def format_target(n, state, moves, policy, value, ube):
    """impl Display for Target (target.rs:56-73)."""
    pol = ",".join("%s:%s" % (api.move_to_ptn(n, int(m)), rust_f32(p)) for m, p in zip(moves, policy))
    return "%s;%s;%s;%s\n" % (api.state_to_tps(state), rust_f32(value), rust_f32(ube), pol)


def parse_target(line, n, half_komi):
    """impl FromStr for Target (target.rs:99-143), without the legal-set check (the device validates moves)."""
    tps, value, ube, pol = line.strip().split(";")
    moves, probs = [], []
    for item in pol.split(","):
        mv, p = item.split(":")
        moves.append(api.move_from_ptn(n, mv))
        probs.append(np.float32(p))
    return api.state_from_tps(tps, n, half_komi), np.array(moves, np.uint16), np.array(probs, np.float32), \
        np.float32(value), np.float32(ube)


def _amax(n):
    return 512 if n < 6 else 1024


def columns(targets, n):
    """list of (state, moves, policy, value, ube) -> padded arrays (states, moves, policy, nmoves, value, ube)."""
    T, amax = len(targets), _amax(n)
    states = np.zeros(T, api.STATE_DTYPE)
    moves = np.zeros((T, amax), np.uint16)
    policy = np.zeros((T, amax), np.float32)
    nmoves = np.zeros(T, np.int32)
    value = np.zeros(T, np.float32)
    ube = np.zeros(T, np.float32)
    for i, (st, mv, pol, v, u) in enumerate(targets):
        k = len(mv)
        states[i] = st
        moves[i, :k] = mv
        policy[i, :k] = pol
        nmoves[i], value[i], ube[i] = k, v, u
    return states, moves, policy, nmoves, value, ube


def format_targets(n, targets):
    """All lines of a batch of targets in one native call (tz_format_targets): same bytes as format_target."""
    import ctypes as C

    from . import _lib

    if not targets:
        return ""
    states, moves, policy, nmoves, value, ube = columns(targets, n)
    cap = int(len(targets) * 160 + int(nmoves.sum()) * 32)
    out = C.create_string_buffer(cap)
    written = C.c_uint64()
    _lib.check(_lib.load().tz_format_targets(n, len(targets), states.ctypes.data, moves.ctypes.data, policy.ctypes.data,
                                             nmoves.ctypes.data, moves.shape[1], value.ctypes.data, ube.ctypes.data, out, cap,
                                             C.byref(written)))
    return out.raw[:written.value].decode()


def parse_targets(data, n, half_komi, chunk=8192):
    """Complete lines of `data` (bytes) -> (targets, bytes consumed, lines skipped), natively (tz_parse_targets)."""
    import ctypes as C

    from . import _lib

    lib, amax = _lib.load(), _amax(n)
    targets, consumed, skipped = [], 0, 0
    while True:
        states = np.zeros(chunk, api.STATE_DTYPE)
        moves = np.zeros((chunk, amax), np.uint16)
        policy = np.zeros((chunk, amax), np.float32)
        nmoves = np.zeros(chunk, np.int32)
        value = np.zeros(chunk, np.float32)
        ube = np.zeros(chunk, np.float32)
        cnt, used, skp = C.c_int32(), C.c_uint64(), C.c_int32()
        view = data[consumed:]
        _lib.check(lib.tz_parse_targets(view, len(view), n, half_komi, chunk, amax, states.ctypes.data, moves.ctypes.data,
                                        policy.ctypes.data, nmoves.ctypes.data, value.ctypes.data, ube.ctypes.data,
                                        C.byref(cnt), C.byref(used), C.byref(skp)))
        for i in range(cnt.value):
            k = int(nmoves[i])
            # a 1-element slice copy keeps the struct's padding bytes zero (np.void.copy() does not)
            targets.append((states[i:i + 1].copy()[0], moves[i, :k].copy(), policy[i, :k].copy(), np.float32(value[i]), np.float32(ube[i])))
        consumed += used.value
        skipped += skp.value
        if cnt.value < chunk or used.value == 0:
            break
    return targets, consumed, skipped


def result_string(reason, winner):
    """takparse GameResult as fast-tak converts it (GameResult::try_from(env.result()), target.rs:226-230)."""
    if winner == 2:
        return "1/2-1/2"
    return RESULTS[(1 if reason == 1 else 2, int(winner))]


def format_replay(n, start_state, moves, result=None):
    """impl Display for Replay (target.rs:215-232)."""
    s = '[TPS "%s"]' % api.state_to_tps(start_state)
    for m in moves:
        s += " " + api.move_to_ptn(n, int(m))
    if result:
        s += " " + result
    return s + "\n"


def parse_replay(line, n, half_komi):
    """impl FromStr for Replay (target.rs:234-268): returns (start_state, move indices).  Move legality is
    re-validated on the device (BatchedMCTS.play_moves)."""
    line = line.strip()
    if not line.startswith('[TPS "'):
        raise ValueError("missing TPS")
    end = line.index('"]')
    state = api.state_from_tps(line[6:end], n, half_komi)
    moves = []
    for tok in line[end + 2:].split():
        if tok in ("R-0", "0-R", "F-0", "0-F", "1/2-1/2", "1-0", "0-1"):
            break
        if tok.endswith(".") and tok[:-1].isdigit():
            continue  # move numbers, if a PTN writer added them
        moves.append(api.move_from_ptn(n, tok))
    return state, moves


def format_buffer_lengths(selfplay, reanalyze):
    return "%d,%d,%d" % (selfplay, reanalyze, selfplay + reanalyze)


def parse_buffer_lengths(text):
    """read_buffer_lengths with its checksum (selfplay/src/main.rs:372-387)."""
    nums = [int(s) for s in text.split(",") if s.strip().isdigit()]
    if len(nums) < 3:
        raise ValueError("missing component")
    if nums[0] + nums[1] != nums[2]:
        raise ValueError("wrong checksum")
    return nums[0], nums[1]
