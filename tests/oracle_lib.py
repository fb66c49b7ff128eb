"""ctypes view of oracle/liboracle.so (CPU oracle, test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_SQ = 36


class TzState(C.Structure):
    _fields_ = [
        ("colors", C.c_uint64 * MAX_SQ),
        ("height", C.c_uint8 * MAX_SQ),
        ("top", C.c_uint8 * MAX_SQ),
        ("stones", C.c_uint8 * 2),
        ("caps", C.c_uint8 * 2),
        ("to_move", C.c_uint8),
        ("n", C.c_uint8),
        ("half_komi", C.c_int8),
        ("pad0", C.c_uint8),
        ("ply", C.c_uint16),
        ("reversible_plies", C.c_uint16),
    ]


class _EvalU(C.Union):
    _fields_ = [("value", C.c_float), ("ply", C.c_uint32)]


class TzRootInfo(C.Structure):
    _fields_ = [
        ("visit_count", C.c_uint32),
        ("n_children", C.c_uint32),
        ("eval_tag", C.c_uint8),
        ("is_terminal_env", C.c_uint8),
        ("ply", C.c_uint16),
        ("eval", _EvalU),
        ("std_dev", C.c_float),
        ("logit", C.c_float),
        ("probability", C.c_float),
    ]


assert C.sizeof(TzState) == 376
assert C.sizeof(TzRootInfo) == 28

AGENT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(TzState), C.POINTER(C.c_uint16),
                       C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                       C.POINTER(C.c_float))

STATE_DTYPE = np.dtype([
    ("colors", np.uint64, (MAX_SQ,)), ("height", np.uint8, (MAX_SQ,)), ("top", np.uint8, (MAX_SQ,)),
    ("stones", np.uint8, (2,)), ("caps", np.uint8, (2,)), ("to_move", np.uint8), ("n", np.uint8),
    ("half_komi", np.int8), ("pad0", np.uint8), ("ply", np.uint16), ("reversible_plies", np.uint16),
], align=True)
assert STATE_DTYPE.itemsize == 376

ROOT_INFO_DTYPE = np.dtype([
    ("visit_count", np.uint32), ("n_children", np.uint32), ("eval_tag", np.uint8),
    ("is_terminal_env", np.uint8), ("ply", np.uint16), ("eval_bits", np.uint32),
    ("std_dev", np.float32), ("logit", np.float32), ("probability", np.float32)], align=True)
assert ROOT_INFO_DTYPE.itemsize == 28

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path) or os.path.exists("/usr/bin/g++"):
        try:
            build()
        except Exception:
            if not os.path.exists(path):
                raise
    lib = C.CDLL(path)
    f32p, u16p, i32p, u32p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint8))
    sp = C.POINTER(TzState)
    lib.tzo_expf.restype = C.c_float
    lib.tzo_expf.argtypes = [C.c_float]
    lib.tzo_logf.restype = C.c_float
    lib.tzo_logf.argtypes = [C.c_float]
    lib.tzo_powif.restype = C.c_float
    lib.tzo_powif.argtypes = [C.c_float, C.c_int]
    lib.tzo_softmax.argtypes = [f32p, C.c_int, f32p]
    lib.tzo_eval_cmp.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_uint32]
    lib.tzo_eval_to_f32.restype = C.c_float
    lib.tzo_eval_to_f32.argtypes = [C.c_int, C.c_uint32]
    lib.tzo_exploration_rate.restype = C.c_float
    lib.tzo_exploration_rate.argtypes = [C.c_float]
    lib.tzo_ucb.restype = C.c_float
    lib.tzo_ucb.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.tzo_state_default.argtypes = [C.c_int, C.c_int, sp]
    lib.tzo_state_from_tps.argtypes = [C.c_char_p, C.c_int, C.c_int, sp]
    lib.tzo_state_to_tps.argtypes = [sp, C.c_char_p, C.c_int]
    lib.tzo_possible_moves.argtypes = [sp, u16p, C.c_int]
    lib.tzo_play.argtypes = [sp, C.c_uint16, sp]
    lib.tzo_terminal.argtypes = [sp]
    lib.tzo_result.argtypes = [sp]
    lib.tzo_flat_diff.argtypes = [sp]
    lib.tzo_game_repr.argtypes = [sp, f32p]
    lib.tzo_move_to_ptn.argtypes = [C.c_int, C.c_uint16, C.c_char_p, C.c_int]
    lib.tzo_move_from_ptn.argtypes = [C.c_int, C.c_char_p, u16p]
    lib.tzo_new_opening.argtypes = [C.c_int, C.c_int, C.c_int, sp]
    lib.tzo_search_create.restype = C.c_void_p
    lib.tzo_search_create.argtypes = [C.c_int, AGENT_FN, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.tzo_search_destroy.argtypes = [C.c_void_p]
    lib.tzo_search_set_positions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.tzo_search_get_positions.argtypes = [C.c_void_p, C.c_void_p]
    lib.tzo_search_new_openings.argtypes = [C.c_void_p, C.c_void_p]
    lib.tzo_search_simulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.tzo_search_apply_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    lib.tzo_search_root_info.argtypes = [C.c_void_p, C.c_void_p]
    lib.tzo_search_root_children.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
    lib.tzo_search_select_best_actions.argtypes = [C.c_void_p, C.c_void_p]
    lib.tzo_search_improved_policy.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    lib.tzo_search_ube_target.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
    lib.tzo_search_selfplay_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_void_p]
    lib.tzo_search_step.argtypes = [C.c_void_p, C.c_void_p]
    lib.tzo_search_restart_terminal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tzo_search_gumbel_sh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.tzo_search_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tzo_search_replay.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.tzo_kat_find_tinue.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_float,
                                       C.c_int, u16p]
    lib.tzo_kat_safecrack.argtypes = [C.c_int]
    _lib = lib
    return lib


# ------------------------------------------------------------------ convenience wrappers
def state_from_tps(lib, tps, n, half_komi):
    s = TzState()
    rc = lib.tzo_state_from_tps(tps.encode(), n, half_komi, C.byref(s))
    assert rc == 0, (rc, tps)
    return s


def state_default(lib, n, half_komi):
    s = TzState()
    lib.tzo_state_default(n, half_komi, C.byref(s))
    return s


def to_tps(lib, s):
    buf = C.create_string_buffer(512)
    assert lib.tzo_state_to_tps(C.byref(s), buf, 512) == 0
    return buf.value.decode()


def possible_moves(lib, s):
    out = (C.c_uint16 * 1024)()
    n = lib.tzo_possible_moves(C.byref(s), out, 1024)
    assert n >= 0
    return list(out[:n])


def play(lib, s, move):
    o = TzState()
    rc = lib.tzo_play(C.byref(s), move, C.byref(o))
    assert rc == 0, rc
    return o


def ptn(lib, n, idx):
    buf = C.create_string_buffer(32)
    assert lib.tzo_move_to_ptn(n, idx, buf, 32) == 0
    return buf.value.decode()


def from_ptn(lib, n, text):
    out = C.c_uint16()
    rc = lib.tzo_move_from_ptn(n, text.encode(), C.byref(out))
    assert rc == 0, (rc, text)
    return out.value


def game_repr(lib, s):
    n = s.n
    ch = lib.tzo_input_channels(n)
    out = np.zeros(ch * n * n, dtype=np.float32)
    lib.tzo_game_repr(C.byref(s), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def states_array(states):
    """list of TzState -> numpy structured array (contiguous tz_state[])."""
    arr = np.zeros(len(states), dtype=STATE_DTYPE)
    for i, s in enumerate(states):
        C.memmove(arr.ctypes.data + i * 376, C.byref(s), 376)
    return arr


class OracleSearch:
    """BatchedMCTS over the oracle with the same call surface as takzero_amd.BatchedMCTS."""

    def __init__(self, lib, batch, n, half_komi, agent_kind=1, agent_fn=None):
        self.lib, self.batch, self.n, self.half_komi = lib, batch, n, half_komi
        self._cb = AGENT_FN(agent_fn) if agent_fn is not None else C.cast(None, AGENT_FN)
        self.h = lib.tzo_search_create(agent_kind, self._cb, None, batch, n, half_komi)

    def close(self):
        if self.h:
            self.lib.tzo_search_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_positions(self, idx, states):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        arr = states_array(states) if isinstance(states, list) else states
        assert self.lib.tzo_search_set_positions(self.h, len(idx), idx.ctypes.data, arr.ctypes.data) == 0

    def get_positions(self):
        arr = np.zeros(self.batch, dtype=STATE_DTYPE)
        self.lib.tzo_search_get_positions(self.h, arr.ctypes.data)
        return arr

    def new_openings(self, choice):
        choice = np.ascontiguousarray(choice, dtype=np.int32)
        self.lib.tzo_search_new_openings(self.h, choice.ctypes.data)

    def simulate(self, betas, n_sims=1):
        betas = np.ascontiguousarray(betas, dtype=np.float32)
        assert self.lib.tzo_search_simulate(self.h, betas.ctypes.data, n_sims) == 0

    def apply_noise(self, noise, ratio):
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        rc = self.lib.tzo_search_apply_noise(self.h, noise.ctypes.data, noise.shape[1], ratio)
        assert rc == 0, rc

    def root_info(self):
        arr = np.zeros(self.batch, dtype=ROOT_INFO_DTYPE)
        self.lib.tzo_search_root_info(self.h, arr.ctypes.data)
        return arr

    def root_children(self, amax=512):
        B = self.batch
        out = dict(move_idx=np.zeros((B, amax), np.uint16), visits=np.zeros((B, amax), np.uint32),
                   eval_tag=np.zeros((B, amax), np.uint8), eval_bits=np.zeros((B, amax), np.uint32),
                   logit=np.zeros((B, amax), np.float32), prob=np.zeros((B, amax), np.float32),
                   std_dev=np.zeros((B, amax), np.float32))
        rc = self.lib.tzo_search_root_children(self.h, amax, *[out[k].ctypes.data for k in
                                               ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev")])
        assert rc == 0
        return out

    def node(self, game, path, amax=512):
        """Mirror of BatchedMCTS.node: None if the path leaves the tree."""
        import ctypes as C

        from takzero_amd._lib import ROOT_INFO_DTYPE

        p = np.ascontiguousarray(path, dtype=np.uint16)
        info = np.zeros(1, ROOT_INFO_DTYPE)
        out = dict(move_idx=np.zeros(amax, np.uint16), visits=np.zeros(amax, np.uint32), eval_tag=np.zeros(amax, np.uint8),
                   eval_bits=np.zeros(amax, np.uint32), logit=np.zeros(amax, np.float32), prob=np.zeros(amax, np.float32),
                   std_dev=np.zeros(amax, np.float32))
        self.lib.tzo_search_node.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        rc = self.lib.tzo_search_node(self.h, game, p.ctypes.data if len(p) else None, len(p), info.ctypes.data, amax,
                                      *[out[k].ctypes.data for k in ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev")])
        if rc != 0:
            return None
        nc = int(info[0]["n_children"])
        return info[0], {k: v[:nc] for k, v in out.items()}

    def select_best_actions(self):
        out = np.zeros(self.batch, np.uint16)
        self.lib.tzo_search_select_best_actions(self.h, out.ctypes.data)
        return out

    def improved_policy(self, visitations, amax=512):
        out = np.zeros((self.batch, amax), np.float32)
        assert self.lib.tzo_search_improved_policy(self.h, visitations, amax, out.ctypes.data) == 0
        return out

    def ube_target(self, beta):
        out = np.zeros(self.batch, np.float32)
        self.lib.tzo_search_ube_target(self.h, beta, out.ctypes.data)
        return out

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.uint16)
        self.lib.tzo_search_step(self.h, actions.ctypes.data)

    def restart_terminal(self, choice):
        choice = np.ascontiguousarray(choice, dtype=np.int32)
        out = np.zeros(self.batch, np.int8)
        self.lib.tzo_search_restart_terminal(self.h, choice.ctypes.data, out.ctypes.data)
        return out

    def gumbel_sh(self, betas, k, budget, gumbel):
        betas = np.ascontiguousarray(betas, dtype=np.float32)
        gumbel = np.ascontiguousarray(gumbel, dtype=np.float32)
        out = np.zeros(self.batch, np.uint16)
        rc = self.lib.tzo_search_gumbel_sh(self.h, betas.ctypes.data, k, budget, gumbel.ctypes.data,
                                           gumbel.shape[1], out.ctypes.data)
        assert rc == 0
        return out

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.tzo_search_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # aliases so that drivers written against takzero_amd.api.BatchedMCTS run on the oracle unchanged
    def gumbel_sequential_halving(self, betas, k, budget, gumbel):
        return self.gumbel_sh(betas, k, budget, gumbel)

    def restart_terminal_envs(self, choice):
        return self.restart_terminal(choice)
