"""Host-side mirror of the reference's interface for the hot path, over the C ABI:

    Net          <- Network + Agent        takzero/src/network/mod.rs:10-45, search/agent.rs:5-14
    BatchedMCTS  <- BatchedMCTS<B, E>      takzero/src/search/node/batched.rs:24-409

Method names, argument meaning and error behaviour follow the reference; randomness is an
argument wherever the reference takes `rng` (the draws stay with the caller).
Everything here is plumbing: the work happens in libtakzero_hip.so on the GPU."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import ROOT_INFO_DTYPE, STATE_DTYPE, TakzeroError, check

ARCH_NET4_SIMHASH, ARCH_NET5, ARCH_NET6_SIMHASH, ARCH_TEST = 4, 5, 6, 100
PREC_BF16, PREC_F32, PREC_F16, PREC_F16X2, PREC_F16C8, PREC_F16C6 = 0, 1, 2, 3, 4, 5
# Arithmetic of the trunk.  fp16 storage with fp32 accumulation (PREC_F16) is the throughput default: its logits are within
# ~2e-4 *relative* of the fp32 graph through the 41 stacked convs (1.5e-4 absolute at the random-init logit scale of 0.2,
# ~2e-3 at a trained net's logit scale of 10).  PREC_F16X2 ("f16x2") carries every operand as a hi / lo pair of halves
# (three fp16 MFMAs per product, fp32 accumulate) and stays within the north star's absolute 1e-3 at trained scale, at
# about 3x the MFMA work; PREC_F16C8 ("f16c8") keeps the fp16 product and takes the two correction products through FP8 (E4M3)
# copies of the operands at twice the MFMA rate: 1.3e-4 at trained scale at 2.3x the fp16 kernel's time - the cheaper of the two
# for a host that wants the reference's moves (under Gumbel 64 / 768 at trained scale the fp16 default picks the fp32 path's
# action in 93 % of games, these two in all of them).  bf16 runs the fp16 kernels 5 % faster at 1e-3 .. 7e-3 (random-init
# scale).  TZ_PRECISION selects.
PREC_NAMES = {"bf16": PREC_BF16, "f16": PREC_F16, "f32": PREC_F32, "f16x2": PREC_F16X2, "f16c8": PREC_F16C8, "f16c6": PREC_F16C6}
PREC_DEFAULT = PREC_NAMES[os.environ.get("TZ_PRECISION", "f16")]
AGENT_NET, AGENT_DUMMY, AGENT_SIMPLE = 0, 1, 2
EVAL_VALUE, EVAL_WIN, EVAL_LOSS, EVAL_DRAW = 0, 1, 2, 3
TERMINAL_NONE, TERMINAL_WIN, TERMINAL_LOSS, TERMINAL_DRAW = -1, 0, 1, 2
DISCOUNT_FACTOR = np.float32(0.997)  # search/mod.rs:7


# ---------------------------------------------------------------- text helpers (takparse formats)
def state_from_tps(tps, n, half_komi):
    out = np.zeros(1, STATE_DTYPE)
    check(_lib.load().tz_state_from_tps(tps.encode(), n, half_komi, out.ctypes.data))
    return out[0]


def state_to_tps(state):
    arr = np.ascontiguousarray(np.asarray(state, dtype=STATE_DTYPE).reshape(1))
    buf = C.create_string_buffer(512)
    check(_lib.load().tz_state_to_tps(arr.ctypes.data, buf, 512))
    return buf.value.decode()


def move_to_ptn(n, move_index):
    buf = C.create_string_buffer(32)
    check(_lib.load().tz_move_to_ptn(n, int(move_index), buf, 32))
    return buf.value.decode()


def move_from_ptn(n, text):
    out = C.c_uint16()
    check(_lib.load().tz_move_from_ptn(n, text.encode(), C.byref(out)))
    return out.value


def policy_size(n):
    return _lib.load().tz_policy_size(n)


def input_channels(n):
    return _lib.load().tz_input_channels(n)


def eval_to_f32(tag, bits):
    """impl From<Eval> for f32 (eval.rs:95-105)."""
    if tag == EVAL_VALUE:
        return np.uint32(bits).view(np.float32)
    r, base, b = np.float32(1.0), DISCOUNT_FACTOR, int(bits)
    while True:  # f32::powi (square and multiply)
        if b & 1:
            r = np.float32(r * base)
        b >>= 1
        if b == 0:
            break
        base = np.float32(base * base)
    return np.float32(r * np.float32({EVAL_WIN: 1.0, EVAL_LOSS: -1.0, EVAL_DRAW: 0.0}[int(tag)]))


def _states(states):
    # field-wise copy into a zeroed buffer: numpy leaves the struct's tail padding uninitialised when it gathers
    # records (fancy indexing), and positions are compared / hashed as raw bytes
    arr = np.asarray(states, dtype=STATE_DTYPE).reshape(-1)
    out = np.zeros(arr.shape[0], STATE_DTYPE)
    out[...] = arr
    return out


class Net:
    """A network on one GPU.  `Net(arch=ARCH_NET5)` is Network::new; `load` is Network::load."""

    def __init__(self, arch=ARCH_NET5, n=0, device=0, precision=None, blocks=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        if precision is None:
            precision = PREC_DEFAULT
        check(self.lib.tz_net_create(n, arch, device, precision, blocks, C.byref(h)))
        self.h = h
        self.arch, self.precision, self.blocks, self.device = arch, precision, blocks, device
        self.n = {ARCH_NET5: 5, ARCH_NET4_SIMHASH: 4, ARCH_NET6_SIMHASH: 6}.get(arch, n)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, path):
        """Network::load (network/mod.rs:24-28; net6_simhash.rs:173-190): a LibTorch archive as the reference's `learn`
        writes it (`.ot`, read natively by the library) or the flat .tzw container, recognised by content; SimHash nets also
        pick up `bitvec.bin` from the same directory."""
        check(self.lib.tz_net_load_weights(self.h, str(path).encode()))
        return self

    def load_tensors(self, tensors):
        from .weights import dumps_tzw

        blob = dumps_tzw(tensors)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        check(self.lib.tz_net_load_weights_mem(self.h, C.addressof(buf), len(blob)))
        return self

    @classmethod
    def new(cls, arch=ARCH_NET5, seed=None, n=0, device=0, precision=None, blocks=0):
        """Network::new(device, seed) (network/mod.rs:11): randomly initialised weights, tch's default initialisers
        (tz_net_init_random)."""
        net = cls(arch=arch, n=n, device=device, precision=precision, blocks=blocks)
        check(net.lib.tz_net_init_random(net.h, 0 if seed is None else int(seed)))
        return net

    def tensors(self):
        """The host-side VarStore: name -> fp32 array (flat), as loaded / initialised."""
        out, buf, cnt = {}, C.create_string_buffer(256), C.c_uint64()
        for i in range(self.lib.tz_net_tensor_count(self.h)):
            check(self.lib.tz_net_tensor_info(self.h, i, buf, 256, C.byref(cnt)))
            arr = np.zeros(cnt.value, np.float32)
            check(self.lib.tz_net_get_tensor(self.h, buf.value, arr.ctypes.data, arr.size, None))
            out[buf.value.decode()] = arr
        return out

    def save(self, path):
        """Network::save (network/mod.rs:16-18; net6_simhash.rs:152-170): `*.tzw` = the flat container, else a LibTorch
        archive as tch writes it (+ `bitvec.bin` beside it for SimHash nets)."""
        check(self.lib.tz_net_save(self.h, str(path).encode()))

    def load_prepare(self, path):
        """First half of a load (tz_net_load_prepare): parse `path` and build its device weights in fresh buffers; the live
        network is not touched, so this may run on another thread while the net is evaluating.  Returns a handle for load_commit."""
        h = C.c_void_p()
        check(self.lib.tz_net_load_prepare(self.h, os.fsencode(str(path)), C.byref(h)))
        return h

    def load_commit(self, pending):
        """Second half (tz_net_load_commit): swap the prepared weights in; not concurrently with a forward of this net."""
        check(self.lib.tz_net_load_commit(self.h, pending))
        return self

    def load_partial(self, path):
        """Network::load_partial (network/mod.rs:30-35): variables missing from the file keep their current values;
        returns their names (tch's VarStore::load_partial)."""
        buf, n = C.create_string_buffer(1 << 16), C.c_int()
        check(self.lib.tz_net_load_partial(self.h, str(path).encode(), buf, len(buf), C.byref(n)))
        return [x for x in buf.value.decode().split("\n") if x]

    def clone(self, device=0):
        """Network::clone(device) (network/mod.rs:37-44): the same weights on another GPU."""
        h = C.c_void_p()
        check(self.lib.tz_net_clone(self.h, device, C.byref(h)))
        other = Net.__new__(Net)
        other.lib, other.h = self.lib, h
        other.arch, other.precision, other.blocks, other.device, other.n = self.arch, self.precision, self.blocks, device, self.n
        return other

    def policy_value_uncertainty(self, env_batch, actions_batch):
        """Agent::policy_value_uncertainty: returns (list of per-env logits arrays, values, variances)."""
        st = _states(env_batch)
        b = len(st)
        if b == 0 or len(actions_batch) != b:
            raise TakzeroError(-1, "env_batch and actions_batch must be non-empty and of equal length")
        amax = max(1, max(len(a) for a in actions_batch))
        idx = np.zeros((b, amax), np.uint16)
        cnt = np.zeros(b, np.int32)
        for i, a in enumerate(actions_batch):
            cnt[i] = len(a)
            idx[i, :len(a)] = a
        logits = np.zeros((b, amax), np.float32)
        value = np.zeros(b, np.float32)
        var = np.zeros(b, np.float32)
        check(self.lib.tz_net_eval(self.h, b, st.ctypes.data, idx.ctypes.data, cnt.ctypes.data, amax,
                                   logits.ctypes.data, value.ctypes.data, var.ctypes.data))
        return [logits[i, :cnt[i]].copy() for i in range(b)], value, var

    def hash_indices(self, env_batch, update=False):
        """HashNetwork::get_indices (and update_counts when update=True), net6_simhash.rs:202-243."""
        st = _states(env_batch)
        out = np.zeros(len(st), np.uint32)
        check(self.lib.tz_net_hash_indices(self.h, len(st), st.ctypes.data, out.ctypes.data, 1 if update else 0))
        return out

    def load_bitset(self, path):
        check(self.lib.tz_net_load_bitset(self.h, str(path).encode()))

    def save_bitset(self, path):
        check(self.lib.tz_net_save_bitset(self.h, str(path).encode()))

    def encode(self, env_batch):
        st = _states(env_batch)
        out = np.zeros((len(st), input_channels(self.n) * self.n * self.n), np.float32)
        check(self.lib.tz_net_encode(self.h, len(st), st.ctypes.data, out.ctypes.data))
        return out

    def forward_raw(self, env_batch):
        st = _states(env_batch)
        b = len(st)
        pol = np.zeros((b, policy_size(self.n)), np.float32)
        val = np.zeros(b, np.float32)
        ube = np.zeros(b, np.float32)
        check(self.lib.tz_net_forward_raw(self.h, b, st.ctypes.data, pol.ctypes.data, val.ctypes.data, ube.ctypes.data))
        return pol, val, ube


class BatchedMCTS:
    """BatchedMCTS<BATCH_SIZE, Game<N, HALF_KOMI>> on one GPU."""

    def __init__(self, batch, n, half_komi, agent=None, agent_kind=None, node_capacity=0):
        self.lib = _lib.load()
        self.batch, self.n, self.half_komi = batch, n, half_komi
        if agent_kind is None:
            agent_kind = AGENT_NET if agent is not None else AGENT_DUMMY
        self.agent = agent
        h = C.c_void_p()
        check(self.lib.tz_search_create(agent.h if agent is not None else None, agent_kind, batch, n, half_komi,
                                        node_capacity, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_search_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- BatchedMCTS::new / from_envs / nodes_and_envs_mut
    def new_openings(self, opening_choice):
        c = np.ascontiguousarray(opening_choice, dtype=np.int32)
        assert c.shape == (self.batch,)
        check(self.lib.tz_search_new_openings(self.h, c.ctypes.data))

    def set_positions(self, game_idx, states):
        idx = np.ascontiguousarray(game_idx, dtype=np.int32)
        st = _states(states)
        assert len(idx) == len(st)
        check(self.lib.tz_search_set_positions(self.h, len(idx), idx.ctypes.data, st.ctypes.data))

    def get_positions(self):
        out = np.zeros(self.batch, STATE_DTYPE)
        check(self.lib.tz_search_get_positions(self.h, out.ctypes.data))
        return out

    # --- search
    def simulate(self, betas, n_sims=1):
        b = np.ascontiguousarray(betas, dtype=np.float32)
        assert b.shape == (self.batch,)
        check(self.lib.tz_search_simulate(self.h, b.ctypes.data, n_sims))

    def apply_noise(self, noise, ratio):
        nz = np.ascontiguousarray(noise, dtype=np.float32)
        assert nz.ndim == 2 and nz.shape[0] == self.batch
        check(self.lib.tz_search_apply_noise(self.h, nz.ctypes.data, nz.shape[1], ratio))

    def gumbel_sequential_halving(self, betas, sampled_actions, search_budget, gumbel):
        b = np.ascontiguousarray(betas, dtype=np.float32)
        gm = np.ascontiguousarray(gumbel, dtype=np.float32)
        out = np.zeros(self.batch, np.uint16)
        check(self.lib.tz_search_gumbel_sh(self.h, b.ctypes.data, sampled_actions, search_budget, gm.ctypes.data,
                                           gm.shape[1], out.ctypes.data))
        return out

    # --- results
    def root_info(self):
        out = np.zeros(self.batch, ROOT_INFO_DTYPE)
        check(self.lib.tz_search_root_info(self.h, out.ctypes.data))
        return out

    def root_children(self, amax=None):
        if amax is None:
            amax = max(1, int(self.root_info()["n_children"].max()))
        B = self.batch
        out = dict(move_idx=np.zeros((B, amax), np.uint16), visits=np.zeros((B, amax), np.uint32),
                   eval_tag=np.zeros((B, amax), np.uint8), eval_bits=np.zeros((B, amax), np.uint32),
                   logit=np.zeros((B, amax), np.float32), prob=np.zeros((B, amax), np.float32),
                   std_dev=np.zeros((B, amax), np.float32))
        check(self.lib.tz_search_root_children(self.h, amax, *[out[k].ctypes.data for k in (
            "move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev")]))
        return out

    def node(self, game, path, amax=None):
        """The node reached from game `game`'s root by the moves of `path` (Node.children below the root, node/mod.rs:14-23):
        (info record, children dict as root_children for that one node)."""
        if amax is None:
            amax = 512 if self.n < 6 else 1024
        p = np.ascontiguousarray(path, dtype=np.uint16)
        info = np.zeros(1, ROOT_INFO_DTYPE)
        out = dict(move_idx=np.zeros(amax, np.uint16), visits=np.zeros(amax, np.uint32), eval_tag=np.zeros(amax, np.uint8),
                   eval_bits=np.zeros(amax, np.uint32), logit=np.zeros(amax, np.float32), prob=np.zeros(amax, np.float32),
                   std_dev=np.zeros(amax, np.float32))
        check(self.lib.tz_search_node(self.h, game, p.ctypes.data if len(p) else None, len(p), info.ctypes.data, amax,
                                      *[out[k].ctypes.data for k in ("move_idx", "visits", "eval_tag", "eval_bits", "logit", "prob", "std_dev")]))
        nc = int(info[0]["n_children"])
        return info[0], {k: v[:nc] for k, v in out.items()}

    def select_best_actions(self):
        out = np.zeros(self.batch, np.uint16)
        check(self.lib.tz_search_select_best_actions(self.h, out.ctypes.data))
        return out

    def select_actions_in_selfplay(self, rng, weighted_random_steps, threshold=32, allowed_eval_drop=0.5):
        """batched.rs:165-183 / node/mod.rs:170-207 with a numpy Generator in place of the Rust rng
        (the draw itself is caller-side randomness; the candidate weights follow the reference)."""
        best = self.select_best_actions()
        info = self.root_info()
        sample = (info["ply"] < weighted_random_steps) & (info["eval_tag"] == EVAL_VALUE) & (info["n_children"] > 0)
        if not sample.any():
            return best
        ch = self.root_children()
        amax = ch["visits"].shape[1]
        valid = np.arange(amax)[None, :] < info["n_children"][:, None]
        keys = eval_sort_keys(ch["eval_tag"], ch["eval_bits"])
        best_key = np.where(valid, keys, np.inf).min(axis=1)
        bi = np.where(valid, keys, np.inf).argmin(axis=1)
        rows = np.arange(self.batch)
        best_is_value = ch["eval_tag"][rows, bi] == EVAL_VALUE
        drop = (ch["eval_bits"][rows, bi].view(np.float32) + np.float32(allowed_eval_drop)).astype(np.float64)
        limit = np.where(best_is_value, drop, best_key)
        w = np.where(valid & (ch["visits"] >= threshold) & (ch["eval_tag"] != EVAL_WIN) & (keys <= limit[:, None]),
                     ch["visits"], 0).astype(np.float64)
        tot = w.sum(axis=1)
        pick = sample & (tot > 0)           # InsufficientNonZero -> best action
        if not pick.any():
            return best
        u = rng.random(self.batch) * tot
        idx = (np.cumsum(w, axis=1) > u[:, None]).argmax(axis=1)
        out = best.copy()
        out[pick] = ch["move_idx"][rows[pick], idx[pick]]
        return out

    def improved_policy(self, visitations, amax=None):
        if amax is None:
            amax = max(1, int(self.root_info()["n_children"].max()))
        out = np.zeros((self.batch, amax), np.float32)
        check(self.lib.tz_search_improved_policy(self.h, visitations, amax, out.ctypes.data))
        return out

    def improved_policy_each(self, visitations, amax=None):
        """improved_policy with one visitation count per game (reanalyze/src/main.rs:196-202)."""
        if amax is None:
            amax = max(1, int(self.root_info()["n_children"].max()))
        v = np.ascontiguousarray(visitations, np.float32)
        out = np.zeros((self.batch, amax), np.float32)
        check(self.lib.tz_search_improved_policy_each(self.h, v.ctypes.data, amax, out.ctypes.data))
        return out

    def ube_target(self, beta):
        out = np.zeros(self.batch, np.float32)
        check(self.lib.tz_search_ube_target(self.h, beta, out.ctypes.data))
        return out

    # --- stepping
    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.uint16)
        assert a.shape == (self.batch,)
        check(self.lib.tz_search_step(self.h, a.ctypes.data))

    def restart_terminal_envs(self, opening_choice):
        c = np.ascontiguousarray(opening_choice, dtype=np.int32)
        out = np.zeros(self.batch, np.int8)
        check(self.lib.tz_search_restart_terminal(self.h, c.ctypes.data, out.ctypes.data))
        return out

    def terminal_details(self):
        reason = np.zeros(self.batch, np.int8)
        winner = np.zeros(self.batch, np.uint8)
        check(self.lib.tz_search_terminal_details(self.h, reason.ctypes.data, winner.ctypes.data))
        return reason, winner

    def play_moves(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.uint16)
        ok = np.zeros(self.batch, np.int8)
        check(self.lib.tz_search_play_moves(self.h, a.ctypes.data, ok.ctypes.data))
        return ok

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.lib.tz_search_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pool_usage(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(self.lib.tz_search_pool_usage(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pool_overflows(self):
        """expansions skipped because a game's node pool was full (see tz_search_pool_overflows)"""
        c = C.c_uint64()
        check(self.lib.tz_search_pool_overflows(self.h, C.byref(c)))
        return c.value

    def sync(self):
        check(self.lib.tz_search_sync(self.h))

    def profile(self, reset=0):
        cm, cl, tm, st = C.c_double(), C.c_uint64(), C.c_double(), C.c_uint64()
        check(self.lib.tz_search_profile(self.h, reset, C.byref(cm), C.byref(cl), C.byref(tm), C.byref(st)))
        return dict(conv_ms=cm.value, conv_launches=cl.value, tree_ms=tm.value, steps=st.value)


def eval_sort_keys(tags, bits):
    """Total order of Eval (eval.rs:138-163) as float64 keys (vectorised): Loss(p) < Value / Draw < Win(p);
    a faster loss is smaller, a faster win is larger, draws sit at CONTEMPT with slower draws first."""
    tags = np.asarray(tags)
    b = np.asarray(bits).astype(np.uint32)
    v = b.view(np.float32).astype(np.float64)
    p = b.astype(np.float64)
    return np.select([tags == EVAL_LOSS, tags == EVAL_WIN, tags == EVAL_DRAW],
                     [-1e9 + p, 1e9 - p, np.float64(np.float32(-0.05)) - p * 1e-12], default=v)
