"""Self-play driver over BatchedMCTS — the outer loop of the reference's `selfplay` binary
(selfplay/src/main.rs:63-205, 238-329) for the classic PUCT + Dirichlet search the north star names
(the driver lines at selfplay/src/main.rs:127-136, live library code batched.rs:63-183) and for
Gumbel sequential halving (selfplay/src/main.rs:138-153).

One SelfPlay object drives one GPU's shard of games.  Randomness (openings, Dirichlet / Gumbel
noise, early-ply move sampling) is drawn here from a counter-keyed numpy generator so that a CPU
run of the oracle can be fed the same values."""
import time

import numpy as np

from . import api, formats

NOISE_ALPHA = 0.05   # selfplay/src/main.rs:39
NOISE_RATIO = 0.2    # selfplay/src/main.rs:40
WEIGHTED_RANDOM_PLIES = 10  # selfplay/src/main.rs:38
BETA = 0.25          # selfplay/src/main.rs:41


def dirichlet_rows(rng, counts, alpha, amax):
    """One symmetric Dir(alpha) sample per row, of dimension counts[row], zero padded to amax."""
    B = len(counts)
    g = rng.standard_gamma(alpha, size=(B, amax)).astype(np.float64)
    mask = np.arange(amax)[None, :] < np.asarray(counts)[:, None]
    g = np.where(mask, g, 0.0)
    s = g.sum(axis=1, keepdims=True)
    # a row whose gammas all underflowed: put the mass on one child (what a Dirichlet with tiny alpha tends to)
    dead = (s[:, 0] <= 0) & (np.asarray(counts) > 0)
    if dead.any():
        g[dead, 0] = 1.0
        s = g.sum(axis=1, keepdims=True)
    s[s == 0] = 1.0
    return (g / s).astype(np.float32)


class SelfPlay:
    def __init__(self, mcts, sims_per_move, seed=0, shard=0, betas=None, search="puct", sampled_actions=64,
                 collect_targets=True):
        self.mcts = mcts
        self.sims = sims_per_move
        self.search = search
        self.k = sampled_actions
        self.rng = np.random.default_rng([seed, shard])
        self.betas = np.zeros(mcts.batch, np.float32) if betas is None else np.asarray(betas, np.float32)
        self.collect = collect_targets
        self.history = []          # per-move IncompleteTarget arrays (selfplay/src/main.rs:230-236)
        self.history_base = 0      # move index of history[0]
        self.game_start = np.zeros(mcts.batch, np.int64)  # move index at which each game's current episode began
        self.start_states = None
        self.moves_played = 0
        self.positions = 0
        self.exploration_replays = []   # filled by _complete for games searched with beta > 0 (feature "exploration")
        self.host_s = {"search": 0.0, "record": 0.0, "step": 0.0, "complete": 0.0}  # wall seconds per phase
        mcts.new_openings(self.rng.integers(0, 16, mcts.batch))
        if self.collect:
            self.start_states = mcts.get_positions()

    def play_move(self):
        """One outer-loop iteration of selfplay::main.  Returns (finished_targets, finished_replays): targets are
        (state, moves, policy, value, ube), replays are (start_state, moves, PTN result)."""
        m, B = self.mcts, self.mcts.batch
        t0 = time.perf_counter()
        if self.search == "puct":
            m.simulate(self.betas, 1)                       # selfplay/src/main.rs:128
            info = m.root_info()
            amax = max(1, int(info["n_children"].max()))
            noise = dirichlet_rows(self.rng, info["n_children"], NOISE_ALPHA, amax)
            m.apply_noise(noise, NOISE_RATIO)               # :131
            m.simulate(self.betas, self.sims)               # :134-136
            actions = m.select_actions_in_selfplay(self.rng, WEIGHTED_RANDOM_PLIES)  # batched.rs:165-183
        elif self.search == "random":
            # uniformly random legal moves (learn's pre_training games, learn/src/main.rs:437-445): one simulation
            # expands the roots, which lists the legal moves
            m.simulate(self.betas, 1)
            ch, info = m.root_children(), m.root_info()
            j = np.minimum((self.rng.random(B) * info["n_children"]).astype(np.int64), info["n_children"] - 1)
            actions = ch["move_idx"][np.arange(B), j].astype(np.uint16)
        else:
            gumbel = self.rng.gumbel(size=(B, 512 if m.n < 6 else 1024)).astype(np.float32)
            actions = m.gumbel_sequential_halving(self.betas, self.k, self.sims, gumbel)  # :138-144
            early = m.root_info()["ply"] < WEIGHTED_RANDOM_PLIES                           # :145-153
            if early.any():
                sampled = m.select_actions_in_selfplay(self.rng, WEIGHTED_RANDOM_PLIES)
                actions = np.where(early, sampled, actions).astype(np.uint16)
        targets, replays = [], []
        t1 = time.perf_counter()
        if self.collect:
            self._record(actions)
        t2 = time.perf_counter()
        m.step(actions)                                     # take_a_step, :238-258
        term = m.restart_terminal_envs(self.rng.integers(0, 16, B))  # :263-329
        t3 = time.perf_counter()
        if self.collect:
            targets, replays = self._complete(term)
        t4 = time.perf_counter()
        for k, v in (("search", t1 - t0), ("record", t2 - t1), ("step", t3 - t2), ("complete", t4 - t3)):
            self.host_s[k] += v
        self.moves_played += 1
        self.positions += B
        return targets, replays

    # ---- target bookkeeping (host side: SURVEY.md §8a rows a22-a23).  One record per move for the whole
    # shard (arrays), resolved per game only when that game ends.
    def _record(self, actions):
        m = self.mcts
        info = m.root_info()
        ch = m.root_children()
        states = m.get_positions()
        if self.search == "puct":
            vis = ch["visits"].astype(np.float32)
            pol = vis / np.maximum(info["visit_count"].astype(np.float32), 1)[:, None]  # target.rs:151-164
        elif self.search == "random":
            pol = np.broadcast_to((np.float32(1.0) / np.maximum(info["n_children"], 1).astype(np.float32))[:, None],
                                  ch["visits"].shape).copy()  # uniform policy, learn/src/main.rs:451-454
        else:
            lg = int(np.log2(self.k))
            visitations = (self.sims // lg // self.k) * (2 ** lg - 1)  # selfplay/src/main.rs:47-52
            pol = m.improved_policy(float(visitations), ch["visits"].shape[1])
        if self.search == "random":
            ube = np.full(m.batch, np.float32(4.0) - np.finfo(np.float32).eps, np.float32)  # learn/src/main.rs:460
        else:
            ube = m.ube_target(BETA)
        stepped = ~((info["eval_tag"] != api.EVAL_VALUE) & (info["eval_bits"] == 0))  # batched.rs:137
        self.history.append(dict(states=states, moves=ch["move_idx"], pol=pol, ube=ube,
                                 nchild=info["n_children"].copy(), stepped=stepped, actions=np.array(actions)))

    def _complete(self, term):
        targets, replays = [], []
        done = np.nonzero(term != api.TERMINAL_NONE)[0]
        new_states = self.mcts.get_positions() if len(done) else None
        reason, winner = self.mcts.terminal_details() if len(done) else (None, None)
        for g in done:
            # value walks back from the terminal Eval, negating at every step (selfplay/src/main.rs:294-326)
            tag = {api.TERMINAL_WIN: api.EVAL_WIN, api.TERMINAL_LOSS: api.EVAL_LOSS, api.TERMINAL_DRAW: api.EVAL_DRAW}[int(term[g])]
            ply, acts = 0, []
            for h in reversed(self.history[self.game_start[g] - self.history_base:]):
                if not h["stepped"][g]:
                    continue
                tag = {api.EVAL_WIN: api.EVAL_LOSS, api.EVAL_LOSS: api.EVAL_WIN, api.EVAL_DRAW: api.EVAL_DRAW}[tag]
                ply += 1
                acts.append(int(h["actions"][g]))
                state = h["states"][g]
                if self.betas[g] == 0.0 or state["ply"] > WEIGHTED_RANDOM_PLIES:
                    k = int(h["nchild"][g])
                    targets.append((state.copy(), h["moves"][g, :k].copy(), h["pol"][g, :k].copy(),
                                    float(api.eval_to_f32(tag, ply)), float(h["ube"][g])))
            replays.append((self.start_states[g].copy(), acts[::-1], formats.result_string(reason[g], winner[g])))
            if self.betas[g] > 0.0:     # the opening of an exploratory game (selfplay/src/main.rs:279-290)
                self.exploration_replays.append((self.start_states[g].copy(), acts[::-1][:WEIGHTED_RANDOM_PLIES], None))
            self.game_start[g] = self.moves_played + 1
            self.start_states[g] = new_states[g]
        # drop history no running game refers to any more
        oldest = int(self.game_start.min())
        while self.history_base < oldest and self.history:
            self.history.pop(0)
            self.history_base += 1
        return targets, replays


# ---------------------------------------------------------------------------------------------
# fixed-stride packed target records for the RCCL all-gather into `learn`
def record_stride(n):
    amax = 512 if n < 6 else 1024
    return 376 + 4 + 4 + 4 + amax * 6


def pack_targets(targets, n):
    stride = record_stride(n)
    amax = 512 if n < 6 else 1024
    out = np.zeros((len(targets), stride), np.uint8)
    for i, (state, moves, pol, value, ube) in enumerate(targets):
        row = out[i]
        row[:376] = np.frombuffer(state.tobytes(), np.uint8)
        row[376:380] = np.frombuffer(np.float32(value).tobytes(), np.uint8)
        row[380:384] = np.frombuffer(np.float32(ube).tobytes(), np.uint8)
        k = min(len(moves), amax)
        row[384:388] = np.frombuffer(np.uint32(k).tobytes(), np.uint8)
        row[388:388 + 2 * k] = np.frombuffer(np.ascontiguousarray(moves[:k], np.uint16).tobytes(), np.uint8)
        off = 388 + 2 * amax
        row[off:off + 4 * k] = np.frombuffer(np.ascontiguousarray(pol[:k], np.float32).tobytes(), np.uint8)
    return out


def unpack_targets(buf, n):
    amax = 512 if n < 6 else 1024
    out = []
    for row in buf:
        state = np.frombuffer(row[:376].tobytes(), api.STATE_DTYPE)[0]
        value = np.frombuffer(row[376:380].tobytes(), np.float32)[0]
        ube = np.frombuffer(row[380:384].tobytes(), np.float32)[0]
        k = int(np.frombuffer(row[384:388].tobytes(), np.uint32)[0])
        moves = np.frombuffer(row[388:388 + 2 * k].tobytes(), np.uint16)
        off = 388 + 2 * amax
        pol = np.frombuffer(row[off:off + 4 * k].tobytes(), np.float32)
        out.append((state, moves, pol, float(value), float(ube)))
    return out


def all_gather_targets(targets, n, device=None):
    """All ranks contribute their finished targets; every rank receives all of them.  Collective:
    torch.distributed all_gather of counts, then of zero-padded fixed-stride records (NCCL = RCCL over
    xGMI on the GPU box, gloo on CPU).  No collective runs inside the search itself."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return targets
    world = dist.get_world_size()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    packed = pack_targets(targets, n)
    cnt = torch.tensor([len(targets)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if mx == 0:
        return []
    stride = record_stride(n)
    local = torch.zeros((mx, stride), dtype=torch.uint8, device=dev)
    if len(targets):
        local[:len(targets)] = torch.from_numpy(packed).to(dev)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    out = []
    for r in range(world):
        if counts[r]:
            out.extend(unpack_targets(gathered[r][:counts[r]].cpu().numpy(), n))
    return out


# ---------------------------------------------------------------------------------------------
class NativeSelfPlay:
    """The same driver in native code (csrc/tz_host.cpp, tz_selfplay_*): the whole outer loop of selfplay::main —
    search, move choice, target bookkeeping, line formatting — runs below the ABI; this class only forwards.
    Finished targets / replays come back as text lines (take_lines) or go straight to files (run)."""

    KINDS = {"puct": 0, "gumbel": 1, "random": 2}

    def __init__(self, mcts, sims_per_move, seed=0, shard=0, search="puct", sampled_actions=64, exploration=False):
        import ctypes as C

        from . import _lib

        self.mcts, self.lib = mcts, _lib.load()
        self.h = C.c_void_p()
        _lib.check(self.lib.tz_selfplay_create(mcts.h, sims_per_move, seed, shard, self.KINDS[search], sampled_actions,
                                               1 if exploration else 0, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_selfplay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def play_move(self):
        from . import _lib

        _lib.check(self.lib.tz_selfplay_play_move(self.h))

    def set_comm(self, comm, writer_rank=0):
        """N shards (tz_selfplay_set_comm): after every play_move, `exchange()` gathers all ranks' finished targets and
        replays; rank `writer_rank` (every rank if < 0) then holds everybody's lines for take_text / run."""
        from . import _lib

        _lib.check(self.lib.tz_selfplay_set_comm(self.h, comm.h if comm is not None else None, writer_rank))
        self._comm = comm   # keep the communicator alive as long as the driver refers to it

    def exchange(self):
        from . import _lib

        _lib.check(self.lib.tz_selfplay_exchange(self.h))

    def counters(self):
        import ctypes as C

        from . import _lib

        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.tz_selfplay_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"moves": a.value, "targets": b.value, "replays": c.value}

    def take_text(self, which=0):
        """bytes of the target (0) / replay (1) / exploration replay (2) lines finished since the last call."""
        import ctypes as C

        from . import _lib

        size = C.c_uint64()
        rc = self.lib.tz_selfplay_take_text(self.h, which, None, 0, C.byref(size))
        if size.value == 0:
            return b""
        buf = C.create_string_buffer(size.value)
        _lib.check(self.lib.tz_selfplay_take_text(self.h, which, buf, size.value, C.byref(size)))
        return buf.raw[:size.value]

    def run(self, directory, moves=None, max_buffer_len=32_000, suffix="", reload=None, wait_limit_s=-1.0):
        """tz_selfplay_run: the directory loop.  `reload` (optional) is called before every move and may swap weights."""
        import ctypes as C

        from . import _lib

        cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p)
        failure = []

        def trampoline(_user):
            try:
                reload()
                return 0
            except Exception as e:   # surfaces after the native loop returns
                failure.append(e)
                return -6

        cb = cb_type(trampoline) if reload is not None else None
        rc = self.lib.tz_selfplay_run(self.h, str(directory).encode(), -1 if moves is None else moves, max_buffer_len,
                                      suffix.encode(), C.cast(cb, C.c_void_p) if cb is not None else None, None, wait_limit_s)
        if failure:
            raise failure[0]
        _lib.check(rc)


def all_gather_bytes(data, device=None):
    """Every rank contributes a byte string (its finished target lines); every rank receives the concatenation in rank
    order.  Sizes first, then zero-padded payloads (NCCL = RCCL on the GPU box, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return data
    world = dist.get_world_size()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    size = torch.tensor([len(data)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    width = max(max(sizes), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    if data:
        mine[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    parts = [torch.zeros(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    return b"".join(bytes(p[:s].cpu().numpy().tobytes()) for p, s in zip(parts, sizes))
