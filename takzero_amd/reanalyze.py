"""Reanalyze driver over BatchedMCTS — the outer loop of the reference's `reanalyze` binary
(reanalyze/src/main.rs:60-244): tail `replays.txt`, expand every replay into its pre-move positions
(Replay::states, target.rs:205-212), sample B positions, fresh trees, search, emit one target per position.

Replays are expanded on the device: a batch of replays is loaded as start positions and stepped with
validated moves (tz_search_play_moves), reading the positions back after every ply, so parsing never needs
host-side rules.  With N GPUs, replay line i belongs to rank i mod N."""
import numpy as np

from . import api, formats

BETA = 0.25  # reanalyze/src/main.rs:45


class PositionBuffer:
    """fill_buffer_with_positions_from_replays (reanalyze/src/main.rs:270-290)."""

    def __init__(self, mcts, n, half_komi, rank=0, world=1):
        self.mcts, self.n, self.half_komi, self.rank, self.world = mcts, n, half_komi, rank, world
        self.offset = 0          # byte offset into replays.txt (incremental seek)
        self.line_no = 0
        self.positions = []      # numpy tz_state records

    def read_new(self, path):
        with open(path, "rb") as f:
            f.seek(self.offset)
            data = f.read()
        last_nl = data.rfind(b"\n")
        if last_nl < 0:
            return 0
        self.offset += last_nl + 1
        replays = []
        for raw in data[:last_nl].split(b"\n"):
            mine = self.line_no % self.world == self.rank
            self.line_no += 1
            if not mine or not raw.strip():
                continue
            try:
                replays.append(formats.parse_replay(raw.decode(), self.n, self.half_komi))
            except Exception:
                continue  # unparsable lines are skipped, as filter_map(...ok()) does (target.rs:283-285)
        return self.expand(replays)

    def expand(self, replays):
        """All pre-move states of every replay (the final position is excluded); replays with an illegal move
        are dropped from that move on (reference: the whole line fails to parse)."""
        B, added = self.mcts.batch, 0
        for i in range(0, len(replays), B):
            chunk = replays[i:i + B]
            idx = np.arange(len(chunk))
            self.mcts.set_positions(idx, np.array([c[0] for c in chunk], dtype=api.STATE_DTYPE))
            alive = np.zeros(B, bool)
            alive[:len(chunk)] = True
            per_game = [[] for _ in chunk]
            ply = 0
            while True:
                acts = np.full(B, 0xFFFF, np.uint16)
                for g, (_, moves) in enumerate(chunk):
                    if alive[g] and ply < len(moves):
                        acts[g] = moves[ply]
                    else:
                        alive[g] = False
                if not alive.any():
                    break
                states = self.mcts.get_positions()
                ok = self.mcts.play_moves(acts)
                for g in np.nonzero(alive)[0]:
                    if ok[g] == 1:
                        per_game[g].append(states[g].copy())
                    else:
                        per_game[g] = None if ok[g] == 0 else per_game[g]
                        alive[g] = False
                ply += 1
            for states in per_game:
                if states:
                    self.positions.extend(states)
                    added += len(states)
        return added

    def sample(self, rng, count):
        """position_buffer.sample(rng, B) — without replacement (reanalyze/src/main.rs:154-158)."""
        pick = rng.choice(len(self.positions), size=count, replace=False)
        return np.array([self.positions[i] for i in pick], dtype=api.STATE_DTYPE)


class Reanalyze:
    def __init__(self, mcts, sims, seed=0, rank=0, world=1, search="puct", sampled_actions=64):
        self.mcts, self.sims, self.search, self.k = mcts, sims, search, sampled_actions
        self.rng = np.random.default_rng([seed, rank, 7])
        self.buffer = PositionBuffer(mcts, mcts.n, mcts.half_komi, rank, world)
        self.zero_beta = np.zeros(mcts.batch, np.float32)

    def iterate(self):
        """One outer-loop iteration (reanalyze/src/main.rs:146-228): returns B targets."""
        m, B = self.mcts, self.mcts.batch
        states = self.buffer.sample(self.rng, B)
        m.set_positions(np.arange(B), states)           # nodes reset, envs overwritten (:159-165)
        if self.search == "puct":
            m.simulate(self.zero_beta, self.sims)       # :167-170
            selected = m.select_best_actions()
        else:
            gumbel = self.rng.gumbel(size=(B, 512 if m.n < 6 else 1024)).astype(np.float32)
            selected = m.gumbel_sequential_halving(self.zero_beta, self.k, self.sims, gumbel)  # :171-177
        info = m.root_info()
        ch = m.root_children()
        amax = ch["visits"].shape[1]
        mvc = ch["visits"].max(axis=1).astype(np.float32)
        # improved_policy(most_visited_count()) per game: the visitation count differs per root (:196-202)
        if hasattr(m, "improved_policy_each"):
            pol = m.improved_policy_each(mvc, amax)
        else:   # a search object without the per-game entry point (the oracle wrapper in the tests)
            pol = np.zeros((B, amax), np.float32)
            for v in np.unique(mvc):
                rows = mvc == v
                pol[rows] = m.improved_policy(float(v), amax)[rows]
        ube = m.ube_target(BETA)                        # :203
        targets = []
        for g in range(B):
            nc = int(info["n_children"][g])
            if info["eval_tag"][g] != api.EVAL_VALUE:   # solved root: its own evaluation (:184-187)
                value = api.eval_to_f32(info["eval_tag"][g], info["eval_bits"][g])
            else:                                       # else the selected child's evaluation, negated, then converted (:188-195):
                j = int(np.nonzero(ch["move_idx"][g, :nc] == selected[g])[0][0])
                tag, bits = int(ch["eval_tag"][g, j]), ch["eval_bits"][g, j]
                if tag == api.EVAL_VALUE:
                    value = -api.eval_to_f32(tag, bits)
                else:                                   # Eval::negate flips a proven result and adds a ply (eval.rs:40-47)
                    flipped = {api.EVAL_WIN: api.EVAL_LOSS, api.EVAL_LOSS: api.EVAL_WIN, api.EVAL_DRAW: api.EVAL_DRAW}[tag]
                    value = api.eval_to_f32(flipped, int(bits) + 1)
            targets.append((states[g], ch["move_idx"][g, :nc].copy(), pol[g, :nc].copy(), float(value), float(ube[g])))
        return targets


class NativeReanalyze:
    """The same driver in native code (csrc/tz_host.cpp, tz_reanalyze_*): position buffer, sampling, search and target
    lines below the ABI; this class only forwards."""

    KINDS = {"puct": 0, "gumbel": 1}

    def __init__(self, mcts, sims, seed=0, rank=0, world=1, search="puct", sampled_actions=64):
        import ctypes as C

        from . import _lib

        self.mcts, self.lib = mcts, _lib.load()
        self.h = C.c_void_p()
        _lib.check(self.lib.tz_reanalyze_create(mcts.h, sims, seed, rank, world, self.KINDS[search], sampled_actions, C.byref(self.h)))
        self.positions = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_reanalyze_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, path):
        """fill_buffer_with_positions_from_replays: positions added from what was appended to `path`."""
        import ctypes as C

        from . import _lib

        added, total = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.tz_reanalyze_feed(self.h, str(path).encode(), C.byref(added), C.byref(total)))
        self.positions = total.value
        return added.value

    def iterate(self):
        from . import _lib

        _lib.check(self.lib.tz_reanalyze_iterate(self.h))

    def take_text(self):
        import ctypes as C

        from . import _lib

        size = C.c_uint64()
        self.lib.tz_reanalyze_take_text(self.h, None, 0, C.byref(size))
        if size.value == 0:
            return b""
        buf = C.create_string_buffer(size.value)
        _lib.check(self.lib.tz_reanalyze_take_text(self.h, buf, size.value, C.byref(size)))
        return buf.raw[:size.value]

    def run(self, directory, iterations=None, min_positions=0, suffix="", reload=None, wait_limit_s=-1.0):
        import ctypes as C

        from . import _lib

        cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p)
        failure = []

        def trampoline(_user):
            try:
                reload()
                return 0
            except Exception as e:
                failure.append(e)
                return -6

        cb = cb_type(trampoline) if reload is not None else None
        rc = self.lib.tz_reanalyze_run(self.h, str(directory).encode(), -1 if iterations is None else iterations, min_positions,
                                       suffix.encode(), C.cast(cb, C.c_void_p) if cb is not None else None, None, wait_limit_s)
        if failure:
            raise failure[0]
        _lib.check(rc)
