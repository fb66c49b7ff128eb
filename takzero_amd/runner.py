"""The directory protocol of the reference's `selfplay` and `reanalyze` binaries, so that an unmodified `learn`
process can sit on the other side of the same directory:

  buffer_lengths.txt     "selfplay,reanalyze,sum" written by learn; we pause while our buffer is over the cap
                         (selfplay/src/main.rs:90-105, 371-387; reanalyze/src/main.rs:78-91)
  model_latest.ot        re-read before every move / iteration (selfplay/src/main.rs:107-121)
  targets-selfplay.txt, replays.txt, targets-reanalyze.txt   appended (selfplay/src/main.rs:332-366,
                         reanalyze/src/main.rs:230-243)

With more than one rank every rank appends to the SAME un-suffixed files, as the reference's N processes in one directory do
(README.md:130): what a move finished goes out as one write() of whole lines on an O_APPEND descriptor, so ranks never
interleave inside a line and `learn` / `reanalyze` read everybody's output.  With `gather` (Python driver) or a communicator
(native driver, NativeSelfPlay.set_comm) the targets are all-gathered first and rank 0 alone writes them."""
import os
import time

import numpy as np

from . import formats
from .reanalyze import Reanalyze
from .selfplay import SelfPlay, all_gather_targets

MAX_SELFPLAY_BUFFER_LEN = 32_000   # selfplay/src/main.rs:43
MAX_REANALYZE_BUFFER_LEN = 32_000  # reanalyze/src/main.rs:40
MIN_POSITIONS = 4000 * 128 // 4    # reanalyze/src/main.rs:38


def _lib_error():
    from ._lib import TakzeroError

    return TakzeroError


def read_buffer_lengths(directory):
    with open(os.path.join(directory, "buffer_lengths.txt")) as f:
        return formats.parse_buffer_lengths(f.read())


class ModelParseError(ValueError):
    """model_latest.ot is there but does not parse (e.g. read while `learn` was writing it)."""


class ModelWatcher:
    """Net::load(directory/model_latest.ot) before every search.  The reference re-reads the file every time;
    we re-read it only when its (mtime, size, inode) changed, which gives the same network."""

    def __init__(self, net, directory, name="model_latest.ot"):
        self.net, self.path, self.stamp, self.reloads = net, os.path.join(directory, name), None, 0

    def refresh(self):
        """True if a (new) model is loaded; raises OSError if the file is missing (caller retries)."""
        st = os.stat(self.path)
        stamp = (st.st_mtime_ns, st.st_size, st.st_ino)   # writers rename a fresh file into place: the inode changes too
        if stamp != self.stamp:
            self.net.load(self.path)
            self.stamp = stamp
            self.reloads += 1
            return True
        return False


class BroadcastModelWatcher(ModelWatcher):
    """Weights reload for one process per GPU: rank 0 watches the file; when it changed, its tensors
    go to every rank as one flat fp32 broadcast (RCCL on the GPU box, gloo in the CPU tests) and each rank loads them
    into its own device copy.  Every rank calls refresh() once per move, so the two collectives stay matched."""

    def __init__(self, net, directory, rank, name="model_latest.ot", device=None):
        super().__init__(net, directory, name)
        self.rank, self.device, self.layout = rank, device, None

    def refresh(self):
        import torch
        import torch.distributed as dist

        from . import ot

        tensors, changed, err = None, 0, None
        if self.rank == 0:
            try:
                st = os.stat(self.path)
                stamp = (st.st_mtime_ns, st.st_size, st.st_ino)   # writers rename a fresh file into place: the inode changes too
                if stamp != self.stamp:
                    tensors = ot.load_ot(self.path) if self.path.endswith(".ot") else None
                    if tensors is None:
                        from . import weights

                        tensors = weights.load_tzw(self.path)
                    self.stamp, changed = stamp, 1
            except OSError as e:     # missing / unreadable file: every rank retries (selfplay/src/main.rs:116-119)
                err, changed = e, -1
            except Exception as e:   # an archive that does not parse (torn write): every rank keeps its net (:112-115)
                err, changed = e, -2
                self.stamp = stamp   # not looked at again until the file changes
        dev = self.device if self.device is not None else "cpu"
        flag = torch.tensor([changed], dtype=torch.int64, device=dev)
        dist.broadcast(flag, src=0)
        # the same exception class on every rank, so that all of them take the same branch in the callers and the next
        # collective matches (rank 0 re-raising a parse error while the others raised OSError hung the job)
        if int(flag.item()) == -1:
            raise OSError("rank 0 could not read the model: %s" % (err if err is not None else "see rank 0"))
        if int(flag.item()) == -2:
            raise ModelParseError("rank 0 could not parse the model: %s" % (err if err is not None else "see rank 0"))
        if int(flag.item()) == 0:
            return False
        if self.rank == 0:
            names = sorted(tensors)
            meta = [(k, tuple(tensors[k].shape)) for k in names]
            flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(tensors[k], np.float32).ravel() for k in names]))
        else:
            meta, flat = None, None
        box = [meta]
        dist.broadcast_object_list(box, src=0)
        meta = box[0]
        total = sum(int(np.prod(shape)) if shape else 1 for _, shape in meta)
        buf = flat.to(dev) if self.rank == 0 else torch.empty(total, dtype=torch.float32, device=dev)
        dist.broadcast(buf, src=0)
        host = buf.cpu().numpy()
        out, off = {}, 0
        for k, shape in meta:
            cnt = int(np.prod(shape)) if shape else 1
            out[k] = host[off:off + cnt].reshape(shape).copy()
            off += cnt
        self.net.load_tensors(out)
        self.reloads += 1
        return True


def wait_until_needed(directory, which, cap, watcher, sleep=1.0, max_wait=None, log=None):
    """The inner `loop` of both binaries: block while learn's buffer for `which` (0 selfplay, 1 reanalyze) is over
    `cap`, then (re)load the model.  A model that cannot be parsed is kept as is for selfplay ("not retrying",
    selfplay/src/main.rs:112-115) and retried for reanalyze (reanalyze/src/main.rs:98-101)."""
    t0 = time.monotonic()

    def expired():
        return max_wait is not None and time.monotonic() - t0 > max_wait

    while True:
        try:
            length = read_buffer_lengths(directory)[which]
        except (OSError, ValueError) as err:
            if log:
                log("Could not read buffer lengths: %s" % err)
            if expired():
                raise TimeoutError("buffer_lengths.txt unreadable for %.0f s" % max_wait)
            time.sleep(sleep)
            continue
        if length > cap:
            if expired():
                raise TimeoutError("buffer over its cap for %.0f s" % max_wait)
            time.sleep(sleep)
            continue
        if watcher is None:
            return
        try:
            watcher.refresh()
            return
        except OSError as err:  # missing file: "some other reason, retrying"
            if log:
                log("Cannot load model: %s, retrying." % err)
            if expired():
                raise TimeoutError("no model for %.0f s" % max_wait)
            time.sleep(sleep)
        except Exception as err:  # archive present but unreadable
            if log:
                log("Cannot load model (parse error): %s" % err)
            if which == 0:
                return
            if expired():
                raise
            time.sleep(sleep)


class AsyncAppender:
    """One writer thread that formats and appends what the search loop hands over, in order.  Formatting thousands of
    target lines per move is native code (tz_format_targets) and the search itself sits inside ctypes calls, both
    with the interpreter lock released, so the file work overlaps the next move's search instead of delaying it."""

    def __init__(self):
        import queue
        import threading

        self.q = queue.Queue(maxsize=8)   # bounded: back-pressure instead of unbounded memory if the disk stalls
        self.error = None
        self.thread = threading.Thread(target=self._run, name="tz-appender", daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            try:
                if self.error is None:
                    job()
            except Exception as e:   # surfaced by the next submit() / close() on the search thread
                self.error = e

    def submit(self, job):
        if self.error is not None:
            raise self.error
        self.q.put(job)

    def close(self):
        self.q.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error


def append_lines(path, lines):
    """OpenOptions::append(true).create(true) + one write of everything: several processes append to the same file
    (one per GPU here, 10 selfplay + 10 reanalyze processes in the reference's deployment) and a reader must never see
    two writers' bytes interleaved inside a line."""
    if not lines or not any(lines):
        return
    data = "".join(lines)
    data = data.encode() if isinstance(data, str) else data
    fd = os.open(path, os.O_WRONLY | os.O_APPEND | os.O_CREAT, 0o644)
    try:
        view = memoryview(data)
        while len(view):                     # a regular file takes it whole; loop only for the sake of the contract
            view = view[os.write(fd, view):]
    finally:
        os.close(fd)


def run_selfplay(directory, mcts, sims_per_move, moves=None, seed=0, rank=0, world=1, search="gumbel",
                 sampled_actions=64, gather=False, watch_model=True, exploration=False, broadcast_model=False,
                 native=False, sleep=1.0, max_wait=None, log=None):
    """selfplay::main (selfplay/src/main.rs:63-205) for `moves` outer iterations (None = forever).
    exploration = the cargo feature of that name: the first half of the games search with beta = 0.25 and the
    openings of those games also go to replays-exploration.txt (:79-86, 279-290).
    broadcast_model: only rank 0 reads model_latest.ot, the other ranks receive the tensors over torch.distributed.
    native: run the loop in native code (tz_selfplay_run, csrc/tz_host.cpp) instead of this Python mirror of it."""
    n = mcts.n
    if native:
        # the whole loop below the ABI (csrc/tz_host.cpp): search, bookkeeping, formatting, file appends, back-pressure
        from .selfplay import NativeSelfPlay

        if gather:
            raise ValueError("native run: hand the shards a communicator instead (NativeSelfPlay.set_comm), rank 0 then writes")
        sp = NativeSelfPlay(mcts, sims_per_move, seed=seed, shard=rank, search=search, sampled_actions=sampled_actions,
                            exploration=exploration)
        watcher = None
        if watch_model:
            watcher = BroadcastModelWatcher(mcts.agent, directory, rank) if broadcast_model and world > 1 else \
                ModelWatcher(mcts.agent, directory)

        def reload():
            if watcher is None:
                return
            t0 = time.monotonic()
            while True:   # missing file: retry (selfplay/src/main.rs:116-119); unreadable archive: keep the old net (:112-115)
                try:
                    watcher.refresh()
                    return
                except OSError as err:
                    if log:
                        log("Cannot load model: %s, retrying." % err)
                    if max_wait is not None and time.monotonic() - t0 > max_wait:
                        raise TimeoutError("no model for %.0f s" % max_wait)
                    time.sleep(sleep)
                except Exception as err:
                    if log:
                        log("Cannot load model (parse error): %s" % err)
                    return

        try:
            sp.run(directory, moves=moves, max_buffer_len=MAX_SELFPLAY_BUFFER_LEN, suffix="",
                   reload=reload if watcher is not None else None, wait_limit_s=-1.0 if max_wait is None else float(max_wait))
        except _lib_error() as e:
            if e.code == -6:   # TZ_ESTATE: buffer_lengths.txt stayed unreadable / over the cap for max_wait
                raise TimeoutError(str(e))
            raise
        return sp
    betas = None
    if exploration:
        from .selfplay import BETA

        betas = np.where(np.arange(mcts.batch) < mcts.batch // 2, BETA, 0.0).astype(np.float32)
    sp = SelfPlay(mcts, sims_per_move, seed=seed, shard=rank, search=search, sampled_actions=sampled_actions, betas=betas)
    watcher = None
    if watch_model:
        watcher = BroadcastModelWatcher(mcts.agent, directory, rank) if broadcast_model and world > 1 else \
            ModelWatcher(mcts.agent, directory)
    writer = AsyncAppender()

    def write(targets, replays, expl):
        # every rank appends to the shared files (one write per move); with `gather` rank 0 holds everybody's targets
        if not gather or rank == 0:
            append_lines(os.path.join(directory, "targets-selfplay.txt"), [formats.format_targets(n, targets)])
        append_lines(os.path.join(directory, "replays.txt"), [formats.format_replay(n, *r) for r in replays])
        if exploration:
            append_lines(os.path.join(directory, "replays-exploration.txt"), [formats.format_replay(n, *r) for r in expl])

    step = 0
    try:
        while moves is None or step < moves:
            wait_until_needed(directory, 0, MAX_SELFPLAY_BUFFER_LEN, watcher, sleep, max_wait, log)
            targets, replays = sp.play_move()
            if gather and world > 1:
                targets = all_gather_targets(targets, n)   # collectives stay on the search thread
            expl, sp.exploration_replays = sp.exploration_replays, []
            writer.submit(lambda t=targets, r=replays, e=expl: write(t, r, e))
            step += 1
    finally:
        writer.close()
    return sp


def run_reanalyze(directory, mcts, sims, iterations=None, seed=0, rank=0, world=1, search="gumbel",
                  sampled_actions=64, min_positions=MIN_POSITIONS, watch_model=True, native=False, sleep=1.0, max_wait=None,
                  log=None):
    """reanalyze::main (reanalyze/src/main.rs:60-244) for `iterations` outer iterations (None = forever)."""
    n = mcts.n
    if native:   # the whole loop below the ABI (csrc/tz_host.cpp)
        from .reanalyze import NativeReanalyze

        nra = NativeReanalyze(mcts, sims, seed=seed, rank=rank, world=world, search=search, sampled_actions=sampled_actions)
        watcher = ModelWatcher(mcts.agent, directory) if watch_model else None

        def reload():
            t0 = time.monotonic()
            while True:   # reanalyze retries every kind of load failure (reanalyze/src/main.rs:93-105)
                try:
                    watcher.refresh()
                    return
                except Exception as err:
                    if log:
                        log("Cannot load model: %s, retrying." % err)
                    if max_wait is not None and time.monotonic() - t0 > max_wait:
                        raise
                    time.sleep(sleep)

        try:
            nra.run(directory, iterations=iterations, min_positions=min_positions, suffix="",
                    reload=reload if watcher is not None else None, wait_limit_s=-1.0 if max_wait is None else float(max_wait))
        except _lib_error() as e:
            if e.code == -6:
                raise TimeoutError(str(e))
            raise
        return nra
    ra = Reanalyze(mcts, sims, seed=seed, rank=rank, world=world, search=search, sampled_actions=sampled_actions)
    watcher = ModelWatcher(mcts.agent, directory) if watch_model else None
    it = 0
    t0 = time.monotonic()
    writer = AsyncAppender()
    path = os.path.join(directory, "targets-reanalyze.txt")   # shared by all ranks (one write per iteration)
    try:
        while iterations is None or it < iterations:
            wait_until_needed(directory, 1, MAX_REANALYZE_BUFFER_LEN, watcher, sleep, max_wait, log)
            try:
                ra.buffer.read_new(os.path.join(directory, "replays.txt"))
            except OSError as err:
                if log:
                    log("Cannot fill position buffer: %s" % err)
            if len(ra.buffer.positions) < max(min_positions, mcts.batch):
                if max_wait is not None and time.monotonic() - t0 > max_wait:
                    raise TimeoutError("not enough positions (%d)" % len(ra.buffer.positions))
                time.sleep(sleep)  # the reference sleeps 60 s here (reanalyze/src/main.rs:135-142)
                continue
            targets = ra.iterate()
            writer.submit(lambda t=targets: append_lines(path, [formats.format_targets(n, t)]))
            it += 1
    finally:
        writer.close()
    return ra
