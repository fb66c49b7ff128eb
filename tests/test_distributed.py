"""world_size-2 gloo test of the only collective on the path: the all-gather of finished targets."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from takzero_amd import selfplay as SP
    from takzero_amd._lib import STATE_DTYPE

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(rank)
    targets = []
    for i in range(3 + 2 * rank):  # ragged: ranks contribute different counts
        st = np.zeros(1, STATE_DTYPE)[0]
        st["ply"] = 10 * rank + i
        st["n"] = 5
        k = int(rng.integers(1, 60))
        targets.append((st, rng.integers(0, 3075, k).astype(np.uint16), rng.random(k).astype(np.float32),
                        float(rank) + 0.5, float(i)))
    got = SP.all_gather_targets(targets, 5)
    empty = SP.all_gather_targets([], 5)  # nobody finished a game this move
    q.put((rank, [(int(t[0]["ply"]), t[1].tolist(), t[2].tolist(), t[3], t[4]) for t in got], len(empty)))
    dist.destroy_process_group()


def test_all_gather_targets_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1]  # every rank sees the same gathered set, in rank order
    assert len(res[0][1]) == 3 + 5 and res[0][2] == 0 and res[1][2] == 0
    plies = [t[0] for t in res[0][1]]
    assert plies == [0, 1, 2, 10, 11, 12, 13, 14]
    assert all(abs(t[3] - (0.5 if t[0] < 10 else 1.5)) < 1e-6 for t in res[0][1])


def test_pack_unpack_round_trip():
    sys.path.insert(0, ROOT)
    from takzero_amd import selfplay as SP
    from takzero_amd._lib import STATE_DTYPE

    rng = np.random.default_rng(0)
    st = np.zeros(1, STATE_DTYPE)[0]
    st["colors"][3] = 5
    st["n"] = 6
    t = [(st, rng.integers(0, 9000, 700).astype(np.uint16), rng.random(700).astype(np.float32), -0.25, 1.5)]
    back = SP.unpack_targets(SP.pack_targets(t, 6), 6)
    assert back[0][0].tobytes() == st.tobytes() and np.array_equal(back[0][1], t[0][1]) and np.array_equal(back[0][2], t[0][2])
    assert back[0][3] == -0.25 and back[0][4] == 1.5


class _RecordingNet:
    def __init__(self):
        self.loads = []

    def load_tensors(self, tensors):
        self.loads.append({k: v.copy() for k, v in tensors.items()})


def _reload_worker(rank, world, port, directory, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from takzero_amd import runner as R
    from takzero_amd import weights as W

    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _RecordingNet()
    # only rank 0 can see the model: the other rank watches an empty directory and must get the tensors by broadcast
    watcher = R.BroadcastModelWatcher(net, directory if rank == 0 else directory + "-nothing-here", rank, name="model_latest.tzw")
    events = []
    try:
        watcher.refresh()
    except OSError:
        events.append("missing")           # no model yet: every rank raises (nobody hangs in the collective)
    dist.barrier()
    if rank == 0:
        W.save_tzw(os.path.join(directory, "model_latest.tzw"),
                   {"a.weight": np.arange(6, dtype=np.float32).reshape(2, 3), "b": np.float32([7.5]), "s": np.float32(3.0)})
    dist.barrier()
    events.append(watcher.refresh())       # True: loaded
    events.append(watcher.refresh())       # False: unchanged
    dist.barrier()
    if rank == 0:
        W.save_tzw(os.path.join(directory, "model_latest.tzw"), {"a.weight": np.full((2, 3), 2.0, np.float32), "c": np.zeros(5, np.float32)})
    dist.barrier()
    events.append(watcher.refresh())
    q.put((rank, events, [{k: (v.shape, v.tolist()) for k, v in load.items()} for load in net.loads]))
    dist.destroy_process_group()


def test_model_reload_is_broadcast_from_rank0(tmp_path):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reload_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == ["missing", True, False, True]
    assert res[0][2] == res[1][2] and len(res[0][2]) == 2
    first, second = res[1][2]
    assert first["a.weight"] == ((2, 3), [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]) and first["b"] == ((1,), [7.5])
    assert set(second) == {"a.weight", "c"} and second["c"][0] == (5,)


def _torn_model_worker(rank, world, port, directory, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from takzero_amd import ot
    from takzero_amd import runner as R
    from takzero_amd import weights as W

    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _RecordingNet()
    watcher = R.BroadcastModelWatcher(net, directory if rank == 0 else directory + "-nothing-here", rank)
    open(os.path.join(directory, "buffer_lengths.txt"), "w").write("0,0,0")
    events = []
    w = W.init_weights(W.ARCH_TEST, n=3, blocks=1, seed=1)
    if rank == 0:      # a model read while `learn` is half way through writing it (the reference's save is not atomic)
        ot.save_ot(os.path.join(directory, "good.ot"), w)
        blob = open(os.path.join(directory, "good.ot"), "rb").read()
        open(os.path.join(directory, "model_latest.ot"), "wb").write(blob[:len(blob) // 2])
    dist.barrier()
    for _ in range(2):
        # selfplay's inner loop (which = 0): an archive that does not parse is "not retrying": EVERY rank keeps its net and
        # goes on to its move, so the next collective is the same on all ranks (the all-gather below)
        R.wait_until_needed(directory, 0, R.MAX_SELFPLAY_BUFFER_LEN, watcher, sleep=0.01, max_wait=5.0)
        events.append(("kept", len(net.loads)))
        gathered = [None, None]
        dist.all_gather_object(gathered, rank)
        events.append(tuple(gathered))
    try:               # called directly: the same exception class on every rank (the torn file is not re-read until it changes)
        if rank == 0:
            os.utime(os.path.join(directory, "model_latest.ot"), ns=(1, 1))
        watcher.refresh()
    except R.ModelParseError:
        events.append("parse-error")
    dist.barrier()
    if rank == 0:
        os.replace(os.path.join(directory, "good.ot"), os.path.join(directory, "model_latest.ot"))
    dist.barrier()
    R.wait_until_needed(directory, 0, R.MAX_SELFPLAY_BUFFER_LEN, watcher, sleep=0.01, max_wait=5.0)
    events.append(("loaded", len(net.loads), sorted(net.loads[-1]) == sorted(w)))
    q.put((rank, events))
    dist.destroy_process_group()


def test_a_torn_model_file_keeps_every_rank_on_the_same_branch(tmp_path):
    """rank 0 finds a truncated model_latest.ot.  Every rank must raise the same exception class, take the same
    branch of the callers and stay matched in the next collective, then pick up the repaired file together."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_torn_model_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [("kept", 0), (0, 1), ("kept", 0), (0, 1), "parse-error", ("loaded", 1, True)]
    assert res[0][1] == want and res[1][1] == want


def _append_worker(rank, path, chunks):
    sys.path.insert(0, ROOT)
    from takzero_amd import runner as R

    for i in range(chunks):
        lines = ["rank%d chunk%05d line%03d %s\n" % (rank, i, j, "x" * (37 + (i * 7 + j * 13) % 5000)) for j in range(1 + (i % 17))]
        R.append_lines(path, lines)


def test_ranks_appending_to_one_file_never_interleave_inside_a_line(tmp_path):
    """ADVICE r1 (rank-suffixed files nobody read): every rank now appends to the shared un-suffixed files, as the
    reference's N processes do.  Four writers, chunks up to ~70 KB, one write() per chunk on an O_APPEND descriptor: the
    reader must find exactly the union of whole lines, and each writer's lines in its own order."""
    import multiprocessing as mp

    path = str(tmp_path / "targets-selfplay.txt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_append_worker, args=(r, path, 300)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    lines = open(path).read().split("\n")
    assert lines[-1] == "" and len(lines) - 1 == 4 * sum(1 + (i % 17) for i in range(300))
    per_rank = {r: [] for r in range(4)}
    for ln in lines[:-1]:
        head, chunk, line, fill = ln.split(" ")
        i, j = int(chunk[5:]), int(line[4:])
        assert fill == "x" * (37 + (i * 7 + j * 13) % 5000), "two writers' bytes inside one line"
        per_rank[int(head[4:])].append((i, j))
    for r in range(4):
        assert per_rank[r] == sorted(per_rank[r]) and len(per_rank[r]) == sum(1 + (i % 17) for i in range(300))
