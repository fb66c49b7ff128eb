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
