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
