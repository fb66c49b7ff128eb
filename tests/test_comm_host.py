"""The shared-directory transport of csrc/tz_comm.cpp from Python (no GPU): the N > 1 rank logic that RCCL carries on a node."""
import pytest


def test_fs_transport_from_python_threads(tmp_path):
    """Two ranks of the shared-directory transport in one process (threads): variable sizes incl. empty, broadcast."""
    import threading

    from takzero_amd import comm as CM

    got = {}

    def rank(r):
        c = CM.Comm.fs(tmp_path, r, 2, 30.0)
        got[r] = [c.all_gather(b"a" * (r * 5)), c.all_gather(bytes([r]) * (7 - r)), c.broadcast(b"hello" if r == 1 else b"xxxxx", root=1)]
        c.barrier()
        c.close()

    ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got[0] == got[1] == [[b"", b"aaaaa"], [b"\x00" * 7, b"\x01" * 6], b"hello"]
