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


def test_fs_transport_ignores_what_an_earlier_job_left_in_the_directory(tmp_path):
    """selfplay's data directory outlives a job: a restarted job must not take the previous one's rendezvous or exchange files for
    its own (csrc/tz_comm.cpp: rank 0 replaces xch-job.bin and the exchange files carry the job's nonce; a file much older than the
    process is not accepted at all)."""
    import os
    import threading
    import time

    from takzero_amd import comm as CM

    d = str(tmp_path)
    # leftovers: a job file from long ago, exchange files of round 0 under the old naming and under another nonce
    stale = os.path.join(d, "xch-job.bin")
    open(stale, "wb").write(b"\x11" * 8)
    old = time.time() - 10000
    os.utime(stale, (old, old))
    for name in ("xch-0-0.bin", "xch-0-1.bin", "xch-1111111111111111-0-0.bin", "xch-1111111111111111-0-1.bin"):
        open(os.path.join(d, name), "wb").write(b"\x07" * 8)
    out = [None, None]

    def rank(r):
        if r == 1:
            time.sleep(0.3)          # rank 1 looks first at a directory that still holds the stale job file ... (rank 0 is late)
        c = CM.Comm.fs(d, r, 2, timeout_s=30)
        out[r] = c.all_gather(b"fresh-%d" % r)
        c.close()

    # rank 1 starts first and must wait for rank 0's fresh file instead of taking the stale one
    t1 = threading.Thread(target=rank, args=(1,))
    t1.start()
    time.sleep(0.1)
    t0 = threading.Thread(target=rank, args=(0,))
    t0.start()
    t0.join(60)
    t1.join(60)
    assert out[0] == out[1] == [b"fresh-0", b"fresh-1"], out


def test_rendezvous_id_file_is_replaced_and_a_stale_one_is_not_taken(tmp_path):
    import ctypes as C
    import os
    import time

    from takzero_amd import _lib

    lib = _lib.load()
    d = str(tmp_path)
    path = os.path.join(d, "rccl_id.bin")
    open(path, "wb").write(b"\x22" * 128)
    old = time.time() - 10000
    os.utime(path, (old, old))
    buf = (C.c_ubyte * 128)()
    rc = lib.tz_comm_rendezvous_id(d.encode(), 1, buf, C.c_double(0.3))
    assert rc != 0 and b"older than this process" in lib.tz_last_error()     # rank 1 does not take the dead job's id
    os.utime(path, None)                                                       # a fresh file is taken
    assert lib.tz_comm_rendezvous_id(d.encode(), 1, buf, C.c_double(5.0)) == 0 and bytes(buf) == b"\x22" * 128
