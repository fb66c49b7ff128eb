"""The library's RCCL communicator on the GPU box (one GPU: world 1 — ncclCommInitRank, ncclAllGather and ncclBroadcast really
run, through librccl opened with dlopen) and the model hand-over on top of it.  The packing / ordering of N > 1 ranks is
covered on the CPU by tests/test_host_over_oracle.py (world 2, "fs" transport, same code above the transport)."""
import numpy as np
import pytest

from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


def test_rccl_communicator_in_clean_processes():
    """Which RCCL the library opens (csrc/tz_comm.cpp rccl()): alone in the process the system's librccl.so.1; after
    `import torch` (bench.py's order under torchrun) PyTorch's own copy, so that one RCCL and one HIP runtime serve both."""
    import os
    import subprocess
    import sys

    require_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for order, lib in (("none", "/opt/rocm"), ("torch_first", "torch/lib/librccl.so")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_copies_check.py"), order], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (order, r.stdout[-500:], r.stderr[-1500:])
        maps = [ln for ln in r.stdout.splitlines() if ln.startswith(order)]
        assert len(maps) == 1 and maps[0].count("librccl") == 1 and lib in maps[0], r.stdout
        assert "[b'abc']" in r.stdout and "'transport': 'rccl'" in r.stdout


def test_torch_imported_after_the_communicator_does_not_abort():
    """The load order that maps two RCCL copies (the library opens the system's, `import torch` brings PyTorch's afterwards) used to
    abort at interpreter exit with a double free: the library's copy is opened RTLD_LOCAL, so the two do not interpose."""
    import os
    import subprocess
    import sys

    require_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_copies_check.py"), "torch_after"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "[b'abc']" in r.stdout and "'transport': 'rccl'" in r.stdout


def test_rccl_communicator_of_one_rank(tmp_path):
    """ncclCommInitRank / ncclAllGather / ncclBroadcast at world 1, the model hand-over and the self-play driver on top of
    them: in a process of its own (tests/comm_world1_script.py) that imports torch first, as bench.py does under torchrun — this
    pytest process has loaded the library before torch and would end up with two RCCL copies (see tools/rccl_copies_check.py)."""
    import os
    import subprocess
    import sys

    require_gpu()
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "comm_world1_script.py"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COMM-WORLD1-OK" in r.stdout, (r.stdout[-800:], r.stderr[-2000:])


def test_model_hand_over_carries_the_simhash_set(tmp_path):
    """tz_net_broadcast for a SimHash net at world 2 (two ranks as threads of this process over the fs transport — the code above
    the transport is what RCCL carries): rank 0's load replaces its set of seen hashes with the model's bitvec.bin
    (net6_simhash.rs:173-190), so the set must reach rank 1 with the variables: afterwards rank 1 reports rank 0's variances, bit
    for bit, for seen and unseen positions alike.  A root without weights fails on both ranks instead of leaving one in the collective."""
    import threading

    import oracle_lib as O
    from gpu_util import random_positions

    A = require_gpu()
    from takzero_amd import comm as CM
    from takzero_amd import weights as W

    oracle = O.load()
    states = random_positions(oracle, O, 4, 4, 24, 5, max_ply=20)
    arr = O.states_array(states)
    acts = [O.possible_moves(oracle, s) for s in states]
    nets = [A.Net(arch=A.ARCH_NET4_SIMHASH, precision=A.PREC_BF16).load_tensors(W.init_weights(W.ARCH_NET4_SIMHASH, seed=3 + r)) for r in (0, 1)]
    empty = A.Net(arch=A.ARCH_NET4_SIMHASH, precision=A.PREC_BF16)
    nets[0].hash_indices(arr[:12], update=True)
    nets[1].hash_indices(arr[12:], update=True)          # rank 1's own set: replaced, not merged
    want = nets[0].policy_value_uncertainty(arr, acts)
    assert np.any(want[2] < 4.0) and np.any(want[2] == 4.0)
    assert not np.array_equal(nets[1].policy_value_uncertainty(arr, acts)[2], want[2])
    errors = {}

    def rank(r):
        c = CM.Comm.fs(tmp_path, r, 2, 120.0)
        c.broadcast_net(nets[r], 0, 0)
        try:
            c.broadcast_net(empty if r == 0 else nets[1], 0, 0)
        except A.TakzeroError as e:
            errors[r] = str(e)
        c.barrier()
        c.close()

    ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    got = nets[1].policy_value_uncertainty(arr, acts)
    assert all(np.array_equal(g, w) for g, w in zip(got[0], want[0]))      # the variables arrived: same logits
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])   # ... and the set: same variances
    assert set(errors) == {0, 1} and all("no weights" in e for e in errors.values()), errors
    for n in nets + [empty]:
        n.close()
