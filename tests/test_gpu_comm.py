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
