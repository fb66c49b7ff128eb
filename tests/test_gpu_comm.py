"""The library's RCCL communicator on the GPU box (one GPU: world 1 — ncclCommInitRank, ncclAllGather and ncclBroadcast really
run, through librccl opened with dlopen) and the model hand-over on top of it.  The packing / ordering of N > 1 ranks is
covered on the CPU by tests/test_host_over_oracle.py (world 2, "fs" transport, same code above the transport)."""
import numpy as np
import pytest

from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


def test_rccl_communicator_of_one_rank(tmp_path):
    A = require_gpu()
    from takzero_amd import comm as CM
    from takzero_amd import weights as W

    c = CM.Comm.rccl(CM.unique_id(), 0, 1, 0)
    assert c.info()["transport"] == "rccl" and c.info()["world"] == 1
    blob = bytes(np.random.default_rng(0).integers(0, 256, 100_003, dtype=np.uint8))
    assert c.all_gather(blob) == [blob] and c.all_gather(b"") == [b""]
    assert c.broadcast(blob) == blob
    c.barrier()
    assert c.info()["collectives"] == 2 and c.info()["bytes_gathered"] == len(blob)
    # Net::load handed over (status 0 = the root has a new model): a no-op for the root itself, and the driver keeps playing
    net = A.Net(arch=A.ARCH_TEST, n=4, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=4, blocks=1, seed=1))
    c.broadcast_net(net, 0, 0)
    c.broadcast_net(net, 0, 1)
    from takzero_amd.selfplay import NativeSelfPlay

    mcts = A.BatchedMCTS(8, 4, 4, agent=net, node_capacity=1 << 12)
    sp = NativeSelfPlay(mcts, 8, seed=1, search="puct")
    sp.set_comm(c, 0)
    for _ in range(30):
        sp.play_move()
        sp.exchange()
    assert sp.take_text(1).count(b"\n") == sp.counters()["replays"]
    sp.close()
    c.close()
    # the id can also travel through a directory (what examples/selfplay_cli.cpp --comm rccl does)
    c2 = CM.Comm.rccl_from_directory(tmp_path, 0, 1, 0)
    assert (tmp_path / "rccl_id.bin").stat().st_size == CM.ID_BYTES and c2.all_gather(b"xy") == [b"xy"]
    c2.close()
