"""Body of tests/test_gpu_comm.py::test_rccl_communicator_of_one_rank, run as a process of its own (torch imported first)."""
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import takzero_amd.api as A  # noqa: E402
from takzero_amd import comm as CM  # noqa: E402
from takzero_amd import weights as W  # noqa: E402
from takzero_amd.selfplay import NativeSelfPlay  # noqa: E402

tmp_path = sys.argv[1]
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
# ... and is gone once the communicator is up: a later job in the same directory must not find this job's id (csrc/tz_comm.cpp)
assert not os.path.exists(os.path.join(tmp_path, "rccl_id.bin")) and c2.all_gather(b"xy") == [b"xy"]
c2.close()
mcts.close()
net.close()
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
assert len(maps) == 1, maps
print("COMM-WORLD1-OK", maps)
