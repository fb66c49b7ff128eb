"""Exercises the RCCL code paths (target all-gather, model broadcast) with a world of one rank on a GPU box."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

from takzero_amd import runner as R
from takzero_amd import selfplay as SP
from takzero_amd import weights as W
from takzero_amd._lib import STATE_DTYPE

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(0)
targets = []
for i in range(5):
    st = np.zeros(1, STATE_DTYPE)[0]
    st["ply"], st["n"] = i, 5
    k = int(rng.integers(1, 60))
    targets.append((st, rng.integers(0, 3075, k).astype(np.uint16), rng.random(k).astype(np.float32), 0.5, float(i)))
got = SP.all_gather_targets(targets, 5, "cuda:0")
assert len(got) == 5 and all(np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(got, targets))
assert SP.all_gather_targets([], 5, "cuda:0") == []


class Net:
    def load_tensors(self, t):
        self.t = t


d = tempfile.mkdtemp()
W.save_tzw(os.path.join(d, "model_latest.tzw"), {"a": np.arange(12, dtype=np.float32).reshape(3, 4), "b": np.float32([1.5])})
net = Net()
w = R.BroadcastModelWatcher(net, d, 0, name="model_latest.tzw", device="cuda:0")
assert w.refresh() and not w.refresh() and np.array_equal(net.t["a"], np.arange(12, dtype=np.float32).reshape(3, 4))
dist.destroy_process_group()
print("nccl world-1 paths ok")
