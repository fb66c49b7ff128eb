"""Checker script (test infrastructure): seconds per learn step of the PyTorch fp32 restatement (oracle/learn_torch.py)
on the host cores, net5 at batch 128 — the CPU figure quoted beside tools/learn_bench.py.  Needs no GPU."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import learn_torch as LT  # noqa: E402
import oracle_lib as O  # noqa: E402
from gpu_util import random_positions  # noqa: E402
from takzero_amd import weights as W  # noqa: E402

B, n, blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 5, 20
oracle = O.load()
rng = np.random.default_rng(0)
states = random_positions(oracle, O, n, 4, B, 1, max_ply=30)
planes = np.stack([O.game_repr(oracle, s) for s in states]).reshape(B, -1, n, n)
out = (3 + 4 * (2 ** n - 2)) * n * n
policy, mask = np.zeros((B, out), np.float32), np.ones((B, out), bool)
for i, s in enumerate(states):
    mv = np.array(O.possible_moves(oracle, s), np.int64)
    policy[i, mv] = 1.0 / len(mv)
    mask[i, mv] = False
w = W.init_weights(W.ARCH_NET5, seed=123)
p = LT.make_params(w)
opt = LT.adam(p, 1e-4)
tt = [torch.from_numpy(planes), torch.from_numpy(mask), torch.from_numpy(policy),
      torch.from_numpy(rng.uniform(-1, 1, B).astype(np.float32)), torch.from_numpy(rng.uniform(0.1, 4, B).astype(np.float32))]
times = []
for i in range(4):
    t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    ls, _ = LT.losses(p, *tt, blocks, True)
    (ls[0] + ls[1] + ls[2]).backward()
    opt.step()
    times.append(time.perf_counter() - t1)
print(json.dumps({"cpu_ms_per_step": min(times[1:]) * 1e3, "cpu_threads": torch.get_num_threads(), "batch": B}))
