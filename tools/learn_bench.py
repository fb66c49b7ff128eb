"""Times the learn step at the reference's configuration: net5, 20 blocks, batch 128
(learn/src/main.rs:43).  `python tools/learn_bench.py [steps] [batch] [--cpu]`; --cpu also times the PyTorch fp32
CPU restatement (oracle/learn_torch.py) on the host cores for a few steps."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import takzero_amd.api as A  # noqa: E402
from takzero_amd import learn as L  # noqa: E402
from takzero_amd import weights as W  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(args[0]) if args else 30
    B = int(args[1]) if len(args) > 1 else 128
    n = 5
    w = W.init_weights(W.ARCH_NET5, seed=123)
    tr = L.Trainer(arch=A.ARCH_NET5, batch=B).load_tensors(w)
    dummy = A.BatchedMCTS(B, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    from takzero_amd.selfplay import SelfPlay

    sp = SelfPlay(dummy, 0, seed=1, search="random")
    targets = []
    while len(targets) < B:
        targets.extend(sp.play_move()[0])
    tensors = L.target_tensors(targets[:B], n)
    for _ in range(3):
        tr.step(*tensors, train_ube=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.step(*tensors, train_ube=True)
    dt = (time.perf_counter() - t0) / steps
    flop = 3 * 1.1975e9 * B  # forward + data gradient + weight gradient of every conv / linear (RND is not trained)
    line = {"metric": "learn steps/s", "value": 1.0 / dt, "ms_per_step": dt * 1e3, "batch": B, "positions_per_s": B / dt,
            "tflops": flop / dt / 1e12, "losses": losses, "dtype": "f32"}
    if "--cpu" in sys.argv:
        import torch

        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import learn_torch as LT
        import oracle_lib as O

        oracle = O.load()
        planes = np.stack([O.game_repr(oracle, O.TzState.from_buffer_copy(s.tobytes())) for s in tensors[0]]).reshape(B, -1, n, n)
        p = LT.make_params(w)
        opt = LT.adam(p, 1e-4)
        tt = [torch.from_numpy(planes), torch.from_numpy(tensors[2].astype(bool)), torch.from_numpy(tensors[1]),
              torch.from_numpy(tensors[3]), torch.from_numpy(tensors[4])]
        times = []
        for i in range(4):
            t1 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            ls, _ = LT.losses(p, *tt, 20, True)
            (ls[0] + ls[1] + ls[2]).backward()
            opt.step()
            times.append(time.perf_counter() - t1)
        line["cpu_ms_per_step"] = min(times[1:]) * 1e3
        line["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(line))


if __name__ == "__main__":
    main()
