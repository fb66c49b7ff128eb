"""Times the learn step at the reference's configuration: net5, 20 blocks, batch 128
(learn/src/main.rs:43).  `python tools/learn_bench.py [steps] [batch]`.  The CPU figure quoted beside it comes from
tests/learn_cpu_time.py (the PyTorch fp32 restatement is test infrastructure and is not imported here)."""
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
    if "--loop" in sys.argv:
        # learn::main end to end: target file tailing, sampling with forced uses, augmentation, dense tensors, step,
        # model files every 100 steps — on a directory holding 20 000 random-game targets
        import tempfile

        from takzero_amd import formats as F

        d = tempfile.mkdtemp()
        while len(targets) < 20000:
            targets.extend(sp.play_move()[0])
        with open(os.path.join(d, "targets-selfplay.txt"), "w") as f:
            f.write(F.format_targets(n, targets))
        loop_steps = 350
        stamps = []
        run = L.run_learn_native if "--native" in sys.argv else L.run_learn
        line["loop_driver"] = "native" if "--native" in sys.argv else "python"
        run(d, tr, steps=loop_steps, seed=1, pre_train_mcts=None, min_selfplay=10000,
                    steps_before_reanalyze=10 ** 9, read_interval=10.0, sleep=0.01, max_wait=60,
                    log=lambda msg: stamps.append(time.perf_counter()))
        dt_loop = (stamps[-1] - stamps[49]) / (len(stamps) - 50)   # steady state: after start-up saves and the first read
        import numpy as _np

        gaps = _np.diff(_np.array(stamps[49:])) * 1e3
        line["loop_ms_per_step_median"] = float(_np.median(gaps))
        line["loop_ms_slowest_steps"] = [(int(i) + 50, round(float(gaps[i]), 1)) for i in _np.argsort(gaps)[-8:]]
        line["loop_ms_per_step"] = dt_loop * 1e3
        line["loop_steps_per_s"] = 1.0 / dt_loop
    print(json.dumps(line))


if __name__ == "__main__":
    main()
