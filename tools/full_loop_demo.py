"""The three binaries at production size on one directory and one GPU, run back to back (not concurrently): learn
(pre-training on random games, model files in the reference's .ot format), selfplay (native driver, hot reload of
model_latest.ot), learn again on the self-play targets, selfplay again with the new model.  Prints what each phase did."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import formats as F
from takzero_amd import learn as L
from takzero_amd import runner as R
from takzero_amd import weights as W

d = tempfile.mkdtemp()
out = {"directory": d}
t0 = time.perf_counter()
trainer = L.Trainer(arch=A.ARCH_NET5).load_tensors(W.init_weights(W.ARCH_NET5, seed=1))
dummy = A.BatchedMCTS(4096, 5, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
net = A.Net.new(arch=A.ARCH_NET5, seed=1)
mcts = A.BatchedMCTS(4096, 5, 4, agent=net)
open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
out["setup_s"] = time.perf_counter() - t0

# learn: no targets yet -> initial model + pre-training, then model_latest.ot; 0 main-loop steps
t0 = time.perf_counter()
steps = L.run_learn(d, trainer, steps=0, seed=1, pre_train_mcts=dummy, pre_training_steps=60, initial_targets=60 * 128,
                    read_interval=0.0, sleep=0.01, max_wait=30)
out["learn_pretraining"] = {"model_steps": steps, "seconds": time.perf_counter() - t0,
                            "files": sorted(f for f in os.listdir(d) if f.endswith(".ot"))}
# selfplay: picks up model_latest.ot, 14 moves of Gumbel 768 / k 64
t0 = time.perf_counter()
sp = R.run_selfplay(d, mcts, 768, moves=14, seed=2, search="gumbel", native=True, max_wait=30)
sims, _ = mcts.counters()
out["selfplay_1"] = {"seconds": time.perf_counter() - t0, "sims_per_s": sims / (time.perf_counter() - t0), **sp.counters()}
# learn on what self-play wrote
t0 = time.perf_counter()
have = sum(1 for _ in open(os.path.join(d, "targets-selfplay.txt")))
steps = L.run_learn(d, trainer, steps=100, seed=3, min_selfplay=min(have, 256), steps_before_reanalyze=10 ** 9, read_interval=0.0,
                    sleep=0.01, max_wait=30)
out["learn_main"] = {"model_steps": steps, "seconds": time.perf_counter() - t0, "targets_available": have,
                     "buffer_lengths": open(os.path.join(d, "buffer_lengths.txt")).read()}
# selfplay again: the watcher sees a new model_latest.ot
t0 = time.perf_counter()
s0, _ = mcts.counters()
sp2 = R.run_selfplay(d, mcts, 768, moves=4, seed=4, search="gumbel", native=True, max_wait=30)
s1, _ = mcts.counters()
out["selfplay_2"] = {"seconds": time.perf_counter() - t0, "sims_per_s": (s1 - s0) / (time.perf_counter() - t0), **sp2.counters()}
out["files"] = {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}
print(json.dumps(out))
