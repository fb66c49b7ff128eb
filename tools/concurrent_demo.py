"""learn and selfplay as two processes on one GPU and one directory, concurrently, the way the reference deploys them
(README: 1 learn + N selfplay + N reanalyze processes): the directory protocol under real concurrency — targets appended
while learn tails them, model_latest.ot replaced while selfplay reloads it.  `python tools/concurrent_demo.py <role> <dir>`
with role learn | selfplay | reanalyze; tools/concurrent_demo.sh starts all and prints progress."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A
from takzero_amd import learn as L
from takzero_amd import runner as R
from takzero_amd import weights as W

role, d = sys.argv[1], sys.argv[2]
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
t0 = time.time()


def log(msg):
    print("[%s %6.1fs] %s" % (role, time.time() - t0, msg), flush=True)


if role == "learn":
    trainer = L.Trainer(arch=A.ARCH_NET5).load_tensors(W.init_weights(W.ARCH_NET5, seed=1))
    dummy = A.BatchedMCTS(2048, 5, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    steps = 0

    def step_log(msg):
        global steps
        steps += 1
        if steps % 100 == 0:
            log(msg + "  buffers " + open(os.path.join(d, "buffer_lengths.txt")).read())
        if time.time() - t0 > seconds:
            raise KeyboardInterrupt

    try:
        L.run_learn(d, trainer, seed=1, pre_train_mcts=dummy, pre_training_steps=100, initial_targets=100 * 128, min_selfplay=3000,
                    steps_before_reanalyze=400, min_reanalyze=1000, read_interval=2.0, sleep=1.0, log=step_log)
    except KeyboardInterrupt:
        log("done after %d logged steps" % steps)
else:
    net = A.Net.new(arch=A.ARCH_NET5, seed=1)
    mcts = A.BatchedMCTS(2048, 5, 4, agent=net)
    while not os.path.exists(os.path.join(d, "model_latest.ot")) or not os.path.exists(os.path.join(d, "buffer_lengths.txt")):
        time.sleep(0.5)
    # one long native run each (the progress lines come from tools/concurrent_demo.sh watching the files)
    if role == "selfplay":
        sp = R.run_selfplay(d, mcts, 384, moves=int(seconds / 0.9), seed=3, search="gumbel", native=True, max_wait=120)
        log("done: %s, simulations %d" % (sp.counters(), mcts.counters()[0]))
    else:
        try:
            R.run_reanalyze(d, mcts, 384, iterations=int(seconds / 3.0), seed=4, search="gumbel", native=True, min_positions=8192,
                            max_wait=seconds)
            log("done: simulations %d" % mcts.counters()[0])
        except TimeoutError as e:
            log("stopped: %s" % e)
