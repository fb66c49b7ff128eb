"""Config 4's net kernel (6x6, net6_simhash, 2048 positions per launch) in its two workgroup forms: 8 boards (compact tap table,
ring loop; the default from 2048 positions on) and 4 boards (TZ_NET_P6=4).  One child process per form (the switch is read once);
prints ms per launch from the engine's HIP events on the live search and the algorithmic fraction of the 2.5 PFLOP/s peak."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOP = 1.4067e9 - 0.0  # net6_simhash per position; the SimHash projection is a separate kernel


def child(games, sims):
    import takzero_amd.api as A
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_NET6_SIMHASH, precision=A.PREC_NAMES[os.environ.get("TZ_PRECISION", "f16")])
    net.load_tensors(W.init_weights(W.ARCH_NET6_SIMHASH, seed=123))
    mcts = A.BatchedMCTS(games, 6, 4, agent=net, node_capacity=1 << 15)
    mcts.new_openings(np.arange(games) % 16)
    betas = np.zeros(games, np.float32)
    mcts.simulate(betas, 8)
    mcts.sync()
    res = []
    for _ in range(3):
        mcts.profile(reset=1)
        s0, e0 = mcts.counters()
        mcts.simulate(betas, sims)
        mcts.sync()
        p = mcts.profile(reset=2)
        s1, e1 = mcts.counters()
        ms = p["conv_ms"] / max(1, p["conv_launches"])
        per_launch = (e1 - e0) / sims
        res.append({"ms_per_launch": round(ms, 4), "positions_per_launch": round(per_launch, 1),
                    "algorithmic_frac_of_2.5PF": round(per_launch * FLOP / (ms * 1e-3) / 2.5e15, 4)})
    print(json.dumps(res))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    out = {}
    for name, extra in (("8 boards per workgroup (default)", {}), ("4 boards per workgroup (TZ_NET_P6=4)", {"TZ_NET_P6": "4"})):
        env = {k: v for k, v in os.environ.items() if k != "TZ_NET_P6"}
        env.update(extra)
        r = subprocess.run([sys.executable, __file__, "--child", str(games), str(sims)], capture_output=True, text=True, env=env, check=True)
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
