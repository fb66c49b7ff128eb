"""The precision question on a net5 that the repo's own closed loop has trained (">= 5 000 steps of the repo's own
closed loop"): self-play (Gumbel sequential halving on 4096 games, fp16 network) feeds learn (native step, batch 128), the trainer's
weights go back into the self-play network every `sync` steps; after `steps` training steps the errors of f16 / f16x2 / bf16 against
the library's fp32 path are measured on positions of fresh self-play games, with the logit scale the trained net has reached.

    python tools/trained_net_precision.py [steps=5000] [sync=250]        -> one JSON object (profiles/r02_trained_net_precision.json)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import takzero_amd.api as A  # noqa: E402
from takzero_amd import learn as L  # noqa: E402
from takzero_amd import precision as P  # noqa: E402
from takzero_amd.selfplay import NativeSelfPlay  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    sync = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    t0 = time.time()
    net = A.Net.new(arch=A.ARCH_NET5, seed=11)
    trainer = L.Trainer(arch=A.ARCH_NET5, batch=128).from_net(net)
    mcts = A.BatchedMCTS(4096, 5, 4, agent=net, node_capacity=1 << 14)
    sp = NativeSelfPlay(mcts, 64, seed=5, search="gumbel", sampled_actions=16)
    loop = L.NativeLearnLoop(trainer, half_komi=4, seed=3)
    done, losses, targets = 0, None, 0
    history = []
    while done < steps:
        while loop.buffer_len(0) < 128 * 40:            # keep learn's buffer fed (every target is used 4 times)
            sp.play_move()
            text = sp.take_text(0)
            sp.take_text(1)
            targets += text.count(b"\n")
            loop.add_lines(0, text, done)
        for _ in range(sync):
            if loop.buffer_len(0) < 128:
                break
            losses = loop.step(using_reanalyze=False, train_ube=True, augment=True)
            done += 1
        trainer.to_net(net)                              # model_latest, without the file
        history.append((done, [round(x, 4) for x in losses]))
        print("step %d losses %r buffer %d targets %d (%.0f s)" % (done, losses, loop.buffer_len(0), targets, time.time() - t0), file=sys.stderr, flush=True)
    sp.close()
    mcts.close()
    import tempfile

    from takzero_amd import weights as W

    with tempfile.TemporaryDirectory() as tmp:           # the trainer's VarStore incl. the RND side networks it carries along
        trainer.save(os.path.join(tmp, "trained.tzw"))
        weights = W.load_tzw(os.path.join(tmp, "trained.tzw"))
    # held-out positions: fresh games played by the trained net
    fresh = A.BatchedMCTS(256, 5, 4, agent=net, node_capacity=1 << 12)
    sp2 = NativeSelfPlay(fresh, 48, seed=99, search="gumbel", sampled_actions=8)
    for _ in range(14):
        sp2.play_move()
    states = fresh.get_positions().copy()
    sp2.close()
    fresh.close()
    net.close()
    report = P.errors_against_f32(A.ARCH_NET5, weights, states, precisions=("f16", "f16c8", "f16x2", "bf16"), legal=P.legal_mask(states, 5))
    out = {"net": "net5 trained by the repo's own closed loop (self-play Gumbel 64 / k 16 on 4096 games -> learn, batch 128)",
           "training_steps": done, "targets_generated": targets, "loss_history": history[::max(1, len(history) // 10)],
           "seconds": round(time.time() - t0, 1), "errors_vs_fp32_path": report}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
