"""A/B of the net kernel's launch forms: board-major rows with full-size workgroups (TZ_NET_ROWS=board TZ_NET_P=full: every
(tap, row tile) pair issued) against the square-major order with zero-tap skipping, and against the default, which also
spreads small batches over 1- and 2-board workgroups.  Runs itself once per form in a child process and compares the raw
outputs bit for bit.
  python tools/net_rows_ab.py [n ...]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, out):
    import takzero_amd.api as A
    from takzero_amd import weights as W

    arch = {3: A.ARCH_TEST, 4: A.ARCH_TEST, 5: A.ARCH_NET5, 6: A.ARCH_NET6_SIMHASH}[n]
    res = {}
    for prec, pname in ((A.PREC_BF16, "bf16"), (A.PREC_F16, "f16")):
        if arch == A.ARCH_TEST:
            net = A.Net(arch=arch, n=n, blocks=3, precision=prec).load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=3, seed=5))
        else:
            net = A.Net.new(arch=arch, seed=5, precision=prec)
        # ragged: counts that are not multiples of the boards per workgroup; from 2048 positions on 6x6 runs 8-board workgroups
        # (compact tap table, ring loop)
        for count in (1, 7, 16, 333, 515, 1030) + ((2048, 2051) if n == 6 else ()):
            dummy = A.BatchedMCTS(count, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 8)
            rng = np.random.default_rng(count)
            dummy.new_openings(rng.integers(0, 16, count))
            for _ in range(6):      # a few random plies so that stacks, walls and capstones appear
                dummy.simulate(np.zeros(count, np.float32), 4)
                dummy.step(dummy.select_best_actions())
                dummy.restart_terminal_envs(rng.integers(0, 16, count))
            states = dummy.get_positions()
            raw = net.forward_raw(states)
            for k, v in zip(("policy", "value", "ube"), raw):
                res["%s_%d_%s" % (pname, count, k)] = np.asarray(v)
            # the Agent surface too: its variance carries the RND networks' output (net5)
            _, _, var = net.policy_value_uncertainty(states, [np.zeros(1, np.uint16)] * count)
            res["%s_%d_variance" % (pname, count)] = np.asarray(var)
    np.savez(out, **res)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), sys.argv[3])
    sizes = [int(a) for a in sys.argv[1:]] or [3, 5, 6]
    bad = 0
    for n in sizes:
        outs = []
        forms = (("board-major, full workgroups", {"TZ_NET_ROWS": "board", "TZ_NET_P": "full"}),
                 ("square-major, full workgroups", {"TZ_NET_ROWS": "square", "TZ_NET_P": "full"}),
                 ("default", {}))
        for i, (name, extra) in enumerate(forms):
            out = "/tmp/net_rows_%d_%d.npz" % (i, n)
            env = {k: v for k, v in os.environ.items() if k not in ("TZ_NET_ROWS", "TZ_NET_P")}
            env.update(extra)
            subprocess.run([sys.executable, __file__, "--child", str(n), out], check=True, env=env)
            outs.append(np.load(out))
        for i in (1, 2):
            for k in outs[0].files:
                same = np.array_equal(outs[0][k].view(np.uint8), outs[i][k].view(np.uint8))
                if not same:
                    bad += 1
                    d = np.abs(outs[0][k].astype(np.float64) - outs[i][k].astype(np.float64))
                    print("n=%d %s: %s differs from %s: max abs %.3g" % (n, k, forms[i][0], forms[0][0], np.nanmax(d)))
        print("n=%d: %d arrays x 2 forms compared" % (n, len(outs[0].files)), flush=True)
    print("BIT-IDENTICAL" if not bad else "%d arrays differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
