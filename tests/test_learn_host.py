"""Host side of learn::main (learn/src/main.rs:99-319, 486-516): target buffers with forced uses, batch tensors,
model file discovery.  The GPU step itself is in test_gpu_learn.py."""
import os

import numpy as np

import oracle_lib as O
from gpu_util import random_positions


def _targets(n, count, seed):
    oracle = O.load()
    rng = np.random.default_rng(seed)
    out = []
    states = O.states_array(random_positions(oracle, O, n, 4, count, seed, max_ply=20))
    for i in range(count):
        s = O.TzState.from_buffer_copy(states[i].tobytes())
        mv = np.array(O.possible_moves(oracle, s), np.uint16)
        p = rng.random(len(mv)).astype(np.float32)
        st = states[i].copy()
        st["reversible_plies"] = 0  # not representable in a target line (TPS)
        out.append((st, mv, p / p.sum(), float(np.float32(rng.uniform(-1, 1))), float(np.float32(rng.uniform(0.1, 4)))))
    return out


def test_target_buffer_reads_incrementally_and_counts_forced_uses(tmp_path):
    from takzero_amd import formats as F
    from takzero_amd import learn as L

    n = 4
    targets = _targets(n, 12, 1)
    path = tmp_path / "targets-selfplay.txt"
    lines = [F.format_target(n, *t) for t in targets]
    path.write_text("".join(lines[:5]) + "garbage line\n" + lines[5][:20])  # an unparsable line and a half-written one
    buf = L.TargetBuffer(n, 4, forced_uses=2)
    assert buf.fill(str(path), 7) == 5 and len(buf) == 5
    assert buf.fill(str(path), 8) == 0
    with open(path, "a") as f:
        f.write(lines[5][20:] + "".join(lines[6:]))
    assert buf.fill(str(path), 9) == 7 and len(buf) == 12
    assert sorted(it[2] for it in buf.items) == [7] * 5 + [9] * 7
    rng = np.random.default_rng(0)
    batch = buf.take(rng, 8)
    assert len(batch) == 8 and len(buf) == 4
    buf.give_back(batch)           # second of two uses left
    assert len(buf) == 12 and sorted(it[1] for it in buf.items) == [1] * 8 + [2] * 4
    batch = buf.take(rng, 12)
    buf.give_back(batch)           # the eight used twice are gone
    assert len(buf) == 4 and all(it[1] == 1 for it in buf.items)


def test_create_batch_mixes_the_two_buffers_and_builds_dense_tensors():
    import takzero_amd.api as A
    from takzero_amd import learn as L

    n, B = 4, 8
    ex, re = L.TargetBuffer(n, 4, 4), L.TargetBuffer(n, 4, 4)
    ex.items = [[t, 4, 0] for t in _targets(n, 10, 2)]
    re.items = [[t, 4, 0] for t in _targets(n, 10, 3)]
    rng = np.random.default_rng(1)
    states, policy, mask, value, ube = L.create_batch(True, ex, re, rng, n, batch=B, augment=True)
    assert len(ex) == 10 and len(re) == 10 and sum(it[1] == 3 for it in ex.items) == 4 == sum(it[1] == 3 for it in re.items)
    assert states.shape == (B,) and policy.shape == (B, A.policy_size(n)) == mask.shape
    oracle = O.load()
    for i in range(B):
        legal = O.possible_moves(oracle, O.TzState.from_buffer_copy(states[i].tobytes()))
        assert sorted(np.nonzero(mask[i] == 0)[0]) == sorted(legal)  # the mask is exactly the non-legal outputs
        assert abs(float(policy[i].sum()) - 1.0) < 1e-5 and float(policy[i][mask[i] == 1].sum()) == 0.0
    states2, *_ = L.create_batch(False, ex, re, rng, n, batch=B, augment=False)
    assert len(re) == 10 and len(ex) == 10


def test_model_path_with_most_steps(tmp_path):
    from takzero_amd import learn as L

    assert L.model_path_with_most_steps(str(tmp_path)) is None
    for name in ("model_latest.ot", "model_0000000.ot", "model_0001000.ot", "model_0000999.ot", "model_0002000.txt"):
        (tmp_path / name).write_bytes(b"")
    steps, path = L.model_path_with_most_steps(str(tmp_path))
    assert steps == 1000 and os.path.basename(path) == "model_0001000.ot"
