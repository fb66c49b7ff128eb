"""The learn step against PyTorch autograd on the CPU (oracle/learn_torch.py): outputs in
training mode, the three losses, every gradient, BatchNorm running statistics and the weights after Adam steps.
fp32 on both sides; tolerances are stated next to each check."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def _batch(oracle, n, B, seed):
    rng = np.random.default_rng(seed)
    states = random_positions(oracle, O, n, 4, B, seed, max_ply=24)
    planes = np.stack([O.game_repr(oracle, s) for s in states]).reshape(B, -1, n, n)
    out = (3 + 4 * (2 ** n - 2)) * n * n
    policy = np.zeros((B, out), np.float32)
    mask = np.ones((B, out), np.uint8)
    for i, s in enumerate(states):
        mv = np.array(O.possible_moves(oracle, s), np.int64)
        p = rng.random(len(mv)).astype(np.float32)
        policy[i, mv] = p / p.sum() * (0.97 if i % 3 == 0 else 1.0)  # visit-count targets need not sum to one
        mask[i, mv] = 0
    value = rng.uniform(-1, 1, B).astype(np.float32)
    ube = rng.uniform(1e-6, 5.0, B).astype(np.float32)  # some above MAXIMUM_VARIANCE, some tiny: both clamps
    ube[0] = 1e-9
    return O.states_array(states), planes, policy, mask, value, ube


@pytest.mark.parametrize("n,blocks", [(4, 2), (5, 1)])
def test_step_matches_torch_autograd(n, blocks):
    import torch

    import learn_torch as LT
    A = require_gpu()
    from takzero_amd import learn as L
    from takzero_amd import weights as W

    oracle = O.load()
    B, lr = 64, 1e-4
    w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=3 + n, trained_stats=True)
    tr = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B, lr=lr).load_tensors(w)
    p = LT.make_params(w)
    opt = LT.adam(p, lr)
    torch.manual_seed(0)
    for step, train_ube in enumerate((True, False, True)):
        states, planes, policy, mask, value, ube = _batch(oracle, n, B, 100 + step)
        got = tr.step(states, policy, mask, value, ube, train_ube=train_ube, apply=True)
        grads = {k: tr.tensor(k, L.GRAD) for k in tr.names if "running_" not in k}
        outs = tr.outputs()
        opt.zero_grad(set_to_none=True)
        # A ReLU input within rounding of zero has no derivative to compare: its mask is whatever side the last fp32 rounding fell on,
        # and one flipped pixel moves a conv's weight gradient by a percent (seen: layer 2, one input at 1.8e-7 against 0.0 under
        # another summation order of the same GEMM).  At those entries the reference takes the side the trainer took.
        relu_masks = [torch.from_numpy(tr.activation(l).transpose(0, 2, 1).reshape(B, 256, n, n) > 0) for l in range(1 + 2 * blocks)]
        want, wouts = LT.losses(p, torch.from_numpy(planes), torch.from_numpy(mask.astype(bool)), torch.from_numpy(policy),
                                torch.from_numpy(value), torch.from_numpy(ube), blocks, train_ube, relu_masks=relu_masks)
        (want[0] + want[1] + want[2]).backward()
        # identical weights on both sides (see the re-synchronisation below), only the fp32 summation order differs
        # (measured: 2e-6 on the outputs, 2e-6 relative on the gradients)
        otol, ltol, gtol = 2e-5, 1e-5, 1e-4
        for g, t_ in zip(outs, wouts):
            assert np.allclose(g, t_.detach().numpy(), atol=otol, rtol=0)
        for g, t_ in zip(got, want):
            assert abs(g - float(t_.detach())) <= ltol * (1 + abs(float(t_.detach()))), (step, got, [float(x.detach()) for x in want])
        # gradients: error relative to the tensor's largest gradient
        for k, g in grads.items():
            tg = p[k].grad
            if tg is None:
                assert k.startswith("ube.") and not train_ube
                continue
            tg = tg.numpy().reshape(g.shape)
            scale = float(np.abs(tg).max()) + 1e-12
            assert float(np.abs(g - tg).max()) <= gtol * scale + 1e-7, (step, k, float(np.abs(g - tg).max()), scale)
        opt.step()
        # weights after the step: Adam moves every weight by about lr, in the direction of its gradient's sign, so
        # agreement to a small fraction of lr means the same update was applied; gradients near zero may differ more
        for k in tr.names:
            a, b = tr.tensor(k), p[k].detach().numpy().reshape(tr.tensor(k).shape)
            diff = np.abs(a - b)
            if "running_" in k:
                assert float(diff.max()) <= 1e-5 + 1e-4 * float(np.abs(b).max()), (step, k)
            else:
                assert float(np.quantile(diff, 0.99)) <= 0.05 * lr, (step, k, float(np.quantile(diff, 0.99)))
                assert float(diff.max()) <= 2.5 * lr, (step, k, float(diff.max()))
            # a weight whose gradient is ~0 is moved by +-lr on either side of zero; copying the trainer's weights over
            # keeps that from compounding, while torch's Adam state (step counts, both moments) stays its own
            p[k].data.copy_(torch.from_numpy(a.reshape(p[k].shape)))
    # the UBE head was stepped twice, everything else three times (torch.optim.Adam keeps a step count per tensor)
    assert float(np.abs(tr.tensor("ube.linear.weight") - w["ube.linear.weight"]).max()) <= 2 * lr * 1.01


def test_trained_weights_feed_the_inference_net():
    """learn -> model file -> selfplay: the trainer's tensors load into the inference Net, and with BatchNorm running
    statistics in place the fp32 inference path reproduces an eval-mode forward of the same weights."""
    A = require_gpu()
    from takzero_amd import learn as L
    from takzero_amd import weights as W

    oracle = O.load()
    n, blocks, B = 4, 1, 64
    w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=9, trained_stats=True)
    tr = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B).load_tensors(w)
    states, planes, policy, mask, value, ube = _batch(oracle, n, B, 7)
    first = tr.step(states, policy, mask, value, ube)
    for _ in range(20):
        last = tr.step(states, policy, mask, value, ube)
    assert sum(last[:2]) < sum(first[:2]), (first, last)  # the same batch 20 times: the loss goes down
    trained = tr.tensors()
    assert set(trained) == set(w)
    net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_F32, blocks=blocks).load_tensors(trained)
    pol, val, _ = net.forward_raw(states)
    assert np.isfinite(pol).all() and np.isfinite(val).all()


def test_trainer_rejects_bad_arguments():
    A = require_gpu()
    from takzero_amd import learn as L

    with pytest.raises(A.TakzeroError):
        L.Trainer(arch=A.ARCH_TEST, n=4, blocks=1, batch=100)  # not a multiple of 64
    tr = L.Trainer(arch=A.ARCH_TEST, n=4, blocks=1, batch=64)
    with pytest.raises(ValueError):
        tr.load_tensors({})
    with pytest.raises(ValueError):
        tr.step(np.zeros(3, A.STATE_DTYPE), np.zeros((3, 4)), np.zeros((3, 4)), np.zeros(3), np.zeros(3))


def test_closed_loop_learn_writes_models_that_selfplay_reloads(tmp_path):
    """learn::main at toy scale next to selfplay on one directory: pre-training on random games, model_*.ot written
    in the reference's format, selfplay targets consumed with forced uses, buffer_lengths.txt written, and the
    self-play side picking up model_latest.ot."""
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import learn as L
    from takzero_amd import ot
    from takzero_amd import runner as R
    from takzero_amd import weights as W

    try:
        ot.build_writer()
    except RuntimeError as e:
        pytest.skip(str(e))
    d, n, blocks, B = str(tmp_path), 4, 1, 64
    w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=21)
    trainer = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B).load_tensors(w)
    dummy = A.BatchedMCTS(96, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    # self-play side first: some targets with the initial weights
    net = A.Net(arch=A.ARCH_TEST, n=n, blocks=blocks).load_tensors(w)
    mcts = A.BatchedMCTS(96, n, 4, agent=net, node_capacity=1 << 13)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    R.run_selfplay(d, mcts, 16, moves=40, seed=2, search="gumbel", sampled_actions=4, watch_model=False, max_wait=5)
    produced = sum(1 for _ in open(os.path.join(d, "targets-selfplay.txt")))
    assert produced >= 2 * B
    logs = []
    steps = L.run_learn(d, trainer, steps=7, seed=1, pre_train_mcts=dummy, min_selfplay=B, steps_before_reanalyze=10 ** 9,
                        steps_per_save=2, steps_per_checkpoint=4, pre_training_steps=5, initial_targets=5 * B,
                        read_interval=0.0, sleep=0.01, max_wait=20, log=logs.append)
    assert steps == 5 + 7  # saved at every second step, so model_latest.ot is the final state
    names = sorted(os.listdir(d))
    for want in ("model_0000000.ot", "model_0000005.ot", "model_0000008.ot", "model_latest.ot", "targets-initial.txt",
                 "buffer_lengths.txt"):
        assert want in names, names
    sp_len, re_len = R.read_buffer_lengths(d)
    assert re_len == 0 and 0 < sp_len <= produced
    initial = open(os.path.join(d, "targets-initial.txt")).read().splitlines(keepends=True)
    assert len(initial) >= 5 * B
    st, mv, pol, value, ube = F.parse_target(initial[0], n, 4)
    assert np.allclose(pol, 1.0 / len(mv)) and abs(ube - 4.0) < 1e-5 and abs(value) <= 1.0
    # the weights moved, and the file on disk is what the trainer holds
    latest = ot.load_ot(os.path.join(d, "model_latest.ot"))
    held = trainer.tensors()
    assert set(latest) == set(w)
    assert np.array_equal(latest["policy.conv2d.weight"], held["policy.conv2d.weight"])
    assert not np.array_equal(latest["policy.conv2d.weight"], w["policy.conv2d.weight"])
    # resuming finds the checkpoint with the most steps
    assert L.model_path_with_most_steps(d)[0] == 12
    # the self-play side reloads it (ModelWatcher) and keeps playing
    sp = R.run_selfplay(d, mcts, 16, moves=3, seed=3, search="gumbel", sampled_actions=4, watch_model=True, max_wait=5)
    pol_a, _, _ = net.forward_raw(mcts.get_positions()[:8])
    fresh = A.Net(arch=A.ARCH_TEST, n=n, blocks=blocks).load_tensors(latest)
    pol_b, _, _ = fresh.forward_raw(mcts.get_positions()[:8])
    assert np.array_equal(pol_a, pol_b) and sp.moves_played == 3

    # --restart-targets (learn/src/main.rs:126-147): one pass over a saved target file in a fresh directory
    d2 = os.path.join(d, "restart")
    os.mkdir(d2)
    fresh = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B).load_tensors(w)
    steps2 = L.run_learn(d2, fresh, steps=0, seed=4, restart_targets=os.path.join(d, "targets-selfplay.txt"),
                         read_interval=0.0, sleep=0.01, max_wait=5)
    assert steps2 == produced // B and os.path.exists(os.path.join(d2, "model_%07d.ot" % steps2))
    assert not np.array_equal(fresh.tensor("policy.conv2d.weight"), w["policy.conv2d.weight"])
    assert np.array_equal(fresh.tensor("ube.linear.weight"), w["ube.linear.weight"])  # the UBE head is not trained there


def test_native_learn_loop_buffers_sampling_and_augmentation(tmp_path):
    """tz_learn_* (csrc/tz_host_learn.cpp): file tailing, forced uses, sampling without replacement, and the batch
    tensors under a random symmetry — every row must be a symmetric image of one of the fed targets, with its policy
    moved to the symmetric moves, the mask exactly the complement of the legal moves, value and UBE untouched."""
    A = require_gpu()
    from takzero_amd import augment as AU
    from takzero_amd import formats as F
    from takzero_amd import learn as L
    from takzero_amd import weights as W
    from test_learn_host import _targets

    oracle = O.load()
    n, B = 4, 64
    targets = _targets(n, 96, 5)
    lines = [F.format_target(n, *t) for t in targets]
    path = tmp_path / "targets-selfplay.txt"
    path.write_text("".join(lines[:80]) + "junk\n" + lines[80][:17])
    trainer = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=1, batch=B).load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=2))
    loop = L.NativeLearnLoop(trainer, 4, seed=1, forced_uses=(2, 2))
    assert loop.feed(0, path) == 80 and loop.buffer_len(0) == 80 and loop.feed(0, path) == 0
    with open(path, "a") as f:
        f.write(lines[80][17:] + "".join(lines[81:]))
    assert loop.feed(0, path) == 16 and loop.buffer_len(0) == 96
    losses = loop.step(using_reanalyze=False, train_ube=True, augment=True)
    assert all(np.isfinite(losses)) and loop.buffer_len(0) == 96          # 64 used once of two: all back
    states, policy, mask, value, ube = loop.last_batch()
    by_key = {}
    for t in targets:   # every symmetric image of every fed target
        parsed = F.parse_target(F.format_target(n, *t), n, 4)
        for sym in range(8):
            st = AU.augment_state(parsed[0], sym, n)
            by_key.setdefault(A.state_to_tps(st) + "|%r|%r" % (float(parsed[3]), float(parsed[4])), []).append((parsed, sym))
    seen = {}
    for i in range(B):
        key = A.state_to_tps(states[i]) + "|%r|%r" % (float(value[i]), float(ube[i]))
        assert key in by_key, i
        legal = O.possible_moves(oracle, O.TzState.from_buffer_copy(np.array([states[i]]).tobytes()))
        assert sorted(legal) == sorted(np.nonzero(mask[i] == 0)[0].tolist())
        ok = False
        for parsed, sym in by_key[key]:   # a position that is its own mirror image matches under more than one symmetry
            moved = AU.augment_moves(parsed[1], sym, n).astype(np.int64)
            assert sorted(moved.tolist()) == sorted(legal)
            ok = ok or np.array_equal(policy[i, moved], parsed[2])
        assert ok, i
        assert abs(float(policy[i].sum(dtype=np.float64)) - float(by_key[key][0][0][2].sum(dtype=np.float64))) < 1e-6
        p0 = by_key[key][0][0]
        ident = A.state_to_tps(p0[0]) + "|%r|%r" % (float(p0[3]), float(p0[4]))
        seen[ident] = seen.get(ident, 0) + 1
    have = {}
    for t in targets:
        p0 = F.parse_target(F.format_target(n, *t), n, 4)
        ident = A.state_to_tps(p0[0]) + "|%r|%r" % (float(p0[3]), float(p0[4]))
        have[ident] = have.get(ident, 0) + 1
    assert sum(seen.values()) == B and all(c <= have[k] for k, c in seen.items())   # without replacement: no target drawn twice
    loop.step(augment=True)                                                 # 64 uses of 192 left, held by 32..64 targets
    left = loop.buffer_len(0)
    assert 32 <= left <= 64
    if left < B:                                                            # fewer targets than a batch: the caller has to wait
        with pytest.raises(A.TakzeroError):
            loop.step()
    once = L.NativeLearnLoop(trainer, 4, seed=2, forced_uses=(1, 1))
    assert once.feed(0, path) == 96
    once.step()
    assert once.buffer_len(0) == 32                                         # one use each: a drawn target is gone
    with pytest.raises(A.TakzeroError):
        once.step()


def test_native_learn_main_loop_on_a_directory(tmp_path):
    """run_learn_native: resume / pre-training / model files from Python host tools, buffers + batches + loop native."""
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import learn as L
    from takzero_amd import ot
    from takzero_amd import runner as R
    from takzero_amd import weights as W

    try:
        ot.build_writer()
    except RuntimeError as e:
        pytest.skip(str(e))
    d, n, blocks, B = str(tmp_path), 4, 1, 64
    w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=21)
    trainer = L.Trainer(arch=A.ARCH_TEST, n=n, blocks=blocks, batch=B).load_tensors(w)
    dummy = A.BatchedMCTS(96, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    net = A.Net(arch=A.ARCH_TEST, n=n, blocks=blocks).load_tensors(w)
    mcts = A.BatchedMCTS(96, n, 4, agent=net, node_capacity=1 << 13)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    R.run_selfplay(d, mcts, 16, moves=40, seed=2, search="gumbel", sampled_actions=4, watch_model=False, native=True, max_wait=5)
    produced = sum(1 for _ in open(os.path.join(d, "targets-selfplay.txt")))
    logs = []
    steps = L.run_learn_native(d, trainer, steps=7, seed=1, pre_train_mcts=dummy, min_selfplay=B, steps_before_reanalyze=10 ** 9,
                               steps_per_save=2, steps_per_checkpoint=4, pre_training_steps=5, initial_targets=5 * B,
                               read_interval=0.0, sleep=0.01, max_wait=20, log=logs.append)
    assert steps == 12 and sum(1 for m in logs if m.startswith("step ")) == 7
    names = sorted(os.listdir(d))
    for want in ("model_0000000.ot", "model_0000005.ot", "model_0000008.ot", "model_0000012.ot", "model_latest.ot", "targets-initial.txt"):
        assert want in names, names
    sp_len, re_len = R.read_buffer_lengths(d)
    assert re_len == 0 and 0 < sp_len <= produced
    latest = ot.load_ot(os.path.join(d, "model_latest.ot"))
    assert np.array_equal(latest["policy.conv2d.weight"], trainer.tensor("policy.conv2d.weight"))
    assert not np.array_equal(latest["policy.conv2d.weight"], w["policy.conv2d.weight"])
    with pytest.raises(TimeoutError):   # nothing left to learn from: the loop waits for targets, here until max_wait
        L.run_learn_native(d, trainer, steps=10 ** 6, min_selfplay=10 ** 6, read_interval=0.0, sleep=0.01, max_wait=0.2)

    class Stop(Exception):
        pass

    def raising(step_no, losses, states):
        if step_no == steps + 2:
            raise Stop()

    loop = L.NativeLearnLoop(trainer, 4, seed=3)
    with pytest.raises(Stop):   # an exception in the step callback ends the native loop and surfaces here
        loop.run(d, steps, 50, min_selfplay=B, steps_before_reanalyze=10 ** 9, read_interval=0.0, sleep=0.01, max_wait=10, on_step=raising)
    loop.close()


def test_stream_k_gemm_equals_the_tile_gemm_on_every_shape_of_a_step(tmp_path):
    """csrc/tz_learn.hip: gemm_sk_kernel + gemm_fixup_kernel (equal shares of (tile, k-slab) pairs per workgroup, parts added in
    ascending k) against gemm_f32_kernel (one workgroup per tile) on the same operands — plain / transposed A, stored / gathered
    im2col view, with and without bias and accumulation, at the shapes of a batch-128 and a batch-64 step on 5x5, a 6x6 shape,
    and with 512, 96 and 7 workgroups (1 to hundreds of slabs each); the workspace starts as NaN, so a part that is read without
    having been written shows.  tools/learn_gemm_check.hip includes the unit itself (the kernels are in an anonymous namespace)."""
    import subprocess

    require_gpu()
    from takzero_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "learn_gemm_check")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w",
                        "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "learn_gemm_check.hip"),
                        "-L" + os.path.dirname(_lib.LIB_PATH), "-ltakzero_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-ldl", "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the check here: " + r.stderr[-300:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all equal to rounding" in r.stdout and "MISMATCH" not in r.stdout, (r.stdout[-1500:], r.stderr[-500:])
    assert r.stdout.count("max |diff|") == 45
