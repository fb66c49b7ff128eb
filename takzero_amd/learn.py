"""The `learn` step on the GPU (learn/src/main.rs:322-423): tensors from a batch of targets
(create_input_and_target_tensors) and compute_loss_and_take_step, over the tz_trainer_* ABI."""
import ctypes as C

import numpy as np

from . import _lib, api
from ._lib import check

BATCH_SIZE = 128        # learn/src/main.rs:43
LEARNING_RATE = 1e-4    # learn/src/main.rs:46
PARAM, GRAD, ADAM_M, ADAM_V = 0, 1, 2, 3


def target_tensors(targets, n, rng=None):
    """create_input_and_target_tensors (learn/src/main.rs:330-374): states, dense policy target (policy_tensor), mask of
    non-legal outputs (move_mask), value and UBE targets (raw variances; the log and clamp are applied inside the
    step).  `targets` = (state, moves, policy, value, ube) tuples; with `rng` every target is first put through a
    random board symmetry (target.augment(rng), target.rs:41-53).  Vectorised over the batch."""
    B, out = len(targets), api.policy_size(n)
    states = np.zeros(B, api.STATE_DTYPE)
    for i, t in enumerate(targets):
        states[i] = t[0]
    counts = np.fromiter((len(t[1]) for t in targets), np.int64, B)
    moves = np.concatenate([np.asarray(t[1], np.int64) for t in targets]) if B else np.zeros(0, np.int64)
    probs = np.concatenate([np.asarray(t[2], np.float32) for t in targets]) if B else np.zeros(0, np.float32)
    rows = np.repeat(np.arange(B), counts)
    if rng is not None:
        from . import augment as AU

        states, moves = AU.augment_batch(states, moves, rows, rng, n)
    policy = np.zeros((B, out), np.float32)
    mask = np.ones((B, out), np.uint8)
    policy[rows, moves] = probs
    mask[rows, moves] = 0
    value = np.fromiter((t[3] for t in targets), np.float32, B)
    ube = np.fromiter((t[4] for t in targets), np.float32, B)
    return states, policy, mask, value, ube


class Trainer:
    """Net + Adam optimizer of learn::main (learn/src/main.rs:100-110) on one GPU."""

    def __init__(self, arch=api.ARCH_NET5, n=0, blocks=0, batch=BATCH_SIZE, lr=LEARNING_RATE, device=0):
        from . import weights as W

        self.lib = _lib.load()
        self.arch, self.n = arch, W.arch_board(arch, n)
        self.blocks, self.batch = W.arch_blocks(arch, blocks), batch
        self.h = C.c_void_p()
        check(self.lib.tz_trainer_create(self.n, arch, device, self.blocks, batch, lr, C.byref(self.h)))
        self.names = {}
        buf = C.create_string_buffer(256)
        cnt = C.c_uint64()
        for i in range(self.lib.tz_trainer_tensor_count(self.h)):
            check(self.lib.tz_trainer_tensor_info(self.h, i, buf, 256, C.byref(cnt)))
            self.names[buf.value.decode()] = int(cnt.value)
        self.extra = {}     # tensors the step never touches (RND nets, SimHash matrix): carried through unchanged
        self.shapes = {}

    def close(self):
        if self.h:
            self.lib.tz_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_tensors(self, tensors):
        """VarStore contents by name (takzero_amd.weights / takzero_amd.ot)."""
        missing = [k for k in self.names if k not in tensors]
        if missing:
            raise ValueError("missing tensors: %s" % missing[:4])
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, np.float32)
            if name in self.names:
                self.shapes[name] = a.shape
                check(self.lib.tz_trainer_set_tensor(self.h, name.encode(), PARAM, a.ctypes.data, a.size))
            else:
                self.extra[name] = a.copy()
        return self

    def load(self, path):
        """Network::load on the trainer's VarStore (learn/src/main.rs:107-120): a LibTorch archive or a .tzw container."""
        check(self.lib.tz_trainer_load(self.h, str(path).encode()))
        return self

    def save(self, path):
        """Network::save (learn/src/main.rs:247-266): a LibTorch archive under tch's variable names (or .tzw), written natively."""
        check(self.lib.tz_trainer_save(self.h, str(path).encode()))

    def from_net(self, net):
        check(self.lib.tz_trainer_from_net(self.h, net.h))
        return self

    def to_net(self, net):
        check(self.lib.tz_trainer_to_net(self.h, net.h))

    def tensor(self, name, what=PARAM):
        out = np.zeros(self.names[name], np.float32)
        check(self.lib.tz_trainer_get_tensor(self.h, name.encode(), what, out.ctypes.data, out.size))
        return out.reshape(self.shapes.get(name, out.shape))

    def tensors(self):
        """Current weights by name, ready for Net.load_tensors / weights.save_tzw."""
        out = {name: self.tensor(name) for name in self.names}
        out.update(self.extra)
        return out

    def step(self, states, policy, mask, value, ube, train_ube=True, apply=True):
        """compute_loss_and_take_step -> (loss_policy, loss_value, loss_ube)."""
        st = api._states(states)
        B = self.batch
        out = api.policy_size(self.n)
        policy = np.ascontiguousarray(policy, np.float32)
        mask = np.ascontiguousarray(mask, np.uint8)
        value = np.ascontiguousarray(value, np.float32)
        ube = np.ascontiguousarray(ube, np.float32)
        if len(st) != B or policy.shape != (B, out) or mask.shape != (B, out) or value.shape != (B,) or ube.shape != (B,):
            raise ValueError("step: batch tensors do not match the trainer's batch size %d" % B)
        losses = np.zeros(3, np.float32)
        check(self.lib.tz_trainer_step(self.h, st.ctypes.data, policy.ctypes.data, mask.ctypes.data, value.ctypes.data,
                                       ube.ctypes.data, 1 if train_ube else 0, 1 if apply else 0, losses.ctypes.data))
        return tuple(float(x) for x in losses)

    def outputs(self):
        B, out = self.batch, api.policy_size(self.n)
        pol, val, ube = np.zeros((B, out), np.float32), np.zeros(B, np.float32), np.zeros(B, np.float32)
        check(self.lib.tz_trainer_outputs(self.h, pol.ctypes.data, val.ctypes.data, ube.ctypes.data))
        return pol, val, ube

    def activation(self, layer):
        """Output of trunk layer `layer` in the last step's forward, [batch, n*n, 256] (tz_trainer_activation)."""
        out = np.zeros((self.batch, self.n * self.n, 256), np.float32)
        check(self.lib.tz_trainer_activation(self.h, layer, out.ctypes.data, out.size))
        return out


# ---------------------------------------------------------------------------------------------
# learn::main (learn/src/main.rs:99-289): replay buffers with forced uses, back-pressure file, model files
STEPS_PER_SAVE = 100               # :44
STEPS_PER_CHECKPOINT = 50_000      # :45
INITIAL_RANDOM_TARGETS = BATCH_SIZE * 2_000  # :49
PRE_TRAINING_STEPS = 1_000         # :50
STEPS_BEFORE_REANALYZE = 5000      # :54
MIN_SELFPLAY_BUFFER_LEN = 10_000   # :55
MIN_REANALYZE_BUFFER_LEN = 2_000   # :57
SELFPLAY_TARGET_FORCED_USES = 4    # :59
REANALYZE_TARGET_FORCED_USES = 4   # :60


class TargetBuffer:
    """Vec<TargetWithContext> fed from an append-only targets file (fill_buffer_with_targets, :291-319)."""

    def __init__(self, n, half_komi, forced_uses):
        self.n, self.half_komi, self.forced_uses = n, half_komi, forced_uses
        self.items = []   # [target, remaining uses, model steps when read]
        self.seek = 0

    def __len__(self):
        return len(self.items)

    def fill(self, path, model_steps):
        from . import formats

        with open(path, "rb") as f:
            f.seek(self.seek)
            data = f.read()
        targets, consumed, _skipped = formats.parse_targets(data, self.n, self.half_komi)  # bad lines are skipped (:308)
        self.seek += consumed
        self.items.extend([t, self.forced_uses, model_steps] for t in targets)
        return len(targets)

    def take(self, rng, count):
        """`count` targets drawn uniformly without replacement and removed — what shuffling the whole buffer and
        draining its tail does (create_batch, :493-499), in O(count) instead of O(len)."""
        pick = sorted((int(i) for i in rng.choice(len(self.items), size=count, replace=False)), reverse=True)
        batch = []
        for i in pick:              # swap-remove, largest index first so earlier picks stay valid
            batch.append(self.items[i])
            self.items[i] = self.items[-1]
            self.items.pop()
        order = rng.permutation(count)
        return [batch[i] for i in order]

    def give_back(self, batch):
        for item in batch:  # TargetWithContext::reuse
            if item[1] > 1:
                item[1] -= 1
                self.items.append(item)


def model_path_with_most_steps(directory):
    """get_model_path_with_most_steps (:272-289): model_<steps>.ot with the largest number."""
    import os

    best = None
    for name in os.listdir(directory):
        stem, ext = os.path.splitext(name)
        if ext != ".ot" or "_" not in stem:
            continue
        tail = stem.split("_", 1)[1]
        if tail.isdigit() and (best is None or int(tail) > best[0]):
            best = (int(tail), os.path.join(directory, name))
    return best


def create_batch(using_reanalyze, exploitation, reanalyze, rng, n, batch=BATCH_SIZE, augment=True):
    """create_batch + create_input_and_target_tensors (:486-516, :330-374)."""
    if using_reanalyze:
        a, b = exploitation.take(rng, batch // 2), reanalyze.take(rng, batch // 2)
        items = a + b
    else:
        a, b = exploitation.take(rng, batch), []
        items = a
    tensors = target_tensors([it[0] for it in items], n, rng if augment else None)
    exploitation.give_back(a)
    reanalyze.give_back(b)
    return tensors


def pre_training(trainer, mcts, rng_seed, directory=None, initial_targets=INITIAL_RANDOM_TARGETS,
                 steps=PRE_TRAINING_STEPS, log=None):
    """pre_training (:425-484): uniformly random games from the openings, uniform policy targets, discounted game
    result as value, UBE target 4 - eps; the UBE head is not trained.  `mcts`: a BatchedMCTS with the Dummy agent."""
    import os

    from . import formats
    from .selfplay import SelfPlay

    sp = SelfPlay(mcts, 0, seed=rng_seed, search="random")
    rng = np.random.default_rng([rng_seed, 11])
    buffer = []
    while len(buffer) < initial_targets:
        targets, _ = sp.play_move()
        buffer.extend(targets)
    order = rng.permutation(len(buffer))
    buffer = [buffer[i] for i in order]
    if directory is not None:
        with open(os.path.join(directory, "targets-initial.txt"), "w") as f:
            f.write(formats.format_targets(mcts.n, buffer))
    B, losses = trainer.batch, []
    for s in range(min(steps, len(buffer) // B)):
        losses.append(trainer.step(*target_tensors(buffer[s * B:(s + 1) * B], mcts.n, rng), train_ube=False))
        if log and s % 100 == 0:
            log("pre-training step %d: %r" % (s, losses[-1]))
    return losses


def save_model(trainer, path, hash_net=None, background=None):
    """Network::save (network/mod.rs:16-18; net6_simhash.rs:152-171 also writes bitvec.bin beside the model).  The
    weights are read back from the GPU here; with `background` (a runner.AsyncAppender) the archive itself is written
    by that thread while training goes on."""
    import os

    from . import ot

    snapshot = trainer.tensors()
    if hash_net is not None:
        hash_net.save_bitset(os.path.join(os.path.dirname(str(path)), "bitvec.bin"))
    if background is None:
        ot.save_ot(path, snapshot)
    else:
        background.submit(lambda: ot.save_ot(path, snapshot))


def run_learn(directory, trainer, half_komi=4, steps=None, seed=0, pre_train_mcts=None, hash_net=None,
              min_selfplay=MIN_SELFPLAY_BUFFER_LEN, min_reanalyze=MIN_REANALYZE_BUFFER_LEN,
              steps_before_reanalyze=STEPS_BEFORE_REANALYZE, steps_per_save=STEPS_PER_SAVE,
              steps_per_checkpoint=STEPS_PER_CHECKPOINT, pre_training_steps=PRE_TRAINING_STEPS,
              initial_targets=INITIAL_RANDOM_TARGETS, restart_targets=None, read_interval=10.0, sleep=30.0,
              max_wait=None, log=None):
    """learn::main (:99-270).  `trainer` must already hold initial weights (Net::new) unless the directory has a
    model_<steps>.ot to resume from.  Returns the number of training steps the model has seen."""
    import os
    import time

    from . import formats, ot

    from .runner import AsyncAppender

    n, rng = trainer.n, np.random.default_rng([seed, 5])
    saver = AsyncAppender()   # model files are written behind the training loop, in order
    resume = model_path_with_most_steps(directory)
    if resume is not None:
        starting_steps = resume[0]
        trainer.load_tensors(ot.load_ot(resume[1]))
    else:
        starting_steps = 0
        save_model(trainer, os.path.join(directory, "model_0000000.ot"), hash_net)
    if restart_targets is not None:
        # --restart-targets (:126-147): one pass over a saved target file, UBE head not trained
        with open(restart_targets) as f:
            saved = []
            for line in f:
                try:
                    saved.append(formats.parse_target(line, n, half_komi))
                except Exception:
                    continue
        rng.shuffle(saved)
        B = trainer.batch
        for s in range(len(saved) // B):
            trainer.step(*target_tensors(saved[s * B:(s + 1) * B], n, rng), train_ube=False)
            starting_steps += 1
        save_model(trainer, os.path.join(directory, "model_%07d.ot" % starting_steps), hash_net)
    elif resume is None:
        if pre_train_mcts is not None and pre_training_steps > 0:
            pre_training(trainer, pre_train_mcts, seed, directory, initial_targets, pre_training_steps, log)
            starting_steps += pre_training_steps
            save_model(trainer, os.path.join(directory, "model_%07d.ot" % starting_steps), hash_net)
    save_model(trainer, os.path.join(directory, "model_latest.ot"), hash_net)
    exploitation = TargetBuffer(n, half_komi, SELFPLAY_TARGET_FORCED_USES)
    reanalyze = TargetBuffer(n, half_komi, REANALYZE_TARGET_FORCED_USES)
    last_loaded = -1e18
    model_steps = starting_steps
    done = 0
    t0 = time.monotonic()
    # The step of batch k runs on a worker thread (inside a ctypes call, interpreter lock released) while this thread
    # samples, augments and densifies batch k+1; results are collected in step order, so logs and model files are those
    # of the sequential loop.
    from concurrent.futures import ThreadPoolExecutor

    def finish(step_no, fut, states):
        losses = fut.result()
        if hash_net is not None:
            hash_net.hash_indices(states, update=True)  # net.update_counts(&tensors.input), :418
        if log:
            log("step %d: loss_policy %.5f loss_value %.5f loss_ube %.5f" % ((step_no,) + losses))
        if step_no % steps_per_save == 0:
            save_model(trainer, os.path.join(directory, "model_latest.ot"), hash_net, saver)
        if step_no % steps_per_checkpoint == 0:
            save_model(trainer, os.path.join(directory, "model_%07d.ot" % step_no), hash_net, saver)

    pending = None
    with ThreadPoolExecutor(1) as pool:
        try:
            while steps is None or done < steps:
                model_steps += 1
                using_reanalyze = restart_targets is not None or model_steps >= steps_before_reanalyze
                while True:
                    if time.monotonic() - last_loaded >= read_interval:
                        for buf, name, use in ((exploitation, "targets-selfplay.txt", True),
                                               (reanalyze, "targets-reanalyze.txt", using_reanalyze)):
                            if use:
                                try:
                                    buf.fill(os.path.join(directory, name), model_steps)
                                except OSError as err:
                                    if log:
                                        log("Cannot read %s: %s" % (name, err))
                        last_loaded = time.monotonic()
                        with open(os.path.join(directory, "buffer_lengths.txt"), "w") as f:
                            f.write(formats.format_buffer_lengths(len(exploitation), len(reanalyze)))
                    if len(exploitation) >= min_selfplay and (not using_reanalyze or len(reanalyze) >= min_reanalyze):
                        break
                    if max_wait is not None and time.monotonic() - t0 > max_wait:
                        raise TimeoutError("not enough targets (%d selfplay, %d reanalyze)" % (len(exploitation), len(reanalyze)))
                    time.sleep(sleep)
                tensors = create_batch(using_reanalyze, exploitation, reanalyze, rng, n, trainer.batch)
                if pending is not None:
                    finish(*pending)
                pending = (model_steps, pool.submit(trainer.step, *tensors, train_ube=True), tensors[0])
                time.sleep(0.0005)   # hand the interpreter lock over so the worker enters its (lock-free) native call now
                done += 1
        finally:
            try:
                if pending is not None:
                    finish(*pending)
            finally:
                saver.close()
    return model_steps


# ---------------------------------------------------------------------------------------------
class NativeLearnLoop:
    """learn::main's buffers, batch construction and training loop in native code (csrc/tz_host_learn.cpp, tz_learn_*)."""

    def __init__(self, trainer, half_komi=4, seed=0, forced_uses=(SELFPLAY_TARGET_FORCED_USES, REANALYZE_TARGET_FORCED_USES)):
        self.trainer, self.lib = trainer, _lib.load()
        self.h = C.c_void_p()
        check(self.lib.tz_learn_create(trainer.h, half_komi, seed, forced_uses[0], forced_uses[1], C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_learn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, which, path, model_steps=0):
        added = C.c_uint64()
        check(self.lib.tz_learn_feed(self.h, which, str(path).encode(), model_steps, C.byref(added)))
        return added.value

    def add_lines(self, which, text, model_steps=0):
        data = text if isinstance(text, bytes) else text.encode()
        added = C.c_uint64()
        check(self.lib.tz_learn_add_lines(self.h, which, data, len(data), model_steps, C.byref(added)))
        return added.value

    def buffer_len(self, which):
        n = C.c_uint64()
        check(self.lib.tz_learn_buffer_len(self.h, which, C.byref(n)))
        return n.value

    def step(self, using_reanalyze=False, train_ube=True, augment=True):
        losses = np.zeros(3, np.float32)
        check(self.lib.tz_learn_step(self.h, 1 if using_reanalyze else 0, 1 if train_ube else 0, 1 if augment else 0, losses.ctypes.data))
        return tuple(float(x) for x in losses)

    def last_batch(self):
        B, out = self.trainer.batch, api.policy_size(self.trainer.n)
        states = np.zeros(B, api.STATE_DTYPE)
        policy, mask = np.zeros((B, out), np.float32), np.zeros((B, out), np.uint8)
        value, ube = np.zeros(B, np.float32), np.zeros(B, np.float32)
        check(self.lib.tz_learn_last_batch(self.h, states.ctypes.data, policy.ctypes.data, mask.ctypes.data, value.ctypes.data,
                                           ube.ctypes.data))
        return states, policy, mask, value, ube

    def run(self, directory, starting_steps, steps, min_selfplay=MIN_SELFPLAY_BUFFER_LEN, min_reanalyze=MIN_REANALYZE_BUFFER_LEN,
            steps_before_reanalyze=STEPS_BEFORE_REANALYZE, read_interval=10.0, sleep=30.0, max_wait=None, on_step=None):
        """tz_learn_run; on_step(model_steps, losses, states) is called after every step."""
        cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p, C.c_int)
        failure = []

        def trampoline(_user, step_no, losses, states, batch):
            try:
                if on_step is not None:
                    st = np.frombuffer((C.c_char * (batch * api.STATE_DTYPE.itemsize)).from_address(states), dtype=api.STATE_DTYPE)
                    on_step(int(step_no), (losses[0], losses[1], losses[2]), st)
                return 0
            except BaseException as e:   # surfaces after the native loop returns
                failure.append(e)
                return 1

        cb = cb_type(trampoline)
        out = C.c_int64()
        rc = self.lib.tz_learn_run(self.h, str(directory).encode(), starting_steps, -1 if steps is None else steps, min_selfplay,
                                   min_reanalyze, steps_before_reanalyze, read_interval, sleep, -1.0 if max_wait is None else max_wait,
                                   C.cast(cb, C.c_void_p), None, C.byref(out))
        if failure:
            raise failure[0]
        if rc == -6 and "not enough targets" in self.lib.tz_last_error().decode(errors="replace"):
            raise TimeoutError(self.lib.tz_last_error().decode(errors="replace"))
        check(rc)
        return out.value


def run_learn_native(directory, trainer, half_komi=4, steps=None, seed=0, pre_train_mcts=None, hash_net=None,
                     min_selfplay=MIN_SELFPLAY_BUFFER_LEN, min_reanalyze=MIN_REANALYZE_BUFFER_LEN,
                     steps_before_reanalyze=STEPS_BEFORE_REANALYZE, steps_per_save=STEPS_PER_SAVE,
                     steps_per_checkpoint=STEPS_PER_CHECKPOINT, pre_training_steps=PRE_TRAINING_STEPS,
                     initial_targets=INITIAL_RANDOM_TARGETS, read_interval=10.0, sleep=30.0, max_wait=None, log=None):
    """learn::main (learn/src/main.rs:99-270) with the buffers, batch construction and loop in native code; this function
    keeps what touches files in the reference's formats through host tools: model discovery / resume and the save
    points (LibTorch archives)."""
    import os

    from . import ot
    from .runner import AsyncAppender
    from .selfplay import NativeSelfPlay

    saver = AsyncAppender()
    try:
        resume = model_path_with_most_steps(directory)
        if resume is not None:
            starting_steps = resume[0]
            trainer.load_tensors(ot.load_ot(resume[1]))
        else:
            starting_steps = 0
            save_model(trainer, os.path.join(directory, "model_0000000.ot"), hash_net)
            if pre_train_mcts is not None and pre_training_steps > 0:   # pre_training (:425-484)
                sp = NativeSelfPlay(pre_train_mcts, 0, seed=seed, search="random")
                lines = []
                while len(lines) < initial_targets:
                    sp.play_move()
                    lines.extend(sp.take_text(0).splitlines(keepends=True))
                np.random.default_rng([seed, 11]).shuffle(lines)
                with open(os.path.join(directory, "targets-initial.txt"), "wb") as f:
                    f.write(b"".join(lines))
                pre = NativeLearnLoop(trainer, half_komi, seed + 1, forced_uses=(1, 1))   # every target used once
                pre.add_lines(0, b"".join(lines))
                for s in range(min(pre_training_steps, len(lines) // trainer.batch)):
                    losses = pre.step(using_reanalyze=False, train_ube=False, augment=True)
                    if log and s % 100 == 0:
                        log("pre-training step %d: %r" % (s, losses))
                pre.close()
                starting_steps += pre_training_steps
                save_model(trainer, os.path.join(directory, "model_%07d.ot" % starting_steps), hash_net)
        save_model(trainer, os.path.join(directory, "model_latest.ot"), hash_net)
        loop = NativeLearnLoop(trainer, half_komi, seed)

        def on_step(step_no, losses, states):
            if hash_net is not None:
                hash_net.hash_indices(states, update=True)
            if log:
                log("step %d: loss_policy %.5f loss_value %.5f loss_ube %.5f" % ((step_no,) + tuple(losses)))
            if step_no % steps_per_save == 0:
                save_model(trainer, os.path.join(directory, "model_latest.ot"), hash_net, saver)
            if step_no % steps_per_checkpoint == 0:
                save_model(trainer, os.path.join(directory, "model_%07d.ot" % step_no), hash_net, saver)

        return loop.run(directory, starting_steps, steps, min_selfplay, min_reanalyze, steps_before_reanalyze, read_interval, sleep,
                        max_wait, on_step)
    finally:
        saver.close()
