"""The `learn` step on the GPU (learn/src/main.rs:322-423): tensors from a batch of targets
(create_input_and_target_tensors) and compute_loss_and_take_step, over the tz_trainer_* ABI."""
import ctypes as C

import numpy as np

from . import _lib, api
from ._lib import check

BATCH_SIZE = 128        # learn/src/main.rs:43
LEARNING_RATE = 1e-4    # learn/src/main.rs:46
PARAM, GRAD, ADAM_M, ADAM_V = 0, 1, 2, 3


def target_tensors(targets, n):
    """create_input_and_target_tensors (learn/src/main.rs:330-374) without the random symmetry: states, dense policy
    target (policy_tensor), mask of non-legal outputs (move_mask), value and UBE targets (raw variances; the log and
    clamp are applied inside the step).  `targets` = (state, moves, policy, value, ube) tuples."""
    B, out = len(targets), api.policy_size(n)
    states = np.zeros(B, api.STATE_DTYPE)
    policy = np.zeros((B, out), np.float32)
    mask = np.ones((B, out), np.uint8)
    value = np.zeros(B, np.float32)
    ube = np.zeros(B, np.float32)
    for i, (st, moves, pol, v, u) in enumerate(targets):
        states[i] = st
        idx = np.asarray(moves, np.int64)
        policy[i, idx] = pol
        mask[i, idx] = 0
        value[i], ube[i] = v, u
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
