"""ctypes binding of libtakzero_hip.so (include/takzero_hip.h).  There is no CPU fallback: if the
HIP library is missing the import fails loudly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtakzero_hip.so")
MAX_SQ = 36

STATE_DTYPE = np.dtype([
    ("colors", np.uint64, (MAX_SQ,)), ("height", np.uint8, (MAX_SQ,)), ("top", np.uint8, (MAX_SQ,)),
    ("stones", np.uint8, (2,)), ("caps", np.uint8, (2,)), ("to_move", np.uint8), ("n", np.uint8),
    ("half_komi", np.int8), ("pad0", np.uint8), ("ply", np.uint16), ("reversible_plies", np.uint16),
], align=True)
ROOT_INFO_DTYPE = np.dtype([
    ("visit_count", np.uint32), ("n_children", np.uint32), ("eval_tag", np.uint8),
    ("is_terminal_env", np.uint8), ("ply", np.uint16), ("eval_bits", np.uint32),
    ("std_dev", np.float32), ("logit", np.float32), ("probability", np.float32)], align=True)
assert STATE_DTYPE.itemsize == 376 and ROOT_INFO_DTYPE.itemsize == 28

# every symbol include/takzero_hip.h declares
SYMBOLS = [
    "tz_last_error", "tz_version", "tz_device_count", "tz_state_from_tps", "tz_state_to_tps", "tz_move_to_ptn",
    "tz_move_from_ptn", "tz_policy_size", "tz_input_channels", "tz_net_create", "tz_net_load_weights",
    "tz_net_load_weights_mem", "tz_net_destroy", "tz_net_eval", "tz_net_encode", "tz_net_forward_raw", "tz_net_hash_indices", "tz_net_load_bitset", "tz_net_save_bitset",
    "tz_search_create", "tz_search_destroy", "tz_search_set_positions", "tz_search_get_positions",
    "tz_search_new_openings", "tz_search_simulate", "tz_search_apply_noise", "tz_search_root_info",
    "tz_search_root_children", "tz_search_node", "tz_search_select_best_actions", "tz_search_improved_policy", "tz_search_ube_target",
    "tz_search_step", "tz_search_restart_terminal", "tz_search_gumbel_sh", "tz_search_counters", "tz_search_sync", "tz_search_pool_usage",
    "tz_search_profile", "tz_device_math", "tz_debug_conv_bench", "tz_debug_tower_bench", "tz_net_load_prepare", "tz_net_load_commit", "tz_net_load_discard", "tz_debug_net_clock", "tz_debug_net_stamps", "tz_search_terminal_details", "tz_search_play_moves",
    "tz_trainer_create", "tz_trainer_destroy", "tz_trainer_tensor_count", "tz_trainer_tensor_info",
    "tz_trainer_set_tensor", "tz_trainer_get_tensor", "tz_trainer_step", "tz_trainer_outputs", "tz_trainer_activation",
    "tz_format_targets", "tz_parse_targets", "tz_search_improved_policy_each", "tz_search_shape",
    "tz_selfplay_create", "tz_selfplay_destroy", "tz_selfplay_play_move", "tz_selfplay_counters", "tz_selfplay_take_text",
    "tz_selfplay_run", "tz_reanalyze_create", "tz_reanalyze_destroy", "tz_reanalyze_feed", "tz_reanalyze_iterate",
    "tz_reanalyze_take_text", "tz_reanalyze_run", "tz_compete", "tz_puzzle_benchmark", "tz_search_pool_overflows", "tz_trainer_shape",
    "tz_net_init_random", "tz_net_load_partial", "tz_net_save", "tz_net_clone", "tz_net_get_tensor", "tz_weights_convert", "tz_net_tensor_count", "tz_net_tensor_info",
    "tz_comm_unique_id", "tz_comm_rendezvous_id", "tz_comm_create_rccl", "tz_comm_create_fs", "tz_comm_destroy", "tz_comm_info",
    "tz_comm_all_gather", "tz_comm_take", "tz_comm_broadcast", "tz_comm_barrier", "tz_net_broadcast", "tz_selfplay_set_comm", "tz_selfplay_exchange",
    "tz_trainer_load", "tz_trainer_save", "tz_trainer_from_net", "tz_trainer_to_net", "tz_learn_set_save_points",
    "tz_learn_create", "tz_learn_destroy", "tz_learn_feed", "tz_learn_add_lines", "tz_learn_buffer_len", "tz_learn_step", "tz_learn_run", "tz_learn_last_batch",
]

_lib = None


class TakzeroError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtakzero_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not built yet: compile it in-tree if hipcc is here (that is still the HIP path, never a CPU substitute)
        try:
            from . import build as _build

            _build.build()
        except Exception as e:
            raise ImportError("%s is missing and could not be built (%s): run `python -m takzero_amd.build` "
                              "(hipcc, gfx950).  takzero_amd has no CPU fallback." % (LIB_PATH, e))
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.tz_last_error.restype = C.c_char_p
    lib.tz_state_from_tps.argtypes = [C.c_char_p, ci, ci, vp]
    lib.tz_state_to_tps.argtypes = [vp, C.c_char_p, ci]
    lib.tz_move_to_ptn.argtypes = [ci, C.c_uint16, C.c_char_p, ci]
    lib.tz_move_from_ptn.argtypes = [ci, C.c_char_p, C.POINTER(C.c_uint16)]
    lib.tz_net_create.argtypes = [ci, ci, ci, ci, ci, C.POINTER(vp)]
    lib.tz_net_load_weights.argtypes = [vp, C.c_char_p]
    lib.tz_net_load_weights_mem.argtypes = [vp, vp, C.c_size_t]
    lib.tz_net_destroy.argtypes = [vp]
    lib.tz_net_init_random.argtypes = [vp, C.c_uint64]
    lib.tz_net_load_partial.argtypes = [vp, C.c_char_p, C.c_char_p, ci, C.POINTER(ci)]
    lib.tz_net_save.argtypes = [vp, C.c_char_p]
    lib.tz_net_clone.argtypes = [vp, ci, C.POINTER(vp)]
    lib.tz_net_get_tensor.argtypes = [vp, C.c_char_p, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.tz_weights_convert.argtypes = [C.c_char_p, C.c_char_p]
    lib.tz_net_tensor_count.argtypes = [vp]
    lib.tz_comm_unique_id.argtypes = [vp]
    lib.tz_comm_rendezvous_id.argtypes = [C.c_char_p, ci, vp, C.c_double]
    lib.tz_comm_create_rccl.argtypes = [vp, ci, ci, ci, C.POINTER(vp)]
    lib.tz_comm_create_fs.argtypes = [C.c_char_p, ci, ci, C.c_double, C.POINTER(vp)]
    lib.tz_comm_destroy.argtypes = [vp]
    lib.tz_comm_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tz_comm_all_gather.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]
    lib.tz_comm_take.argtypes = [vp, vp, C.c_uint64]
    lib.tz_comm_broadcast.argtypes = [vp, vp, C.c_uint64, ci]
    lib.tz_comm_barrier.argtypes = [vp]
    lib.tz_net_broadcast.argtypes = [vp, vp, ci, ci]
    lib.tz_selfplay_set_comm.argtypes = [vp, vp, ci]
    lib.tz_selfplay_exchange.argtypes = [vp]
    lib.tz_trainer_load.argtypes = [vp, C.c_char_p]
    lib.tz_trainer_save.argtypes = [vp, C.c_char_p]
    lib.tz_trainer_from_net.argtypes = [vp, vp]
    lib.tz_trainer_to_net.argtypes = [vp, vp]
    lib.tz_learn_set_save_points.argtypes = [vp, ci, ci, vp]
    lib.tz_net_tensor_info.argtypes = [vp, ci, C.c_char_p, ci, C.POINTER(C.c_uint64)]
    lib.tz_net_eval.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp]
    lib.tz_net_encode.argtypes = [vp, ci, vp, vp]
    lib.tz_net_forward_raw.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.tz_net_hash_indices.argtypes = [vp, ci, vp, vp, ci]
    lib.tz_net_load_bitset.argtypes = [vp, C.c_char_p]
    lib.tz_net_save_bitset.argtypes = [vp, C.c_char_p]
    lib.tz_search_create.argtypes = [vp, ci, ci, ci, ci, ci, C.POINTER(vp)]
    lib.tz_search_destroy.argtypes = [vp]
    lib.tz_search_set_positions.argtypes = [vp, ci, vp, vp]
    lib.tz_search_get_positions.argtypes = [vp, vp]
    lib.tz_search_new_openings.argtypes = [vp, vp]
    lib.tz_search_simulate.argtypes = [vp, vp, ci]
    lib.tz_search_apply_noise.argtypes = [vp, vp, ci, cf]
    lib.tz_search_root_info.argtypes = [vp, vp]
    lib.tz_search_root_children.argtypes = [vp, ci] + [vp] * 7
    lib.tz_search_node.argtypes = [vp, ci, vp, ci, vp, ci] + [vp] * 7
    lib.tz_search_select_best_actions.argtypes = [vp, vp]
    lib.tz_search_improved_policy.argtypes = [vp, cf, ci, vp]
    lib.tz_search_improved_policy_each.argtypes = [vp, vp, ci, vp]
    lib.tz_search_ube_target.argtypes = [vp, cf, vp]
    lib.tz_search_step.argtypes = [vp, vp]
    lib.tz_search_restart_terminal.argtypes = [vp, vp, vp]
    lib.tz_search_gumbel_sh.argtypes = [vp, vp, ci, ci, vp, ci, vp]
    lib.tz_search_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tz_search_sync.argtypes = [vp]
    lib.tz_search_pool_overflows.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.tz_search_pool_usage.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.tz_search_profile.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint64)]
    lib.tz_search_terminal_details.argtypes = [vp, vp, vp]
    lib.tz_search_play_moves.argtypes = [vp, vp, vp]
    lib.tz_device_math.argtypes = [ci, vp, vp, vp, ci]
    lib.tz_debug_conv_bench.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_float)]
    lib.tz_debug_tower_bench.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_float)]
    lib.tz_debug_net_clock.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.tz_debug_net_stamps.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]
    lib.tz_net_load_prepare.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.tz_net_load_commit.argtypes = [vp, vp]
    lib.tz_net_load_discard.argtypes = [vp]
    lib.tz_format_targets.argtypes = [ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.tz_parse_targets.argtypes = [vp, C.c_uint64, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.tz_search_shape.argtypes = [vp] + [C.POINTER(ci)] * 4
    lib.tz_selfplay_create.argtypes = [vp, ci, C.c_uint64, ci, ci, ci, ci, C.POINTER(vp)]
    lib.tz_selfplay_destroy.argtypes = [vp]
    lib.tz_selfplay_play_move.argtypes = [vp]
    lib.tz_selfplay_counters.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 3
    lib.tz_selfplay_take_text.argtypes = [vp, ci, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.tz_selfplay_run.argtypes = [vp, C.c_char_p, ci, ci, C.c_char_p, vp, vp, C.c_double]
    lib.tz_reanalyze_create.argtypes = [vp, ci, C.c_uint64, ci, ci, ci, ci, C.POINTER(vp)]
    lib.tz_reanalyze_destroy.argtypes = [vp]
    lib.tz_reanalyze_feed.argtypes = [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tz_reanalyze_iterate.argtypes = [vp]
    lib.tz_reanalyze_take_text.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.tz_reanalyze_run.argtypes = [vp, C.c_char_p, ci, ci, C.c_char_p, vp, vp, C.c_double]
    lib.tz_compete.argtypes = [vp, vp, vp, cf, cf, C.c_uint64, ci, ci, ci, vp]
    lib.tz_puzzle_benchmark.argtypes = [vp, vp, vp, ci, ci, C.c_uint64, ci, ci, vp]
    lib.tz_trainer_shape.argtypes = [vp] + [C.POINTER(ci)] * 3
    lib.tz_learn_create.argtypes = [vp, ci, C.c_uint64, ci, ci, C.POINTER(vp)]
    lib.tz_learn_destroy.argtypes = [vp]
    lib.tz_learn_feed.argtypes = [vp, ci, C.c_char_p, ci, C.POINTER(C.c_uint64)]
    lib.tz_learn_add_lines.argtypes = [vp, ci, vp, C.c_uint64, ci, C.POINTER(C.c_uint64)]
    lib.tz_learn_buffer_len.argtypes = [vp, ci, C.POINTER(C.c_uint64)]
    lib.tz_learn_step.argtypes = [vp, ci, ci, ci, vp]
    lib.tz_learn_last_batch.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.tz_learn_run.argtypes = [vp, C.c_char_p, C.c_int64, C.c_int64, ci, ci, C.c_int64, C.c_double, C.c_double, C.c_double, vp, vp,
                                 C.POINTER(C.c_int64)]
    lib.tz_trainer_create.argtypes = [ci, ci, ci, ci, ci, cf, C.POINTER(vp)]
    lib.tz_trainer_destroy.argtypes = [vp]
    lib.tz_trainer_tensor_count.argtypes = [vp]
    lib.tz_trainer_tensor_info.argtypes = [vp, ci, C.c_char_p, ci, C.POINTER(C.c_uint64)]
    lib.tz_trainer_set_tensor.argtypes = [vp, C.c_char_p, ci, vp, C.c_uint64]
    lib.tz_trainer_get_tensor.argtypes = [vp, C.c_char_p, ci, vp, C.c_uint64]
    lib.tz_trainer_step.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.tz_trainer_outputs.argtypes = [vp, vp, vp, vp]
    lib.tz_trainer_activation.argtypes = [vp, ci, vp, C.c_uint64]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise TakzeroError(rc, load().tz_last_error().decode(errors="replace"))
