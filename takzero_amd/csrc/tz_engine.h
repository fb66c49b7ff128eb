// tz_engine.h — host-side structures shared by the tree (tz_tree.hip), network (tz_nn.hip) and
// C-ABI (tz_capi.cpp) translation units of libtakzero_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/takzero_hip.h"

#define TZ_MAX_DEPTH 512

void tz_set_error(const std::string& msg);
int tz_fail(int code, const std::string& msg);
#define TZ_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return tz_fail(TZ_EDEVICE, std::string(#call) + ": " + hipGetErrorString(_e));       \
    } while (0)

// ----------------------------------------------------------------------------------------------
// Tree pools: structure of arrays, one slab of `cap` node slots per (bank, game).  A node's
// children are contiguous [child0, child0 + nchild).  Index 0 of a game's slab is its root.
// Two banks so that BatchedMCTS::step (subtree reuse, node/mod.rs:95-102) can compact the kept
// subtree into the other bank.
struct TreeArrays {
    uint8_t* eval_tag;    // Eval discriminant (eval.rs:8-13)
    uint32_t* eval_bits;  // f32 bits of Value, or ply of Win/Loss/Draw
    uint32_t* visits;     // visit_count
    float* prob;          // probability
    float* logit;         // logit
    float* std_dev;       // std_dev
    uint32_t* child0;
    uint16_t* nchild;
    uint16_t* action;     // move_index of the edge leading to this node
};

struct SearchDev {
    int batch, n, half_komi, cap, max_actions, agent_kind;
    int strict_capacity;  // 1: a full node pool is an error (TZ_ECAPACITY); 0: the leaf is evaluated but not expanded, and counted
    TreeArrays t;       // arrays of size 2 * batch * cap
    uint8_t* bank;      // [batch] current bank of each game
    uint32_t* alloc;    // [batch] next free slot in the current bank
    tz_state* env;      // [batch] root positions
    float* betas;       // [batch]
    // per-simulation scratch
    uint32_t* traj;       // [batch][TZ_MAX_DEPTH] node indices, root first
    uint32_t* traj_len;   // [batch]
    int32_t* start_node;  // [batch] node to start the descent from (0 = root), gumbel uses children
    uint8_t* leaf_kind;   // [batch] 0 = finished (Known), 1 = needs network
    uint16_t* leaf_nact;  // [batch]
    uint16_t* leaf_act;   // [batch][max_actions]
    tz_state* leaf_env;   // [batch]
    int32_t* nn_game;     // [batch] compacted list of games that need the network
    int32_t* nn_count;    // [1]
    uint32_t* bfs_src;    // [batch][cap] scratch for subtree compaction
    unsigned long long* counters;  // [0] simulations, [1] nn leaf evals, [2] expansions skipped because a game's pool was full
    int32_t* error_flag;  // [1] sticky: 1 node pool overflow, 2 depth overflow, 3 action overflow
    int8_t* term_reason;  // [batch] reason of the last terminal seen by restart_kernel (0 none, 1 road, 2 flats, 3 reversible plies)
    uint8_t* term_winner; // [batch] 0 white, 1 black, 2 draw
};

// network outputs consumed by the expand kernel (device pointers, indexed by nn slot)
struct NetOut {
    const float* policy;  // [slots][NN][policy_stride]  (NHWC: pixel-major, channel contiguous)
    int policy_stride;
    const float* value;     // [slots]
    const float* variance;  // [slots]
};

// tz_tree.hip
int tz_tree_descend(const SearchDev& s, bool from_start_nodes, hipStream_t st);
int tz_tree_compact_leaves(const SearchDev& s, hipStream_t st);
int tz_tree_expand(const SearchDev& s, const NetOut& out, hipStream_t st);
int tz_tree_apply_noise(const SearchDev& s, const float* noise_dev, int amax, float ratio, hipStream_t st);
int tz_tree_root_info(const SearchDev& s, tz_root_info* out_dev, hipStream_t st);
int tz_tree_root_children(const SearchDev& s, int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag,
                          uint32_t* eval_bits, float* logit, float* prob, float* std_dev, hipStream_t st);
int tz_tree_node(const SearchDev& s, int game, const uint16_t* path_dev, int len, uint32_t* out_words, int* status, int amax,
                 uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob, float* std_dev,
                 hipStream_t st);
int tz_tree_select_best(const SearchDev& s, uint16_t* out_dev, hipStream_t st);
int tz_tree_step(const SearchDev& s, const uint16_t* actions_dev, hipStream_t st);
int tz_tree_restart(const SearchDev& s, const int32_t* choice_dev, int8_t* terminal_dev, bool force_all,
                    bool with_opening_moves, hipStream_t st);
int tz_tree_reset_games(const SearchDev& s, const int32_t* idx_dev, int count, hipStream_t st);
int tz_tree_set_start_children(const SearchDev& s, const int32_t* child_index_dev, hipStream_t st);
int tz_tree_gumbel_root_fixup(const SearchDev& s, hipStream_t st);
int tz_tree_play_moves(const SearchDev& s, const uint16_t* actions_dev, int8_t* ok_dev, hipStream_t st);
