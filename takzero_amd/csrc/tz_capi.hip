// tz_capi.hip — BatchedMCTS surface of the C ABI (include/takzero_hip.h) over the tree kernels
// (tz_tree.hip) and the network (tz_nn.hip).  Reference: takzero/src/search/node/batched.rs:32-409.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tz_math.h"
#include "tz_nn.h"

struct tz_search {
    SearchDev d;
    tz_net* net = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device staging
    float* noise_dev = nullptr;
    int noise_cap = 0;
    uint16_t* act_dev = nullptr;
    int32_t* i32_dev = nullptr;
    int8_t* i8_dev = nullptr;
    tz_root_info* info_dev = nullptr;
    void* child_dev = nullptr;  // staging for root_children
    size_t child_cap = 0;
    // one lock-step simulation captured as a HIP graph (index 0: from the roots, 1: from start nodes):
    // ~60 launches per simulation would otherwise leave the GPU waiting for the host
    hipGraphExec_t graph[2] = {nullptr, nullptr};
    int warm[2] = {0, 0};
    uint64_t graph_gen[2] = {0, 0};  // tz_net::weights_gen the graph was captured with
    bool use_graph = true;
    uint64_t sim_index = 0;
    // profiling
    bool profile = false;
    double tree_ms = 0.0;
    uint64_t steps = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tree_events;
};

namespace {

// node slots per game and bank; 62 B each: 4096 games of 5x5 = 66 GB, 2048 games of 6x6 = 66 GB of the 288 GB
int default_capacity(int n) { return n <= 3 ? 16384 : n == 4 ? 65536 : n == 5 ? 262144 : 524288; }
int default_max_actions(int n) { return n <= 3 ? 64 : n == 4 ? 192 : n == 5 ? 512 : 1024; }

template <typename T>
int dev_alloc(T** p, size_t count) {
    TZ_HIP(hipMalloc(p, count * sizeof(T)));
    return TZ_OK;
}

int check_error_flag(tz_search* s) {
    int32_t flag = 0;
    TZ_HIP(hipMemcpyAsync(&flag, s->d.error_flag, 4, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    if (!flag) return TZ_OK;
    TZ_HIP(hipMemsetAsync(s->d.error_flag, 0, 4, s->stream));
    switch (flag) {
        case 1: return tz_fail(TZ_ECAPACITY, "search: a game's node pool overflowed (raise node_capacity)");
        case 2: return tz_fail(TZ_ECAPACITY, "search: tree depth exceeded TZ_MAX_DEPTH");
        case 3: return tz_fail(TZ_ECAPACITY, "search: a position has more legal moves than max_actions");
        case 4: return tz_fail(TZ_ESTATE, "search: no child eligible for selection (policy.rs:94 expect)");
        case 6: return tz_fail(TZ_ESTATE, "search: noise applied to an un-expanded root (noise.rs:12-15 assert)");
        case 7: return tz_fail(TZ_ENUMERIC, "search: the network produced NaN (net5.rs:263 / mcts.rs:194 expect)");
        default: return tz_fail(TZ_EDEVICE, "search: unknown device error flag");
    }
}

int drain_profile(tz_search* s) {
    for (auto& ev : s->tree_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) s->tree_ms += ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    s->tree_events.clear();
    if (s->net) {
        for (auto& ev : s->net->conv_events) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) s->net->conv_ms += ms;
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        s->net->conv_events.clear();
    }
    return TZ_OK;
}

int one_simulation_eager(tz_search* s, bool from_start, bool profile);

// one lock-step simulation for every game (batched.rs:63-128)
int one_simulation(tz_search* s, bool from_start) {
    const int gi = from_start ? 1 : 0;
    s->sim_index++;
    // with profiling on, every 8th simulation runs eagerly with HIP events around its kernels
    const bool sample = s->profile && (s->sim_index % 8 == 0);
    if (!s->use_graph || sample || s->warm[gi] < 2) {
        s->warm[gi]++;
        return one_simulation_eager(s, from_start, sample);
    }
    if (s->graph[gi] && s->net && s->graph_gen[gi] != s->net->weights_gen) {
        // hot reload (selfplay/src/main.rs:107-110): the captured launches point at the old weights
        (void)hipGraphExecDestroy(s->graph[gi]);
        s->graph[gi] = nullptr;
    }
    if (!s->graph[gi]) {
        hipGraph_t g = nullptr;
        TZ_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        const int rc = one_simulation_eager(s, from_start, false);
        const hipError_t e = hipStreamEndCapture(s->stream, &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&s->graph[gi], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        s->graph_gen[gi] = s->net ? s->net->weights_gen : 0;
    }
    TZ_HIP(hipGraphLaunch(s->graph[gi], s->stream));
    return TZ_OK;
}

int one_simulation_eager(tz_search* s, bool from_start, bool profile) {
    int rc;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    const bool saved_net_profile = s->net ? s->net->profile : false;
    if (s->net) s->net->profile = profile;
    struct Restore {
        tz_search* s;
        bool v;
        ~Restore() {
            if (s->net) s->net->profile = v;
        }
    } restore{s, saved_net_profile};
    if (profile) {
        TZ_HIP(hipEventCreate(&e0));
        TZ_HIP(hipEventCreate(&e1));
        TZ_HIP(hipEventCreate(&e2));
        TZ_HIP(hipEventCreate(&e3));
        TZ_HIP(hipEventRecord(e0, s->stream));
    }
    if ((rc = tz_tree_descend(s->d, from_start, s->stream))) return rc;
    if ((rc = tz_tree_compact_leaves(s->d, s->stream))) return rc;
    if (profile) TZ_HIP(hipEventRecord(e1, s->stream));
    NetOut out{nullptr, 0, nullptr, nullptr};
    if (s->d.agent_kind == TZ_AGENT_NET) {
        if ((rc = tz_net_forward_device(s->net, s->d.leaf_env, s->d.nn_game, s->d.nn_count, 0, s->d.batch, s->stream, &out)))
            return rc;
    }
    if (profile) TZ_HIP(hipEventRecord(e2, s->stream));
    if ((rc = tz_tree_expand(s->d, out, s->stream))) return rc;
    if (profile) {
        TZ_HIP(hipEventRecord(e3, s->stream));
        s->tree_events.push_back({e0, e1});
        s->tree_events.push_back({e2, e3});
        s->steps++;
    }
    return TZ_OK;
}

int ensure_noise(tz_search* s, int amax) {
    if ((size_t)amax * s->d.batch <= (size_t)s->noise_cap) return TZ_OK;
    if (s->noise_dev) (void)hipFree(s->noise_dev);
    s->noise_dev = nullptr;
    TZ_HIP(hipMalloc(&s->noise_dev, (size_t)amax * s->d.batch * sizeof(float)));
    s->noise_cap = amax * s->d.batch;
    return TZ_OK;
}

int ensure_child(tz_search* s, size_t bytes) {
    if (bytes <= s->child_cap) return TZ_OK;
    if (s->child_dev) (void)hipFree(s->child_dev);
    s->child_dev = nullptr;
    TZ_HIP(hipMalloc(&s->child_dev, bytes));
    s->child_cap = bytes;
    return TZ_OK;
}

}  // namespace

extern "C" {

int tz_search_create(tz_net* net, int agent_kind, int batch, int board_n, int half_komi, int node_capacity,
                     tz_search** out) {
    if (!out || batch <= 0 || board_n < 3 || board_n > 6) return tz_fail(TZ_EINVAL, "tz_search_create: bad argument");
    if (agent_kind != TZ_AGENT_NET && agent_kind != TZ_AGENT_DUMMY && agent_kind != TZ_AGENT_SIMPLE)
        return tz_fail(TZ_EINVAL, "tz_search_create: unknown agent kind");
    if (agent_kind == TZ_AGENT_NET && (!net || net->n != board_n))
        return tz_fail(TZ_EINVAL, "tz_search_create: TZ_AGENT_NET needs a network of the same board size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return tz_fail(TZ_EDEVICE, "tz_search_create: no HIP device available (the HIP path has no CPU fallback)");
    tz_search* s = new tz_search();
    s->use_graph = getenv("TZ_NO_GRAPH") == nullptr;
    s->net = agent_kind == TZ_AGENT_NET ? net : nullptr;
    s->device = s->net ? s->net->device : 0;
    TZ_HIP(hipSetDevice(s->device));
    if (s->net) {
        s->stream = s->net->stream;
    } else {
        TZ_HIP(hipStreamCreate(&s->stream));
        s->own_stream = true;
    }
    SearchDev& d = s->d;
    memset(&d, 0, sizeof d);
    d.batch = batch;
    d.n = board_n;
    d.half_komi = half_komi;
    d.cap = node_capacity > 0 ? node_capacity : default_capacity(board_n);
    d.strict_capacity = getenv("TZ_STRICT_CAPACITY") != nullptr;
    d.max_actions = default_max_actions(board_n);
    d.agent_kind = agent_kind;
    const size_t nodes = (size_t)2 * batch * d.cap;
    int rc = 0;
    rc |= dev_alloc(&d.t.eval_tag, nodes);
    rc |= dev_alloc(&d.t.eval_bits, nodes);
    rc |= dev_alloc(&d.t.visits, nodes);
    rc |= dev_alloc(&d.t.prob, nodes);
    rc |= dev_alloc(&d.t.logit, nodes);
    rc |= dev_alloc(&d.t.std_dev, nodes);
    rc |= dev_alloc(&d.t.child0, nodes);
    rc |= dev_alloc(&d.t.nchild, nodes);
    rc |= dev_alloc(&d.t.action, nodes);
    rc |= dev_alloc(&d.bank, batch);
    rc |= dev_alloc(&d.alloc, batch);
    rc |= dev_alloc(&d.env, batch);
    rc |= dev_alloc(&d.betas, batch);
    rc |= dev_alloc(&d.traj, (size_t)batch * TZ_MAX_DEPTH);
    rc |= dev_alloc(&d.traj_len, batch);
    rc |= dev_alloc(&d.start_node, batch);
    rc |= dev_alloc(&d.leaf_kind, batch);
    rc |= dev_alloc(&d.leaf_nact, batch);
    rc |= dev_alloc(&d.leaf_act, (size_t)batch * d.max_actions);
    rc |= dev_alloc(&d.leaf_env, batch);
    rc |= dev_alloc(&d.nn_game, batch);
    rc |= dev_alloc(&d.nn_count, 1);
    rc |= dev_alloc(&d.bfs_src, (size_t)batch * d.cap);
    rc |= dev_alloc(&d.counters, 3);
    rc |= dev_alloc(&d.error_flag, 1);
    rc |= dev_alloc(&d.term_reason, batch);
    rc |= dev_alloc(&d.term_winner, batch);
    rc |= dev_alloc(&s->act_dev, batch);
    rc |= dev_alloc(&s->i32_dev, batch);
    rc |= dev_alloc(&s->i8_dev, batch);
    rc |= dev_alloc(&s->info_dev, batch);
    if (rc) {
        tz_search_destroy(s);
        return tz_fail(TZ_ENOMEM, "tz_search_create: device allocation failed (lower node_capacity or batch)");
    }
    TZ_HIP(hipMemsetAsync(d.bank, 0, batch, s->stream));
    TZ_HIP(hipMemsetAsync(d.env, 0, (size_t)batch * sizeof(tz_state), s->stream));
    TZ_HIP(hipMemsetAsync(d.leaf_env, 0, (size_t)batch * sizeof(tz_state), s->stream));
    TZ_HIP(hipMemsetAsync(d.betas, 0, batch * sizeof(float), s->stream));
    TZ_HIP(hipMemsetAsync(d.start_node, 0, batch * sizeof(int32_t), s->stream));
    TZ_HIP(hipMemsetAsync(d.counters, 0, 3 * sizeof(unsigned long long), s->stream));
    TZ_HIP(hipMemsetAsync(d.error_flag, 0, 4, s->stream));
    TZ_HIP(hipMemsetAsync(d.nn_count, 0, 4, s->stream));
    if (s->net && (rc = tz_net_ensure_batch(s->net, batch))) {
        tz_search_destroy(s);
        return rc;
    }
    // default envs + fresh trees (BatchedMCTS::from_envs with Game::default())
    if ((rc = tz_tree_restart(d, nullptr, nullptr, true, false, s->stream))) {
        tz_search_destroy(s);
        return rc;
    }
    TZ_HIP(hipStreamSynchronize(s->stream));
    *out = s;
    return TZ_OK;
}

int tz_search_destroy(tz_search* s) {
    if (!s) return TZ_OK;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    drain_profile(s);
    for (int i = 0; i < 2; i++)
        if (s->graph[i]) (void)hipGraphExecDestroy(s->graph[i]);
    SearchDev& d = s->d;
    void* ptrs[] = {d.t.eval_tag, d.t.eval_bits, d.t.visits, d.t.prob, d.t.logit, d.t.std_dev, d.t.child0, d.t.nchild,
                    d.t.action, d.bank, d.alloc, d.env, d.betas, d.traj, d.traj_len, d.start_node, d.leaf_kind,
                    d.leaf_nact, d.leaf_act, d.leaf_env, d.nn_game, d.nn_count, d.bfs_src, d.counters, d.error_flag,
                    d.term_reason, d.term_winner,
                    s->noise_dev, s->act_dev, s->i32_dev, s->i8_dev, s->info_dev, s->child_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return TZ_OK;
}

int tz_search_set_positions(tz_search* s, int count, const int32_t* game_idx, const tz_state* states) {
    if (!s || count < 0 || (count && (!game_idx || !states))) return tz_fail(TZ_EINVAL, "tz_search_set_positions: bad argument");
    if (count == 0) return TZ_OK;
    TZ_HIP(hipSetDevice(s->device));
    if (count > s->d.batch) return tz_fail(TZ_EINVAL, "tz_search_set_positions: more positions than games");
    for (int i = 0; i < count; i++) {
        if (game_idx[i] < 0 || game_idx[i] >= s->d.batch) return tz_fail(TZ_EINVAL, "tz_search_set_positions: game index out of range");
        if (states[i].n != s->d.n) return tz_fail(TZ_EINVAL, "tz_search_set_positions: board size mismatch");
    }
    TZ_HIP(hipStreamSynchronize(s->stream));
    for (int i = 0; i < count; i++)
        TZ_HIP(hipMemcpyAsync(&s->d.env[game_idx[i]], &states[i], sizeof(tz_state), hipMemcpyHostToDevice, s->stream));
    TZ_HIP(hipMemcpyAsync(s->i32_dev, game_idx, count * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    int rc = tz_tree_reset_games(s->d, s->i32_dev, count, s->stream);
    if (rc) return rc;
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_get_positions(tz_search* s, tz_state* states_out) {
    if (!s || !states_out) return tz_fail(TZ_EINVAL, "tz_search_get_positions: null argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipMemcpyAsync(states_out, s->d.env, (size_t)s->d.batch * sizeof(tz_state), hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_new_openings(tz_search* s, const int32_t* opening_choice) {
    if (!s || !opening_choice) return tz_fail(TZ_EINVAL, "tz_search_new_openings: null argument");
    TZ_HIP(hipSetDevice(s->device));
    for (int g = 0; g < s->d.batch; g++)
        if (opening_choice[g] < 0 || opening_choice[g] >= 16) return tz_fail(TZ_EINVAL, "opening choice must be in [0,16)");
    TZ_HIP(hipMemcpyAsync(s->i32_dev, opening_choice, s->d.batch * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    int rc = tz_tree_restart(s->d, s->i32_dev, nullptr, true, true, s->stream);
    if (rc) return rc;
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_simulate(tz_search* s, const float* betas, int n_sims) {
    if (!s || !betas || n_sims < 0) return tz_fail(TZ_EINVAL, "tz_search_simulate: bad argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipMemcpyAsync(s->d.betas, betas, s->d.batch * sizeof(float), hipMemcpyHostToDevice, s->stream));
    for (int i = 0; i < n_sims; i++) {
        int rc = one_simulation(s, false);
        if (rc) return rc;
    }
    int rc = check_error_flag(s);
    drain_profile(s);
    return rc;
}

int tz_search_apply_noise(tz_search* s, const float* noise, int amax, float ratio) {
    if (!s || !noise || amax <= 0) return tz_fail(TZ_EINVAL, "tz_search_apply_noise: bad argument");
    TZ_HIP(hipSetDevice(s->device));
    int rc = ensure_noise(s, amax);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(s->noise_dev, noise, (size_t)amax * s->d.batch * sizeof(float), hipMemcpyHostToDevice, s->stream));
    if ((rc = tz_tree_apply_noise(s->d, s->noise_dev, amax, ratio, s->stream))) return rc;
    return check_error_flag(s);
}

int tz_search_root_info(tz_search* s, tz_root_info* out) {
    if (!s || !out) return tz_fail(TZ_EINVAL, "tz_search_root_info: null argument");
    TZ_HIP(hipSetDevice(s->device));
    int rc = tz_tree_root_info(s->d, s->info_dev, s->stream);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(out, s->info_dev, (size_t)s->d.batch * sizeof(tz_root_info), hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_root_children(tz_search* s, int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag,
                            uint32_t* eval_bits, float* logit, float* prob, float* std_dev) {
    if (!s || amax <= 0) return tz_fail(TZ_EINVAL, "tz_search_root_children: bad argument");
    TZ_HIP(hipSetDevice(s->device));
    const size_t cells = (size_t)s->d.batch * amax;
    int rc = ensure_child(s, cells * (2 + 4 + 1 + 4 + 4 + 4 + 4 + 8));
    if (rc) return rc;
    unsigned char* base = (unsigned char*)s->child_dev;
    uint32_t* d_vis = (uint32_t*)base;
    uint32_t* d_bits = d_vis + cells;
    float* d_logit = (float*)(d_bits + cells);
    float* d_prob = d_logit + cells;
    float* d_std = d_prob + cells;
    uint16_t* d_move = (uint16_t*)(d_std + cells);
    uint8_t* d_tag = (uint8_t*)(d_move + cells);
    if ((rc = tz_tree_root_children(s->d, amax, move_idx ? d_move : nullptr, visits ? d_vis : nullptr, eval_tag ? d_tag : nullptr,
                                    eval_bits ? d_bits : nullptr, logit ? d_logit : nullptr, prob ? d_prob : nullptr,
                                    std_dev ? d_std : nullptr, s->stream)))
        return rc;
    if (move_idx) TZ_HIP(hipMemcpyAsync(move_idx, d_move, cells * 2, hipMemcpyDeviceToHost, s->stream));
    if (visits) TZ_HIP(hipMemcpyAsync(visits, d_vis, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (eval_tag) TZ_HIP(hipMemcpyAsync(eval_tag, d_tag, cells, hipMemcpyDeviceToHost, s->stream));
    if (eval_bits) TZ_HIP(hipMemcpyAsync(eval_bits, d_bits, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (logit) TZ_HIP(hipMemcpyAsync(logit, d_logit, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (prob) TZ_HIP(hipMemcpyAsync(prob, d_prob, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (std_dev) TZ_HIP(hipMemcpyAsync(std_dev, d_std, cells * 4, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    // a root with more children than amax cannot be represented
    std::vector<tz_root_info> info(s->d.batch);
    if ((rc = tz_search_root_info(s, info.data()))) return rc;
    for (auto& r : info)
        if ((int)r.n_children > amax) return tz_fail(TZ_EINVAL, "tz_search_root_children: amax smaller than a root's child count");
    return TZ_OK;
}

int tz_search_node(tz_search* s, int game, const uint16_t* path, int path_len, tz_root_info* node_out, int amax, uint16_t* move_idx,
                   uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob, float* std_dev) {
    if (!s || game < 0 || game >= s->d.batch || path_len < 0 || path_len > TZ_MAX_DEPTH || (path_len && !path) || amax <= 0)
        return tz_fail(TZ_EINVAL, "tz_search_node: bad argument");
    TZ_HIP(hipSetDevice(s->device));
    const size_t cells = (size_t)amax;
    int rc = ensure_child(s, cells * (2 + 4 + 1 + 4 + 4 + 4 + 4 + 8) + 64 + 2 * (size_t)TZ_MAX_DEPTH + 64);
    if (rc) return rc;
    unsigned char* base = (unsigned char*)s->child_dev;
    uint32_t* d_vis = (uint32_t*)base;
    uint32_t* d_bits = d_vis + cells;
    float* d_logit = (float*)(d_bits + cells);
    float* d_prob = d_logit + cells;
    float* d_std = d_prob + cells;
    uint32_t* d_words = (uint32_t*)(d_std + cells);     // 8 words
    int* d_status = (int*)(d_words + 8);
    uint16_t* d_move = (uint16_t*)(d_status + 2);
    uint16_t* d_path = d_move + cells;
    uint8_t* d_tag = (uint8_t*)(d_path + TZ_MAX_DEPTH);
    if (path_len) TZ_HIP(hipMemcpyAsync(d_path, path, (size_t)path_len * 2, hipMemcpyHostToDevice, s->stream));
    if ((rc = tz_tree_node(s->d, game, d_path, path_len, d_words, d_status, amax, d_move, d_vis, d_tag, d_bits, d_logit, d_prob, d_std, s->stream)))
        return rc;
    uint32_t words[8];
    int status = 0;
    TZ_HIP(hipMemcpyAsync(words, d_words, sizeof words, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipMemcpyAsync(&status, d_status, sizeof status, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    if (status >= 0) return tz_fail(TZ_EINVAL, "tz_search_node: the path leaves the tree at depth " + std::to_string(status));
    if ((int)words[1] > amax) return tz_fail(TZ_EINVAL, "tz_search_node: amax smaller than the node's child count");
    if (node_out) {
        std::vector<tz_root_info> info(s->d.batch);
        if ((rc = tz_search_root_info(s, info.data()))) return rc;
        tz_root_info r;
        memset(&r, 0, sizeof r);
        r.visit_count = words[0];
        r.n_children = words[1];
        r.eval_tag = (uint8_t)words[2];
        r.eval.ply = words[3];
        memcpy(&r.std_dev, &words[4], 4);
        memcpy(&r.logit, &words[5], 4);
        memcpy(&r.probability, &words[6], 4);
        r.ply = (uint16_t)(info[game].ply + path_len);
        r.is_terminal_env = r.eval_tag != TZ_EVAL_VALUE && r.eval.ply == 0;   // Node::is_terminal (node/mod.rs:106-108)
        *node_out = r;
    }
    if (move_idx) TZ_HIP(hipMemcpyAsync(move_idx, d_move, cells * 2, hipMemcpyDeviceToHost, s->stream));
    if (visits) TZ_HIP(hipMemcpyAsync(visits, d_vis, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (eval_tag) TZ_HIP(hipMemcpyAsync(eval_tag, d_tag, cells, hipMemcpyDeviceToHost, s->stream));
    if (eval_bits) TZ_HIP(hipMemcpyAsync(eval_bits, d_bits, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (logit) TZ_HIP(hipMemcpyAsync(logit, d_logit, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (prob) TZ_HIP(hipMemcpyAsync(prob, d_prob, cells * 4, hipMemcpyDeviceToHost, s->stream));
    if (std_dev) TZ_HIP(hipMemcpyAsync(std_dev, d_std, cells * 4, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_shape(tz_search* s, int* batch_out, int* board_n_out, int* half_komi_out, int* max_actions_out) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_shape: null handle");
    if (batch_out) *batch_out = s->d.batch;
    if (board_n_out) *board_n_out = s->d.n;
    if (half_komi_out) *half_komi_out = s->d.half_komi;
    if (max_actions_out) *max_actions_out = s->d.max_actions;
    return TZ_OK;
}

int tz_search_select_best_actions(tz_search* s, uint16_t* actions_out) {
    if (!s || !actions_out) return tz_fail(TZ_EINVAL, "tz_search_select_best_actions: null argument");
    TZ_HIP(hipSetDevice(s->device));
    int rc = tz_tree_select_best(s->d, s->act_dev, s->stream);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(actions_out, s->act_dev, s->d.batch * sizeof(uint16_t), hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

// host-side post-search statistics over the root children (same f32 expressions as the reference)
namespace {
struct RootData {
    int amax;
    std::vector<tz_root_info> info;
    std::vector<uint32_t> visits, bits;
    std::vector<uint8_t> tag;
    std::vector<float> logit, prob, stdv;
};
int fetch_roots(tz_search* s, RootData& r) {
    r.info.resize(s->d.batch);
    int rc = tz_search_root_info(s, r.info.data());
    if (rc) return rc;
    int amax = 1;
    for (auto& i : r.info) amax = std::max(amax, (int)i.n_children);
    r.amax = amax;
    const size_t cells = (size_t)s->d.batch * amax;
    r.visits.resize(cells);
    r.bits.resize(cells);
    r.tag.resize(cells);
    r.logit.resize(cells);
    r.prob.resize(cells);
    r.stdv.resize(cells);
    return tz_search_root_children(s, amax, nullptr, r.visits.data(), r.tag.data(), r.bits.data(), r.logit.data(), r.prob.data(),
                                   r.stdv.data());
}
struct HEv {
    uint32_t tag, bits;
};
HEv h_negate(HEv e) {
    switch (e.tag) {
        case TZ_EVAL_VALUE: return {TZ_EVAL_VALUE, tz_float_to_bits(-tz_bits_to_float(e.bits))};
        case TZ_EVAL_WIN: return {TZ_EVAL_LOSS, e.bits + 1};
        case TZ_EVAL_DRAW: return {TZ_EVAL_DRAW, e.bits + 1};
        default: return {TZ_EVAL_WIN, e.bits + 1};
    }
}
float h_to_notnan(HEv e) {
    if (e.tag == TZ_EVAL_VALUE) return tz_bits_to_float(e.bits);
    const float base = tz_powif(TZ_DISCOUNT, (int)e.bits);
    return base * (e.tag == TZ_EVAL_WIN ? 1.0f : e.tag == TZ_EVAL_LOSS ? -1.0f : 0.0f);
}
}  // namespace

// Node::improved_policy (policy.rs:20-62) for every root; visitations[g] (or the one value when `each` is null)
static int improved_policy_impl(tz_search* s, float visitations, const float* each, int amax, float* policy_out) {
    if (!s || !policy_out || amax <= 0) return tz_fail(TZ_EINVAL, "tz_search_improved_policy: bad argument");
    RootData r;
    int rc = fetch_roots(s, r);
    if (rc) return rc;
    std::vector<float> p;
    for (int g = 0; g < s->d.batch; g++) {
        const float sq = sqrtf(each ? each[g] : visitations);
        const int nc = (int)r.info[g].n_children;
        if (nc > amax) return tz_fail(TZ_EINVAL, "tz_search_improved_policy: amax too small");
        const HEv root{r.info[g].eval_tag, r.info[g].eval.ply};
        p.resize(nc);
        float mx = 0.0f;
        for (int i = 0; i < nc; i++) {
            const size_t o = (size_t)g * r.amax + i;
            const HEv ce{r.tag[o], r.bits[o]};
            // policy.rs:36-48: an un-expanded, unknown child completes with the parent's evaluation.
            // Children of the root are un-expanded iff they were never visited.
            const bool needs_init = r.visits[o] == 0 && ce.tag == TZ_EVAL_VALUE;
            const float completed = h_to_notnan(needs_init ? root : h_negate(ce));
            p[i] = (completed + r.stdv[o] * 0.0f) * sq + r.logit[o];
            if (i == 0 || p[i] > mx) mx = p[i];
        }
        float sum = 0.0f;
        for (int i = 0; i < nc; i++) {
            p[i] = tz_expf(p[i] - mx);
            sum = sum + p[i];
        }
        for (int i = 0; i < amax; i++) policy_out[(size_t)g * amax + i] = i < nc ? p[i] / sum : 0.0f;
    }
    return TZ_OK;
}

int tz_search_improved_policy(tz_search* s, float visitations, int amax, float* policy_out) {
    return improved_policy_impl(s, visitations, nullptr, amax, policy_out);
}

int tz_search_improved_policy_each(tz_search* s, const float* visitations, int amax, float* policy_out) {
    if (!visitations) return tz_fail(TZ_EINVAL, "tz_search_improved_policy_each: null visitations");
    return improved_policy_impl(s, 0.0f, visitations, amax, policy_out);
}

int tz_search_ube_target(tz_search* s, float beta, float* out) {
    if (!s || !out) return tz_fail(TZ_EINVAL, "tz_search_ube_target: null argument");
    RootData r;
    int rc = fetch_roots(s, r);
    if (rc) return rc;
    for (int g = 0; g < s->d.batch; g++) {
        const int nc = (int)r.info[g].n_children;
        if (r.info[g].eval_tag != TZ_EVAL_VALUE || nc == 0) {
            out[g] = 0.0f;
            continue;
        }
        int best = 0;
        float best_key = 0.0f;
        for (int i = 0; i < nc; i++) {
            const size_t o = (size_t)g * r.amax + i;
            const float key = h_to_notnan(h_negate(HEv{r.tag[o], r.bits[o]})) + r.stdv[o] * beta;
            if (i == 0 || !(key < best_key)) {
                best = i;
                best_key = key;
            }
        }
        const float sd = r.stdv[(size_t)g * r.amax + best];
        out[g] = sd * sd;
    }
    return TZ_OK;
}

int tz_search_step(tz_search* s, const uint16_t* actions) {
    if (!s || !actions) return tz_fail(TZ_EINVAL, "tz_search_step: null argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipMemcpyAsync(s->act_dev, actions, s->d.batch * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
    int rc = tz_tree_step(s->d, s->act_dev, s->stream);
    if (rc) return rc;
    return check_error_flag(s);
}

int tz_search_restart_terminal(tz_search* s, const int32_t* opening_choice, int8_t* terminal_out) {
    if (!s || !opening_choice || !terminal_out) return tz_fail(TZ_EINVAL, "tz_search_restart_terminal: null argument");
    TZ_HIP(hipSetDevice(s->device));
    for (int g = 0; g < s->d.batch; g++)
        if (opening_choice[g] < 0 || opening_choice[g] >= 16) return tz_fail(TZ_EINVAL, "opening choice must be in [0,16)");
    TZ_HIP(hipMemcpyAsync(s->i32_dev, opening_choice, s->d.batch * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    int rc = tz_tree_restart(s->d, s->i32_dev, s->i8_dev, false, true, s->stream);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(terminal_out, s->i8_dev, s->d.batch, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_terminal_details(tz_search* s, int8_t* reason_out, uint8_t* winner_out) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_terminal_details: null argument");
    TZ_HIP(hipSetDevice(s->device));
    if (reason_out) TZ_HIP(hipMemcpyAsync(reason_out, s->d.term_reason, s->d.batch, hipMemcpyDeviceToHost, s->stream));
    if (winner_out) TZ_HIP(hipMemcpyAsync(winner_out, s->d.term_winner, s->d.batch, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_play_moves(tz_search* s, const uint16_t* actions, int8_t* ok_out) {
    if (!s || !actions || !ok_out) return tz_fail(TZ_EINVAL, "tz_search_play_moves: null argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipMemcpyAsync(s->act_dev, actions, s->d.batch * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
    int rc = tz_tree_play_moves(s->d, s->act_dev, s->i8_dev, s->stream);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(ok_out, s->i8_dev, s->d.batch, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_gumbel_sh(tz_search* s, const float* betas, int sampled_actions, int search_budget, const float* gumbel,
                        int amax, uint16_t* selected_out) {
    if (!s || !betas || !gumbel || !selected_out || sampled_actions <= 0 || amax <= 0)
        return tz_fail(TZ_EINVAL, "tz_search_gumbel_sh: bad argument");
    const int lg = 31 - __builtin_clz((unsigned)sampled_actions);
    if (lg == 0 || search_budget % (lg * sampled_actions) != 0)
        return tz_fail(TZ_EINVAL, "the search budget should be a multiple of k*log2(k) (batched.rs:216-220)");
    TZ_HIP(hipSetDevice(s->device));
    const int B = s->d.batch;
    int rc = tz_search_simulate(s, betas, 1);  // batched.rs:223
    if (rc) return rc;
    RootData r;
    if ((rc = fetch_roots(s, r))) return rc;
    std::vector<uint16_t> moves((size_t)B * r.amax);
    if ((rc = tz_search_root_children(s, r.amax, moves.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr))) return rc;
    struct Cand {
        float key;
        int child;
    };
    std::vector<std::vector<Cand>> sets(B);
    for (int g = 0; g < B; g++) {
        const int nc = (int)r.info[g].n_children;
        if (nc == 0) return tz_fail(TZ_ESTATE, "gumbel_sequential_halving on a root without children (batched.rs:260 divides by zero)");
        if (nc > amax) return tz_fail(TZ_EINVAL, "tz_search_gumbel_sh: amax smaller than a root's child count");
        for (int i = 0; i < nc; i++) sets[g].push_back({r.logit[(size_t)g * r.amax + i] + gumbel[(size_t)g * amax + i], i});
        std::stable_sort(sets[g].begin(), sets[g].end(), [](const Cand& a, const Cand& b) { return a.key > b.key; });
        if ((int)sets[g].size() > sampled_actions) sets[g].resize(sampled_actions);
    }
    const int visits_per_step = search_budget / lg;
    int visits_to_most = 0, remaining = sampled_actions;
    std::vector<int32_t> child(B);
    for (int step = 0; step < lg; step++) {
        const int vpa = visits_per_step / remaining;
        for (int i = 0; i < remaining; i++) {
            for (int g = 0; g < B; g++) child[g] = sets[g][i % sets[g].size()].child;
            TZ_HIP(hipMemcpyAsync(s->i32_dev, child.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
            if ((rc = tz_tree_set_start_children(s->d, s->i32_dev, s->stream))) return rc;
            TZ_HIP(hipStreamSynchronize(s->stream));  // child[] is reused below
            for (int v = 0; v < vpa; v++)
                if ((rc = one_simulation(s, true))) return rc;
        }
        if ((rc = check_error_flag(s))) return rc;
        drain_profile(s);
        visits_to_most += vpa;
        remaining /= 2;
        if ((rc = fetch_roots(s, r))) return rc;
        for (int g = 0; g < B; g++) {
            std::vector<std::pair<float, Cand>> keyed;
            for (auto& c : sets[g]) {
                const size_t o = (size_t)g * r.amax + c.child;
                const float q = h_to_notnan(h_negate(HEv{r.tag[o], r.bits[o]}));
                const float sig = (q + r.stdv[o] * betas[g]) * (50.0f + (float)visits_to_most);  // policy.rs:121-128
                keyed.push_back({c.key + sig, c});
            }
            std::stable_sort(keyed.begin(), keyed.end(),
                             [](const std::pair<float, Cand>& a, const std::pair<float, Cand>& b) { return a.first > b.first; });
            sets[g].clear();
            for (size_t j = 0; j < keyed.size() && (int)j < remaining; j++) sets[g].push_back(keyed[j].second);
        }
    }
    for (int g = 0; g < B; g++) selected_out[g] = moves[(size_t)g * r.amax + sets[g][0].child];
    if ((rc = tz_tree_gumbel_root_fixup(s->d, s->stream))) return rc;
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_pool_overflows(tz_search* s, uint64_t* skipped_expansions) {
    if (!s || !skipped_expansions) return tz_fail(TZ_EINVAL, "tz_search_pool_overflows: null argument");
    TZ_HIP(hipSetDevice(s->device));
    unsigned long long c = 0;
    TZ_HIP(hipMemcpyAsync(&c, s->d.counters + 2, sizeof c, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    *skipped_expansions = c;
    return TZ_OK;
}

int tz_search_counters(tz_search* s, uint64_t* simulations, uint64_t* nn_leaf_evals) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_counters: null argument");
    TZ_HIP(hipSetDevice(s->device));
    unsigned long long c[2];
    TZ_HIP(hipMemcpyAsync(c, s->d.counters, sizeof c, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    if (simulations) *simulations = c[0];
    if (nn_leaf_evals) *nn_leaf_evals = c[1];
    return TZ_OK;
}

int tz_search_pool_usage(tz_search* s, uint32_t* max_used, uint32_t* capacity) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_pool_usage: null argument");
    TZ_HIP(hipSetDevice(s->device));
    std::vector<uint32_t> h(s->d.batch);
    TZ_HIP(hipMemcpyAsync(h.data(), s->d.alloc, h.size() * 4, hipMemcpyDeviceToHost, s->stream));
    TZ_HIP(hipStreamSynchronize(s->stream));
    uint32_t m = 0;
    for (uint32_t v : h) m = std::max(m, v);
    if (max_used) *max_used = m;
    if (capacity) *capacity = (uint32_t)s->d.cap;
    return TZ_OK;
}

int tz_search_sync(tz_search* s) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_sync: null argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipStreamSynchronize(s->stream));
    return TZ_OK;
}

int tz_search_profile(tz_search* s, int reset, double* conv_ms, uint64_t* conv_launches, double* tree_ms, uint64_t* steps) {
    if (!s) return tz_fail(TZ_EINVAL, "tz_search_profile: null argument");
    TZ_HIP(hipSetDevice(s->device));
    TZ_HIP(hipStreamSynchronize(s->stream));
    drain_profile(s);
    if (conv_ms) *conv_ms = s->net ? s->net->conv_ms : 0.0;
    if (conv_launches) *conv_launches = s->net ? s->net->conv_launches : 0;
    if (tree_ms) *tree_ms = s->tree_ms;
    if (steps) *steps = s->steps;
    if (reset == 1 || reset == 2) {  // 1: reset and enable, 2: reset and disable
        s->tree_ms = 0.0;
        s->steps = 0;
        s->profile = reset == 1;
        if (s->net) {
            s->net->conv_ms = 0.0;
            s->net->conv_launches = 0;
        }
    }
    return TZ_OK;
}

}  // extern "C"
