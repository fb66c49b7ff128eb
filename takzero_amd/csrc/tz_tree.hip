// tz_tree.hip — batched MCTS on the device: one wavefront per game.
//
// Replaces, for B concurrent games stepped in lock-step (takzero/src/search/node/batched.rs:63-128):
//   Node::forward            mcts.rs:107-138   -> descend_kernel   (select with PUCT policy.rs:78-95)
//   Node::backward_known_eval mcts.rs:141-163  -> descend_kernel   (immediately, as batched.rs:78-82)
//   softmax + backward_network_eval policy.rs:10-19, mcts.rs:171-225 -> expand_kernel
//   propagate_child_eval / node_solver mcts.rs:49-102 -> backup()
//   apply_dirichlet noise.rs:10-26, descend node/mod.rs:95-102, select_best_action node/mod.rs:132-161,
//   step / restart_terminal_envs batched.rs:131-203.
//
// Arithmetic is f32 in exactly the reference's expression order; this file is compiled with
// -ffp-contract=off and uses tz_math.h for exp/ln/powi so results are bit-identical with the
// CPU oracle.  Node statistics are SoA slabs in HBM (tz_engine.h); a node's children are
// contiguous so a child scan is a handful of coalesced loads, and argmax / min / all-known are
// wavefront reductions.
#include "tz_engine.h"
#include "tz_math.h"
#include "tz_tak_dev.h"

using namespace tzd;

namespace {

struct Ev {
    uint32_t tag, bits;
};
__device__ __forceinline__ Ev ev_value(float v) { return Ev{TZ_EVAL_VALUE, tz_float_to_bits(v)}; }
__device__ __forceinline__ bool ev_known(Ev e) { return e.tag != TZ_EVAL_VALUE; }
// eval.rs:40-47
__device__ __forceinline__ Ev ev_negate(Ev e) {
    switch (e.tag) {
        case TZ_EVAL_VALUE: return ev_value(-tz_bits_to_float(e.bits));
        case TZ_EVAL_WIN: return Ev{TZ_EVAL_LOSS, e.bits + 1};
        case TZ_EVAL_DRAW: return Ev{TZ_EVAL_DRAW, e.bits + 1};
        default: return Ev{TZ_EVAL_WIN, e.bits + 1};
    }
}
// eval.rs:95-105
__device__ __forceinline__ float ev_to_f32(Ev e) {
    const float base = tz_powif(TZ_DISCOUNT, e.tag == TZ_EVAL_VALUE ? 0 : (int)e.bits);
    const float x = e.tag == TZ_EVAL_VALUE ? tz_bits_to_float(e.bits)
                    : e.tag == TZ_EVAL_WIN ? 1.0f
                    : e.tag == TZ_EVAL_LOSS ? -1.0f
                                            : 0.0f;
    return base * x;
}
// eval.rs:107-116
__device__ __forceinline__ float ev_to_notnan(Ev e) {
    return e.tag == TZ_EVAL_VALUE ? tz_bits_to_float(e.bits) : ev_to_f32(e);
}
__device__ __forceinline__ int cmpf(float a, float b) { return a < b ? -1 : a > b ? 1 : 0; }
__device__ __forceinline__ int cmpu(uint32_t a, uint32_t b) { return a < b ? -1 : a > b ? 1 : 0; }
// impl Ord for Eval, eval.rs:138-163 (CONTEMPT = -0.05, eval.rs:128)
__device__ __forceinline__ int ev_cmp(Ev a, Ev b) {
    const float contempt = -0.05f;
    switch (a.tag) {
        case TZ_EVAL_VALUE:
            switch (b.tag) {
                case TZ_EVAL_VALUE: return cmpf(tz_bits_to_float(a.bits), tz_bits_to_float(b.bits));
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmpf(tz_bits_to_float(a.bits), contempt);
                default: return 1;
            }
        case TZ_EVAL_WIN: return b.tag == TZ_EVAL_WIN ? cmpu(b.bits, a.bits) : 1;
        case TZ_EVAL_DRAW:
            switch (b.tag) {
                case TZ_EVAL_VALUE: return cmpf(contempt, tz_bits_to_float(b.bits));
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmpu(b.bits, a.bits);
                default: return 1;
            }
        default: return b.tag == TZ_EVAL_LOSS ? cmpu(a.bits, b.bits) : -1;
    }
}

__device__ __forceinline__ size_t slab_base(const SearchDev& s, int bank, int g) {
    return ((size_t)bank * s.batch + g) * (size_t)s.cap;
}

__device__ __forceinline__ void load_state(tz_state* dst, const tz_state* src) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(src);
    uint32_t* b = reinterpret_cast<uint32_t*>(dst);
    for (int i = lane_id(); i < (int)(sizeof(tz_state) / 4); i += 64) b[i] = a[i];
}

__device__ __forceinline__ void write_default_node(const SearchDev& s, size_t i) {
    s.t.eval_tag[i] = TZ_EVAL_VALUE;
    s.t.eval_bits[i] = 0;
    s.t.visits[i] = 0;
    s.t.prob[i] = 0.0f;
    s.t.logit[i] = 0.0f;
    s.t.std_dev[i] = 0.0f;
    s.t.child0[i] = 0;
    s.t.nchild[i] = 0;
    s.t.action[i] = 0xFFFF;
}

// ---- wave reductions over (key, index) pairs -------------------------------------------------
// first minimum of Eval over children [c0, c0+nc), with one child's value overridden (the path
// child whose fresh value is still in registers).  Also reports whether all are known.
__device__ __forceinline__ void children_min_allknown(const SearchDev& s, size_t base, uint32_t c0, int nc,
                                                      uint32_t patch_node, Ev patch, Ev& min_out, bool& all_known) {
    const int l = lane_id();
    Ev best{0, 0};
    int best_i = 0x7fffffff;
    bool known = true;
    for (int i = l; i < nc; i += 64) {
        const uint32_t node = c0 + i;
        Ev e{s.t.eval_tag[base + node], s.t.eval_bits[base + node]};
        if (node == patch_node) e = patch;
        known = known && ev_known(e);
        if (best_i == 0x7fffffff || ev_cmp(e, best) < 0) {
            best = e;
            best_i = i;
        }
    }
    all_known = __all(known);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Ev o{(uint32_t)__shfl_xor((int)best.tag, d), (uint32_t)__shfl_xor((int)best.bits, d)};
        const int oi = __shfl_xor(best_i, d);
        if (oi != 0x7fffffff) {
            const int c = best_i == 0x7fffffff ? -1 : ev_cmp(o, best);
            if (c < 0 || (c == 0 && oi < best_i)) {
                best = o;
                best_i = oi;
            }
        }
    }
    min_out = best;
}

// mcts.rs:78-102 applied along the recorded path, leaf side first.  tnode/tvis are in LDS
// (node index and visit_count after this simulation's increment).  `stored` is the evaluation
// now stored in the node below the one being processed.
__device__ void backup(const SearchDev& s, size_t base, const uint32_t* tnode, const uint32_t* tvis, int leaf_level,
                       Ev child_eval, float child_var, Ev stored) {
    for (int lvl = leaf_level - 1; lvl >= 0; lvl--) {
        const uint32_t node = tnode[lvl], below = tnode[lvl + 1];
        Ev ev{s.t.eval_tag[base + node], s.t.eval_bits[base + node]};
        float sd = s.t.std_dev[base + node];
        const uint32_t c0 = s.t.child0[base + node];
        const int nc = s.t.nchild[base + node];
        // node_solver, mcts.rs:66-76
        Ev mn;
        bool all_known;
        children_min_allknown(s, base, c0, nc, below, stored, mn, all_known);
        if (child_eval.tag == TZ_EVAL_LOSS || all_known) {
            ev = ev_negate(mn);
            sd = 0.0f;
        }
        Ev up;
        float up_var;
        if (ev_known(ev)) {
            up = ev;
            up_var = sd * sd;
        } else {
            const float negated = ev_to_notnan(ev_negate(child_eval));
            const float n = (float)tvis[lvl];
            float mean = tz_bits_to_float(ev.bits);
            mean = mean + (-mean + negated) / n;
            ev = ev_value(mean);
            sd = sd + (-sd + sqrtf(child_var)) / n;
            up = ev_value(negated * TZ_DISCOUNT);
            up_var = child_var * TZ_DISCOUNT * TZ_DISCOUNT;
        }
        if (lane_id() == 0) {
            s.t.eval_tag[base + node] = (uint8_t)ev.tag;
            s.t.eval_bits[base + node] = ev.bits;
            s.t.std_dev[base + node] = sd;
        }
        stored = ev;
        child_eval = up;
        child_var = up_var;
    }
}

// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void descend_kernel(SearchDev s, int from_start) {
    constexpr int NN = N * N;
    const int g = blockIdx.x, l = lane_id();
    __shared__ tz_state env;
    __shared__ uint32_t tnode[TZ_MAX_DEPTH];
    __shared__ uint32_t tvis[TZ_MAX_DEPTH];
    __shared__ uint8_t reach[NN * 4];
    load_state(&env, &s.env[g]);
    const size_t base = slab_base(s, s.bank[g], g);
    uint32_t node = 0;
    float beta = s.betas[g];
    if (from_start) {
        node = (uint32_t)s.start_node[g];
        beta = 0.0f;  // batched.rs:279
    }
    __syncthreads();
    if (from_start && node != 0) {
        apply_move<N>(&env, s.t.action[base + node]);
        __syncthreads();
    }
    int depth = 0;
    int kind = -1;  // 0 known, 1 needs network, 2 error
    Ev known{0, 0};
    for (;;) {
        const uint32_t vis = s.t.visits[base + node] + 1;
        if (l == 0) {
            s.t.visits[base + node] = vis;
            tnode[depth] = node;
            tvis[depth] = vis;
        }
        depth++;
        const Ev ev{s.t.eval_tag[base + node], s.t.eval_bits[base + node]};
        const int nc = s.t.nchild[base + node];
        if (ev_known(ev) && ev.bits == 0) {  // is_terminal, node/mod.rs:106-108
            known = ev;
            kind = 0;
            break;
        }
        if (nc == 0 && !ev_known(ev)) {  // needs_initialization, node/mod.rs:83-85
            const int t = terminal<N>(&env);
            if (t != TZ_TERMINAL_NONE) {
                known = Ev{(uint32_t)(t == TZ_TERMINAL_WIN ? TZ_EVAL_WIN : t == TZ_TERMINAL_LOSS ? TZ_EVAL_LOSS : TZ_EVAL_DRAW), 0};
                if (l == 0) {
                    s.t.eval_tag[base + node] = (uint8_t)known.tag;
                    s.t.eval_bits[base + node] = 0;
                    s.t.std_dev[base + node] = 0.0f;
                }
                kind = 0;
            } else {
                kind = 1;
            }
            break;
        }
        if (depth >= TZ_MAX_DEPTH) {
            if (l == 0) atomicMax(s.error_flag, 2);
            kind = 2;
            break;
        }
        // select_with_puct, policy.rs:78-95 (ties -> last index: Iterator::max_by_key)
        const uint32_t c0 = s.t.child0[base + node];
        const float parent = (float)vis;
        const float er = tz_logf(((1.0f + parent) + 500.0f) / 500.0f) + 4.0f;  // policy.rs:143-145
        const float sq = sqrtf(parent);
        const bool parent_loss = ev.tag == TZ_EVAL_LOSS;
        float best_score = 0.0f;
        int best_i = -1;
        for (int i = l; i < nc; i += 64) {
            const size_t ci = base + c0 + i;
            const Ev ce{s.t.eval_tag[ci], s.t.eval_bits[ci]};
            if (!(parent_loss || ce.tag != TZ_EVAL_WIN)) continue;
            const float q = ev_to_notnan(ev_negate(ce));
            const float puct = ((er * s.t.prob[ci]) * sq) / (1.0f + (float)s.t.visits[ci]);
            const float score = (q + puct) + s.t.std_dev[ci] * beta;
            if (best_i < 0 || !(score < best_score)) {
                best_score = score;
                best_i = i;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float os = __shfl_xor(best_score, d);
            const int oi = __shfl_xor(best_i, d);
            if (oi >= 0 && (best_i < 0 || os > best_score || (os == best_score && oi > best_i))) {
                best_score = os;
                best_i = oi;
            }
        }
        if (best_i < 0) {  // reference: expect("there should always be a child to simulate")
            if (l == 0) atomicMax(s.error_flag, 4);
            kind = 2;
            break;
        }
        node = c0 + (uint32_t)best_i;
        apply_move<N>(&env, s.t.action[base + node]);
        __syncthreads();
    }
    __syncthreads();
    if (kind == 0) {
        // backward_known_eval, mcts.rs:141-163: the leaf itself is not updated
        backup(s, base, tnode, tvis, depth - 1, known, 0.0f, known);
        if (l == 0) {
            s.leaf_kind[g] = 0;
            s.traj_len[g] = 0;
        }
        return;
    }
    if (kind == 2) {
        if (l == 0) {
            s.leaf_kind[g] = 0;
            s.traj_len[g] = 0;
        }
        return;
    }
    // NeedsNetwork: env.populate_actions (batched.rs:84) + hand the leaf position to the net
    uint16_t* acts = s.leaf_act + (size_t)g * s.max_actions;
    const int nact = gen_moves<N>(&env, reach, acts, s.max_actions);
    if (nact > s.max_actions && l == 0) atomicMax(s.error_flag, 3);
    {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(&env);
        uint32_t* b = reinterpret_cast<uint32_t*>(&s.leaf_env[g]);
        for (int i = l; i < (int)(sizeof(tz_state) / 4); i += 64) b[i] = a[i];
    }
    for (int i = l; i < depth; i += 64) s.traj[(size_t)g * TZ_MAX_DEPTH + i] = tnode[i];
    if (l == 0) {
        s.leaf_kind[g] = 1;
        s.leaf_nact[g] = (uint16_t)(nact > s.max_actions ? s.max_actions : nact);
        s.traj_len[g] = (uint32_t)depth;
    }
}

// games whose leaf needs the network, in ascending game order (batched.rs:66-90 filter_map order)
__global__ __launch_bounds__(1024) void compact_kernel(SearchDev s) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int chunk = 0; chunk < s.batch; chunk += 1024) {
        const int g = chunk + tid;
        const int f = (g < s.batch && s.leaf_kind[g] == 1) ? 1 : 0;
        const int incl = wave_incl_scan(f);
        if (l == 63) wsum[w] = incl;
        __syncthreads();
        int off = carry;
        for (int i = 0; i < w; i++) off += wsum[i];
        if (f) s.nn_game[off + incl - 1] = g;
        __syncthreads();
        if (tid == 1023) carry = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        *s.nn_count = carry;
        s.counters[0] += (unsigned long long)s.batch;
        s.counters[1] += (unsigned long long)carry;
    }
}

// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void expand_kernel(SearchDev s, NetOut out) {
    constexpr int NN = N * N;
    const int slot = blockIdx.x, l = lane_id();
    if (slot >= *s.nn_count) return;
    const int g = s.nn_game[slot];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tnode = reinterpret_cast<uint32_t*>(smem);
    uint32_t* tvis = tnode + TZ_MAX_DEPTH;
    float* lg = reinterpret_cast<float*>(tvis + TZ_MAX_DEPTH);
    float* ex = lg + s.max_actions;
    const size_t base = slab_base(s, s.bank[g], g);
    const int len = (int)s.traj_len[g];
    for (int i = l; i < len; i += 64) {
        const uint32_t nd = s.traj[(size_t)g * TZ_MAX_DEPTH + i];
        tnode[i] = nd;
        tvis[i] = s.t.visits[base + nd];
    }
    const int nact = s.leaf_nact[g];
    const uint16_t* acts = s.leaf_act + (size_t)g * s.max_actions;
    // Agent::policy_value_uncertainty outputs for this leaf
    float value = 0.0f, variance = 0.0f;
    if (s.agent_kind == TZ_AGENT_NET) {
        value = out.value[slot];
        variance = out.variance[slot];
    } else if (s.agent_kind == TZ_AGENT_SIMPLE) {  // agent.rs:66-70
        __shared__ tz_state le;
        load_state(&le, &s.leaf_env[g]);
        __syncthreads();
        const int fd = flat_diff<N>(&le) - s.half_komi / 2;
        value = (float)fd / (float)NN;
        if (le.to_move == 1) value = -value;
    }
    bool nan = false;
    float mx = -3.4028235e38f;
    for (int i = l; i < nact; i += 64) {
        const int a = acts[i];
        float x;
        if (s.agent_kind == TZ_AGENT_NET) {
            x = out.policy[((size_t)slot * NN + (a % NN)) * out.policy_stride + a / NN];
        } else if (s.agent_kind == TZ_AGENT_DUMMY) {
            x = 1.0f;
        } else {
            const int ch = a / NN;
            x = ch == 0 ? 4.0f : ch == 1 ? 2.0f : ch == 2 ? 3.0f : 1.0f;  // agent.rs:73-80
        }
        nan = nan || !(x == x);
        lg[i] = x;
        mx = x > mx ? x : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(mx, d);
        mx = o > mx ? o : mx;
    }
    if (__any(nan) || !(value == value) || !(variance == variance)) {
        if (l == 0) atomicMax(s.error_flag, 7);
        return;
    }
    // softmax over the legal moves only, policy.rs:10-19 (sequential f32 sum)
    for (int i = l; i < nact; i += 64) ex[i] = tz_expf(lg[i] - mx);
    __syncthreads();
    float sum = 0.0f;
    for (int i = 0; i < nact; i++) sum = sum + ex[i];
    // leaf update, mcts.rs:190-196
    const uint32_t leaf = tnode[len - 1];
    const float n = (float)tvis[len - 1];
    float mean = tz_bits_to_float(s.t.eval_bits[base + leaf]);
    mean = mean + (-mean + value) / n;
    float sd = s.t.std_dev[base + leaf];
    sd = sd + (-sd + sqrtf(variance)) / n;
    const uint32_t c0 = s.alloc[g];
    // The reference's trees live on the heap and are unbounded; a game's pool here is not.  A full pool is either an
    // error (strict) or the leaf is evaluated and backed up without getting children: it stays a leaf, is simply
    // evaluated again when it is reached again, and the next `step` compacts the kept subtree and frees the rest.
    const bool full = c0 + (uint32_t)nact > (uint32_t)s.cap;
    if (full && s.strict_capacity) {
        if (l == 0) atomicMax(s.error_flag, 1);
        return;
    }
    // children, node/mod.rs:66-79
    const uint32_t child_bits = tz_float_to_bits(-mean);
    for (int i = l; i < (full ? 0 : nact); i += 64) {
        const size_t ci = base + c0 + i;
        s.t.eval_tag[ci] = TZ_EVAL_VALUE;
        s.t.eval_bits[ci] = child_bits;
        s.t.visits[ci] = 0;
        s.t.prob[ci] = ex[i] / sum;
        s.t.logit[ci] = lg[i];
        s.t.std_dev[ci] = sd;
        s.t.child0[ci] = 0;
        s.t.nchild[ci] = 0;
        s.t.action[ci] = acts[i];
    }
    if (l == 0) {
        s.t.eval_bits[base + leaf] = tz_float_to_bits(mean);
        s.t.std_dev[base + leaf] = sd;
        if (full) {
            atomicAdd(&s.counters[2], 1ull);
        } else {
            s.alloc[g] = c0 + (uint32_t)nact;
            s.t.child0[base + leaf] = c0;
            s.t.nchild[base + leaf] = (uint16_t)nact;
        }
    }
    __syncthreads();
    backup(s, base, tnode, tvis, len - 1, ev_value(value * TZ_DISCOUNT), variance * TZ_DISCOUNT * TZ_DISCOUNT, ev_value(mean));
}

// noise.rs:10-26 with the sample supplied by the caller
__global__ __launch_bounds__(64) void noise_kernel(SearchDev s, const float* noise, int amax, float ratio) {
    const int g = blockIdx.x, l = lane_id();
    const size_t base = slab_base(s, s.bank[g], g);
    const int nc = s.t.nchild[base];
    const bool known = s.t.eval_tag[base] != TZ_EVAL_VALUE;
    if (nc == 0 && !known) {
        if (l == 0) atomicMax(s.error_flag, 6);
        return;
    }
    const uint32_t c0 = s.t.child0[base];
    const float keep = 1.0f - ratio;
    for (int i = l; i < nc && i < amax; i += 64) {
        const size_t ci = base + c0 + i;
        const float p = s.t.prob[ci] * keep + noise[(size_t)g * amax + i] * ratio;
        s.t.prob[ci] = p;
        s.t.logit[ci] = tz_logf(p);
    }
}

template <int N>
__global__ __launch_bounds__(64) void root_info_kernel(SearchDev s, tz_root_info* out) {
    const int g = blockIdx.x, l = lane_id();
    __shared__ tz_state env;
    load_state(&env, &s.env[g]);
    __syncthreads();
    const int t = terminal<N>(&env);
    if (l != 0) return;
    const size_t base = slab_base(s, s.bank[g], g);
    tz_root_info r;
    r.visit_count = s.t.visits[base];
    r.n_children = s.t.nchild[base];
    r.eval_tag = s.t.eval_tag[base];
    r.is_terminal_env = t != TZ_TERMINAL_NONE;
    r.ply = env.ply;
    r.eval.ply = s.t.eval_bits[base];
    r.std_dev = s.t.std_dev[base];
    r.logit = s.t.logit[base];
    r.probability = s.t.prob[base];
    out[g] = r;
}

__global__ __launch_bounds__(64) void root_children_kernel(SearchDev s, int amax, uint16_t* move_idx, uint32_t* visits,
                                                           uint8_t* eval_tag, uint32_t* eval_bits, float* logit,
                                                           float* prob, float* std_dev) {
    const int g = blockIdx.x, l = lane_id();
    const size_t base = slab_base(s, s.bank[g], g);
    const int nc = s.t.nchild[base];
    const uint32_t c0 = s.t.child0[base];
    for (int i = l; i < amax; i += 64) {
        const size_t o = (size_t)g * amax + i;
        const bool in = i < nc;
        const size_t ci = base + c0 + i;
        if (move_idx) move_idx[o] = in ? s.t.action[ci] : 0;
        if (visits) visits[o] = in ? s.t.visits[ci] : 0;
        if (eval_tag) eval_tag[o] = in ? s.t.eval_tag[ci] : 0;
        if (eval_bits) eval_bits[o] = in ? s.t.eval_bits[ci] : 0;
        if (logit) logit[o] = in ? s.t.logit[ci] : 0.0f;
        if (prob) prob[o] = in ? s.t.prob[ci] : 0.0f;
        if (std_dev) std_dev[o] = in ? s.t.std_dev[ci] : 0.0f;
    }
}

// A node below the root (Node.children is a public field of the reference, node/mod.rs:14-23, read by puzzle / visualize_search):
// one wave walks `path` (move indices) from game g's root; status = -1 found, else the depth at which the path left the tree.
// out_words: [0] visits [1] nchild [2] tag [3] bits [4] std_dev [5] logit [6] prob; children rows as root_children_kernel.
__global__ __launch_bounds__(64) void node_kernel(SearchDev s, int g, const uint16_t* path, int len, uint32_t* out_words, int* status,
                                                  int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits,
                                                  float* logit, float* prob, float* std_dev) {
    const int l = lane_id();
    const size_t base = slab_base(s, s.bank[g], g);
    size_t node = base;
    for (int d = 0; d < len; d++) {
        const int nc = s.t.nchild[node];
        const uint32_t c0 = s.t.child0[node];
        int found = -1;
        for (int i0 = 0; i0 < nc && found < 0; i0 += 64) {
            const int i = i0 + l;
            const bool hit = i < nc && s.t.action[base + c0 + i] == path[d];
            const unsigned long long m = __ballot(hit);
            if (m) found = i0 + __ffsll((long long)m) - 1;
        }
        if (found < 0) {
            if (l == 0) *status = d;
            return;
        }
        node = base + c0 + found;
    }
    const int nc = s.t.nchild[node];
    const uint32_t c0 = s.t.child0[node];
    if (l == 0) {
        *status = -1;
        out_words[0] = s.t.visits[node];
        out_words[1] = (uint32_t)nc;
        out_words[2] = s.t.eval_tag[node];
        out_words[3] = s.t.eval_bits[node];
        out_words[4] = __float_as_uint(s.t.std_dev[node]);
        out_words[5] = __float_as_uint(s.t.logit[node]);
        out_words[6] = __float_as_uint(s.t.prob[node]);
    }
    for (int i = l; i < amax; i += 64) {
        const bool in = i < nc;
        const size_t ci = base + c0 + i;
        move_idx[i] = in ? s.t.action[ci] : 0;
        visits[i] = in ? s.t.visits[ci] : 0;
        eval_tag[i] = in ? s.t.eval_tag[ci] : 0;
        eval_bits[i] = in ? s.t.eval_bits[ci] : 0;
        logit[i] = in ? s.t.logit[ci] : 0.0f;
        prob[i] = in ? s.t.prob[ci] : 0.0f;
        std_dev[i] = in ? s.t.std_dev[ci] : 0.0f;
    }
}

// Node::select_best_action, node/mod.rs:132-161
__global__ __launch_bounds__(64) void select_best_kernel(SearchDev s, uint16_t* out) {
    const int g = blockIdx.x, l = lane_id();
    const size_t base = slab_base(s, s.bank[g], g);
    const int nc = s.t.nchild[base];
    const uint32_t c0 = s.t.child0[base];
    if (nc == 0) {
        if (l == 0) out[g] = 0xFFFF;
        return;
    }
    int pick;
    if (s.t.eval_tag[base] != TZ_EVAL_VALUE) {  // solved: first minimum child evaluation
        Ev best{0, 0};
        int best_i = 0x7fffffff;
        for (int i = l; i < nc; i += 64) {
            const Ev e{s.t.eval_tag[base + c0 + i], s.t.eval_bits[base + c0 + i]};
            if (best_i == 0x7fffffff || ev_cmp(e, best) < 0) {
                best = e;
                best_i = i;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const Ev o{(uint32_t)__shfl_xor((int)best.tag, d), (uint32_t)__shfl_xor((int)best.bits, d)};
            const int oi = __shfl_xor(best_i, d);
            if (oi != 0x7fffffff) {
                const int c = best_i == 0x7fffffff ? -1 : ev_cmp(o, best);
                if (c < 0 || (c == 0 && oi < best_i)) {
                    best = o;
                    best_i = oi;
                }
            }
        }
        pick = best_i;
    } else {  // most visited (last max); all zero -> highest prior (last max)
        uint32_t bv = 0;
        int bvi = -1;
        float bp = 0.0f;
        int bpi = -1;
        for (int i = l; i < nc; i += 64) {
            const uint32_t v = s.t.visits[base + c0 + i];
            const float p = s.t.prob[base + c0 + i];
            if (bvi < 0 || v >= bv) {
                bv = v;
                bvi = i;
            }
            if (bpi < 0 || !(p < bp)) {
                bp = p;
                bpi = i;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t ov = (uint32_t)__shfl_xor((int)bv, d);
            const int ovi = __shfl_xor(bvi, d);
            if (ovi >= 0 && (bvi < 0 || ov > bv || (ov == bv && ovi > bvi))) {
                bv = ov;
                bvi = ovi;
            }
            const float op = __shfl_xor(bp, d);
            const int opi = __shfl_xor(bpi, d);
            if (opi >= 0 && (bpi < 0 || op > bp || (op == bp && opi > bpi))) {
                bp = op;
                bpi = opi;
            }
        }
        pick = bv == 0 ? bpi : bvi;
    }
    if (l == 0) out[g] = s.t.action[base + c0 + pick];
}

__device__ __forceinline__ void copy_node(const SearchDev& s, size_t dst, size_t src) {
    s.t.eval_tag[dst] = s.t.eval_tag[src];
    s.t.eval_bits[dst] = s.t.eval_bits[src];
    s.t.visits[dst] = s.t.visits[src];
    s.t.prob[dst] = s.t.prob[src];
    s.t.logit[dst] = s.t.logit[src];
    s.t.std_dev[dst] = s.t.std_dev[src];
    s.t.nchild[dst] = s.t.nchild[src];
    s.t.action[dst] = s.t.action[src];
    s.t.child0[dst] = 0;
}

// BatchedMCTS::step, batched.rs:131-144: descend (subtree reuse) + env.step, skipped for terminal
// roots.  The kept subtree is copied breadth-first into the game's other bank.
template <int N>
__global__ __launch_bounds__(64) void step_kernel(SearchDev s, const uint16_t* actions) {
    const int g = blockIdx.x, l = lane_id();
    __shared__ tz_state env;
    const int sbank = s.bank[g], dbank = 1 - sbank;
    const size_t sb = slab_base(s, sbank, g), db = slab_base(s, dbank, g);
    if (s.t.eval_tag[sb] != TZ_EVAL_VALUE && s.t.eval_bits[sb] == 0) return;  // root.is_terminal()
    const int action = actions[g];
    const int nc = s.t.nchild[sb];
    const uint32_t c0 = s.t.child0[sb];
    int found = -1;
    for (int i = l; i < nc; i += 64)
        if (s.t.action[sb + c0 + i] == action) found = i;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(found, d);
        found = o > found ? o : found;
    }
    uint32_t* bsrc = s.bfs_src + (size_t)g * s.cap;
    uint32_t count = 1;
    if (found < 0) {
        if (l == 0) write_default_node(s, db);
    } else {
        if (l == 0) {
            copy_node(s, db, sb + c0 + found);
            bsrc[0] = c0 + (uint32_t)found;
        }
        __threadfence_block();
        __syncthreads();
        uint32_t head = 0;
        bool overflow = false;
        while (head < count && !overflow) {
            const uint32_t i = head + l;
            const bool have = i < count;
            const uint32_t src = have ? bsrc[i] : 0;
            const int cn = have ? (int)s.t.nchild[sb + src] : 0;
            const uint32_t cc0 = have ? s.t.child0[sb + src] : 0;
            const int incl = wave_incl_scan(cn);
            const uint32_t off = count + (uint32_t)(incl - cn);
            const uint32_t total = (uint32_t)__shfl(incl, 63);
            if (count + total > (uint32_t)s.cap) {
                overflow = true;
                break;
            }
            if (have && cn) s.t.child0[db + i] = off;
            unsigned long long m = __ballot(cn > 0);
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t jc0 = (uint32_t)__shfl((int)cc0, j);
                const int jn = __shfl(cn, j);
                const uint32_t joff = (uint32_t)__shfl((int)off, j);
                for (int k = l; k < jn; k += 64) {
                    copy_node(s, db + joff + k, sb + jc0 + k);
                    bsrc[joff + k] = jc0 + k;
                }
            }
            __threadfence_block();
            __syncthreads();
            const uint32_t processed = count - head < 64 ? count - head : 64;
            head += processed;
            count += total;
        }
        if (overflow && l == 0) atomicMax(s.error_flag, 1);
    }
    load_state(&env, &s.env[g]);
    __syncthreads();
    apply_move<N>(&env, action);
    __syncthreads();
    {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(&env);
        uint32_t* b = reinterpret_cast<uint32_t*>(&s.env[g]);
        for (int i = l; i < (int)(sizeof(tz_state) / 4); i += 64) b[i] = a[i];
    }
    if (l == 0) {
        s.bank[g] = (uint8_t)dbank;
        s.alloc[g] = count;
    }
}

// restart_terminal_envs, batched.rs:185-203 (and BatchedMCTS::new when force_all)
template <int N>
__global__ __launch_bounds__(64) void restart_kernel(SearchDev s, const int32_t* choice, int8_t* terminal_out, int force_all,
                                                     int with_moves) {
    const int g = blockIdx.x, l = lane_id();
    __shared__ tz_state env;
    load_state(&env, &s.env[g]);
    __syncthreads();
    int why = 0;
    const int t = force_all ? TZ_TERMINAL_NONE : terminal<N>(&env, &why);
    if (l != 0) return;
    if (terminal_out) terminal_out[g] = (int8_t)t;
    s.term_reason[g] = (int8_t)why;
    // winner colour: the terminal is from the side to move
    s.term_winner[g] = (uint8_t)(t == TZ_TERMINAL_DRAW ? 2 : t == TZ_TERMINAL_WIN ? env.to_move : 1 - env.to_move);
    if (t == TZ_TERMINAL_NONE && !force_all) return;
    write_opening<N>(&s.env[g], s.half_komi, choice ? choice[g] : 0, with_moves != 0);
    write_default_node(s, slab_base(s, s.bank[g], g));
    s.alloc[g] = 1;
}

// Validated replay step (Replay::from_str re-validates every move, target.rs:248-268): applies actions[g] to
// game g's position iff it is one of its legal moves; ok[g] = 1 applied, 0 illegal (position unchanged),
// -1 position already terminal.  Trees are reset.  0xFFFF = no move for this game.
template <int N>
__global__ __launch_bounds__(64) void play_moves_kernel(SearchDev s, const uint16_t* actions, int8_t* ok) {
    constexpr int NN = N * N;
    const int g = blockIdx.x, l = lane_id();
    __shared__ tz_state env;
    __shared__ uint8_t reach[NN * 4];
    __shared__ uint16_t acts[1024];
    const int a = actions[g];
    if (a == 0xFFFF) {
        if (l == 0) ok[g] = 0;
        return;
    }
    load_state(&env, &s.env[g]);
    __syncthreads();
    if (terminal<N>(&env) != TZ_TERMINAL_NONE) {
        if (l == 0) ok[g] = -1;
        return;
    }
    const int n = gen_moves<N>(&env, reach, acts, 1024);
    bool hit = false;
    for (int i = l; i < n && i < 1024; i += 64) hit = hit || acts[i] == a;
    if (!__any(hit)) {
        if (l == 0) ok[g] = 0;
        return;
    }
    apply_move<N>(&env, a);
    __syncthreads();
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&env);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&s.env[g]);
        for (int i = l; i < (int)(sizeof(tz_state) / 4); i += 64) dst[i] = src[i];
    }
    if (l == 0) {
        ok[g] = 1;
        write_default_node(s, slab_base(s, s.bank[g], g));
        s.alloc[g] = 1;
    }
}

__global__ void reset_games_kernel(SearchDev s, const int32_t* idx, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int g = idx[i];
    write_default_node(s, slab_base(s, s.bank[g], g));
    s.alloc[g] = 1;
}

// gumbel: start_node[g] = index of the chosen root child (child_index modulo handled by host)
__global__ void set_start_kernel(SearchDev s, const int32_t* child_index) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= s.batch) return;
    const size_t base = slab_base(s, s.bank[g], g);
    s.start_node[g] = (int32_t)(s.t.child0[base] + (uint32_t)child_index[g]);
}

// recompute root statistics after sequential halving, batched.rs:373-406
__global__ __launch_bounds__(64) void gumbel_fixup_kernel(SearchDev s) {
    const int g = blockIdx.x, l = lane_id();
    const size_t base = slab_base(s, s.bank[g], g);
    const int nc = s.t.nchild[base];
    const uint32_t c0 = s.t.child0[base];
    if (nc == 0) return;
    uint32_t sum = 0;
    bool any_loss = false;
    for (int i = l; i < nc; i += 64) {
        sum += s.t.visits[base + c0 + i];
        any_loss = any_loss || s.t.eval_tag[base + c0 + i] == TZ_EVAL_LOSS;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
    Ev mn;
    bool all_known;
    children_min_allknown(s, base, c0, nc, 0xFFFFFFFFu, Ev{0, 0}, mn, all_known);
    Ev ev;
    if (__any(any_loss) || all_known) {
        ev = ev_negate(mn);
    } else {
        float sp = 0.0f, wq = 0.0f;  // sequential sums in child order
        for (int i = 0; i < nc; i++)
            if (s.t.visits[base + c0 + i] > 0) sp = sp + s.t.prob[base + c0 + i];
        for (int i = 0; i < nc; i++)
            if (s.t.visits[base + c0 + i] > 0) {
                const Ev ce{s.t.eval_tag[base + c0 + i], s.t.eval_bits[base + c0 + i]};
                wq = wq + s.t.prob[base + c0 + i] * ev_to_f32(ev_negate(ce));
            }
        ev = ev_value(wq / sp);
    }
    if (l == 0) {
        s.t.visits[base] = sum + 1;
        s.t.eval_tag[base] = (uint8_t)ev.tag;
        s.t.eval_bits[base] = ev.bits;
        if (ev_known(ev)) s.t.std_dev[base] = 0.0f;
    }
}

}  // namespace

#define TZ_DISPATCH_N(n, CALL)                     \
    switch (n) {                                   \
        case 3: { constexpr int NB = 3; CALL; } break; \
        case 4: { constexpr int NB = 4; CALL; } break; \
        case 5: { constexpr int NB = 5; CALL; } break; \
        case 6: { constexpr int NB = 6; CALL; } break; \
        default: return tz_fail(TZ_EINVAL, "unsupported board size"); \
    }

// diagnostic: the f32 primitives the tree kernels rely on being bit-identical with the host
__global__ void device_math_kernel(int op, const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op) {
        case 0: out[i] = tz_expf(a[i]); break;
        case 1: out[i] = tz_logf(a[i]); break;
        case 2: out[i] = sqrtf(a[i]); break;
        case 3: out[i] = a[i] / b[i]; break;
        case 4: out[i] = tz_powif(TZ_DISCOUNT, (int)a[i]); break;
        default: out[i] = (a[i] + b[i]) * a[i]; break;
    }
}

#define TZ_LAUNCH_CHECK()                                                                       \
    do {                                                                                        \
        hipError_t _e = hipGetLastError();                                                      \
        if (_e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("kernel launch: ") + hipGetErrorString(_e)); \
    } while (0)

extern "C" int tz_device_math(int op, const float* a, const float* b, float* out, int n) {
    if (!a || !b || !out || n <= 0) return tz_fail(TZ_EINVAL, "tz_device_math: bad argument");
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    TZ_HIP(hipMalloc(&da, n * 4));
    TZ_HIP(hipMalloc(&db, n * 4));
    TZ_HIP(hipMalloc(&dout, n * 4));
    TZ_HIP(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    TZ_HIP(hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice));
    device_math_kernel<<<(n + 255) / 256, 256>>>(op, da, db, dout, n);
    TZ_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return TZ_OK;
}

int tz_tree_descend(const SearchDev& s, bool from_start_nodes, hipStream_t st) {
    TZ_DISPATCH_N(s.n, (descend_kernel<NB><<<s.batch, 64, 0, st>>>(s, from_start_nodes ? 1 : 0)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_compact_leaves(const SearchDev& s, hipStream_t st) {
    compact_kernel<<<1, 1024, 0, st>>>(s);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_expand(const SearchDev& s, const NetOut& out, hipStream_t st) {
    const size_t smem = 2 * TZ_MAX_DEPTH * sizeof(uint32_t) + 2 * (size_t)s.max_actions * sizeof(float);
    TZ_DISPATCH_N(s.n, (expand_kernel<NB><<<s.batch, 64, smem, st>>>(s, out)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_apply_noise(const SearchDev& s, const float* noise_dev, int amax, float ratio, hipStream_t st) {
    noise_kernel<<<s.batch, 64, 0, st>>>(s, noise_dev, amax, ratio);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_root_info(const SearchDev& s, tz_root_info* out_dev, hipStream_t st) {
    TZ_DISPATCH_N(s.n, (root_info_kernel<NB><<<s.batch, 64, 0, st>>>(s, out_dev)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_root_children(const SearchDev& s, int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag,
                          uint32_t* eval_bits, float* logit, float* prob, float* std_dev, hipStream_t st) {
    root_children_kernel<<<s.batch, 64, 0, st>>>(s, amax, move_idx, visits, eval_tag, eval_bits, logit, prob, std_dev);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_node(const SearchDev& s, int game, const uint16_t* path_dev, int len, uint32_t* out_words, int* status, int amax,
                 uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob, float* std_dev,
                 hipStream_t st) {
    node_kernel<<<1, 64, 0, st>>>(s, game, path_dev, len, out_words, status, amax, move_idx, visits, eval_tag, eval_bits, logit, prob, std_dev);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_select_best(const SearchDev& s, uint16_t* out_dev, hipStream_t st) {
    select_best_kernel<<<s.batch, 64, 0, st>>>(s, out_dev);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_step(const SearchDev& s, const uint16_t* actions_dev, hipStream_t st) {
    TZ_DISPATCH_N(s.n, (step_kernel<NB><<<s.batch, 64, 0, st>>>(s, actions_dev)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_restart(const SearchDev& s, const int32_t* choice_dev, int8_t* terminal_dev, bool force_all,
                    bool with_opening_moves, hipStream_t st) {
    TZ_DISPATCH_N(s.n, (restart_kernel<NB><<<s.batch, 64, 0, st>>>(s, choice_dev, terminal_dev, force_all ? 1 : 0,
                                                                  with_opening_moves ? 1 : 0)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_play_moves(const SearchDev& s, const uint16_t* actions_dev, int8_t* ok_dev, hipStream_t st) {
    TZ_DISPATCH_N(s.n, (play_moves_kernel<NB><<<s.batch, 64, 0, st>>>(s, actions_dev, ok_dev)));
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_reset_games(const SearchDev& s, const int32_t* idx_dev, int count, hipStream_t st) {
    if (count <= 0) return TZ_OK;
    reset_games_kernel<<<(count + 255) / 256, 256, 0, st>>>(s, idx_dev, count);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_set_start_children(const SearchDev& s, const int32_t* child_index_dev, hipStream_t st) {
    set_start_kernel<<<(s.batch + 255) / 256, 256, 0, st>>>(s, child_index_dev);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
int tz_tree_gumbel_root_fixup(const SearchDev& s, hipStream_t st) {
    gumbel_fixup_kernel<<<s.batch, 64, 0, st>>>(s);
    TZ_LAUNCH_CHECK();
    return TZ_OK;
}
