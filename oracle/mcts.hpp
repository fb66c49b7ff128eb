// oracle/mcts.hpp — CPU restatement of takzero's tree search
//   takzero/src/search/eval.rs, search/node/{mod,mcts,policy,noise,batched}.rs, search/agent.rs
//
// TEST INFRASTRUCTURE ONLY (see oracle/tak.hpp).  Single threaded on purpose: the reference is
// (batched.rs:21-22) and the order of f32 operations below is the order the reference performs
// them in.  Compile with -ffp-contract=off.
//
// exp / ln / powi come from takzero_amd/csrc/tz_math.h (see the note there); set
// tzo::g_use_libm = true to use the host libm instead, as the Rust binary does.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "../takzero_amd/csrc/tz_math.h"
#include "tak.hpp"

namespace tzo {

static bool g_use_libm = false;
static inline float m_exp(float x) { return g_use_libm ? expf(x) : tz_expf(x); }
static inline float m_ln(float x) { return g_use_libm ? logf(x) : tz_logf(x); }

// ------------------------------------------------------------------ eval.rs
struct Eval {
    uint8_t tag = TZ_EVAL_VALUE;
    float value = 0.0f;  // Value
    uint32_t ply = 0;    // Win / Loss / Draw
    static Eval Value(float v) {
        Eval e;
        e.value = v;
        return e;
    }
    static Eval Known(uint8_t tag, uint32_t ply) {
        Eval e;
        e.tag = tag;
        e.ply = ply;
        return e;
    }
    bool is_known() const { return tag != TZ_EVAL_VALUE; }
    bool is_win() const { return tag == TZ_EVAL_WIN; }
    bool is_loss() const { return tag == TZ_EVAL_LOSS; }
    bool is_draw() const { return tag == TZ_EVAL_DRAW; }
    // eval.rs:40-47
    Eval negate() const {
        switch (tag) {
            case TZ_EVAL_VALUE: return Value(-value);
            case TZ_EVAL_WIN: return Known(TZ_EVAL_LOSS, ply + 1);
            case TZ_EVAL_DRAW: return Known(TZ_EVAL_DRAW, ply + 1);
            default: return Known(TZ_EVAL_WIN, ply + 1);
        }
    }
    // impl From<Eval> for f32, eval.rs:95-105
    float to_f32() const {
        float base = tz_powif(TZ_DISCOUNT, is_known() ? (int)ply : 0);
        float x = tag == TZ_EVAL_VALUE ? value : tag == TZ_EVAL_WIN ? 1.0f : tag == TZ_EVAL_LOSS ? -1.0f : 0.0f;
        return base * x;
    }
    // impl From<Eval> for NotNan<f32>, eval.rs:107-116
    float to_notnan() const { return tag == TZ_EVAL_VALUE ? value : to_f32(); }
    uint32_t bits() const { return tag == TZ_EVAL_VALUE ? tz_float_to_bits(value) : ply; }
};

static constexpr float CONTEMPT = -0.05f;  // eval.rs:128

static inline int cmp_f32(float a, float b) { return a < b ? -1 : a > b ? 1 : 0; }
static inline int cmp_u32(uint32_t a, uint32_t b) { return a < b ? -1 : a > b ? 1 : 0; }
// impl Ord for Eval, eval.rs:138-163
static inline int eval_cmp(const Eval& a, const Eval& b) {
    switch (a.tag) {
        case TZ_EVAL_VALUE:
            switch (b.tag) {
                case TZ_EVAL_VALUE: return cmp_f32(a.value, b.value);
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmp_f32(a.value, CONTEMPT);
                default: return 1;
            }
        case TZ_EVAL_WIN: return b.tag == TZ_EVAL_WIN ? cmp_u32(b.ply, a.ply) : 1;
        case TZ_EVAL_DRAW:
            switch (b.tag) {
                case TZ_EVAL_VALUE: return cmp_f32(CONTEMPT, b.value);
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmp_u32(b.ply, a.ply);
                default: return 1;
            }
        default: return b.tag == TZ_EVAL_LOSS ? cmp_u32(a.ply, b.ply) : -1;
    }
}

static inline Eval eval_from_terminal(int t) {
    return Eval::Known(t == TZ_TERMINAL_WIN ? TZ_EVAL_WIN : t == TZ_TERMINAL_LOSS ? TZ_EVAL_LOSS : TZ_EVAL_DRAW, 0);
}

// ------------------------------------------------------------------ policy.rs:10-19
static inline void softmax(const std::vector<float>& logits, std::vector<float>& out) {
    out.resize(logits.size());
    float mx = 0.0f;
    if (!logits.empty()) {
        mx = logits[0];
        for (float x : logits)
            if (x > mx) mx = x;
    }
    float sum = 0.0f;
    for (size_t i = 0; i < logits.size(); i++) {
        out[i] = m_exp(logits[i] - mx);
        sum = sum + out[i];
    }
    for (size_t i = 0; i < logits.size(); i++) out[i] = out[i] / sum;
}

// policy.rs:140-156
static inline float exploration_rate(float visit_count) {
    return m_ln(((1.0f + visit_count) + 500.0f) / 500.0f) + 4.0f;
}
static inline float ucb_with_predictor(float parent_visits, float visits, float prob) {
    return ((exploration_rate(parent_visits) * prob) * sqrtf(parent_visits)) / (1.0f + visits);
}
// policy.rs:121-138
static inline float sigma_select(float q, float std_dev, float beta, float visit_count) {
    return (q + std_dev * beta) * (50.0f + visit_count);
}
static inline float sigma_improve(float q, float std_dev, float beta, float visit_count) {
    return (q + std_dev * beta) * sqrtf(visit_count);
}

// ------------------------------------------------------------------ agent.rs
template <class E>
struct Agent {
    virtual ~Agent() {}
    // one output per input, logits for exactly the given actions in the given order
    virtual void policy_value_uncertainty(const std::vector<E>& envs,
                                          const std::vector<std::vector<int>>& actions,
                                          std::vector<std::vector<float>>& logits,
                                          std::vector<float>& value, std::vector<float>& variance) = 0;
};

// ------------------------------------------------------------------ node/mod.rs:14-23
template <class E>
struct Node {
    Eval evaluation;
    uint32_t visit_count = 0;
    float logit = 0.0f;
    float probability = 0.0f;
    float std_dev = 0.0f;
    std::vector<std::pair<int, Node>> children;

    bool needs_initialization() const { return children.empty() && !evaluation.is_known(); }  // mod.rs:83-85
    bool is_terminal() const { return evaluation.is_known() && evaluation.ply == 0; }         // mod.rs:106-108
    float q_value() const { return evaluation.negate().to_notnan(); }                          // mod.rs:114-124

    // mod.rs:95-102
    void descend(int action) {
        Node me = std::move(*this);
        *this = Node();
        for (auto& c : me.children)
            if (c.first == action) {
                Node child = std::move(c.second);
                *this = std::move(child);
                return;
            }
    }

    // mcts.rs:49-61
    void update_mean_value(float value) {
        if (evaluation.tag == TZ_EVAL_VALUE)
            evaluation.value = evaluation.value + (-evaluation.value + value) / (float)visit_count;
    }
    void update_standard_deviation(float variance) {
        if (evaluation.is_known()) return;
        std_dev = std_dev + (-std_dev + sqrtf(variance)) / (float)visit_count;
    }
    Eval min_child_eval() const {  // Iterator::min -> first minimum
        Eval best = children[0].second.evaluation;
        for (size_t i = 1; i < children.size(); i++)
            if (eval_cmp(children[i].second.evaluation, best) < 0) best = children[i].second.evaluation;
        return best;
    }
    // mcts.rs:66-76
    void node_solver(const Eval& child_eval) {
        bool all_known = true;
        for (auto& c : children)
            if (!c.second.evaluation.is_known()) {
                all_known = false;
                break;
            }
        if (child_eval.is_loss() || all_known) {
            evaluation = min_child_eval().negate();
            std_dev = 0.0f;
        }
    }
    struct Propagated {
        Eval eval;
        float variance;
    };
    // mcts.rs:78-102
    Propagated propagate_child_eval(const Eval& child_eval, float child_variance) {
        node_solver(child_eval);
        if (evaluation.is_known()) return {evaluation, std_dev * std_dev};
        float negated = child_eval.negate().to_notnan();
        update_mean_value(negated);
        update_standard_deviation(child_variance);
        return {Eval::Value(negated * TZ_DISCOUNT), child_variance * TZ_DISCOUNT * TZ_DISCOUNT};
    }

    // policy.rs:78-95
    size_t select_with_puct(float beta) const {
        float parent = (float)visit_count;
        bool have = false;
        size_t best = 0;
        float best_score = 0.0f;
        for (size_t i = 0; i < children.size(); i++) {
            const Node& ch = children[i].second;
            if (!(evaluation.is_loss() || !ch.evaluation.is_win())) continue;
            float q = ch.q_value();
            float puct = ucb_with_predictor(parent, (float)ch.visit_count, ch.probability);
            float score = (q + puct) + ch.std_dev * beta;
            if (!have || !(score < best_score)) {  // max_by_key keeps the last maximum
                have = true;
                best = i;
                best_score = score;
            }
        }
        return best;  // reference panics when no child is eligible
    }

    enum ForwardKind { KNOWN, NEEDS_NETWORK };
    // mcts.rs:107-138
    ForwardKind forward(std::vector<size_t>& trajectory, E& env, float beta, Eval& known_out) {
        Node* node = this;
        for (;;) {
            node->visit_count += 1;
            if (node->is_terminal()) {
                known_out = node->evaluation;
                return KNOWN;
            }
            if (node->needs_initialization()) {
                int t = env.terminal();
                if (t != TZ_TERMINAL_NONE) {
                    node->evaluation = eval_from_terminal(t);
                    node->std_dev = 0.0f;
                    known_out = node->evaluation;
                    return KNOWN;
                }
                return NEEDS_NETWORK;
            }
            size_t index = node->select_with_puct(beta);
            trajectory.push_back(index);
            env.step(node->children[index].first);
            node = &node->children[index].second;
        }
    }
    // mcts.rs:141-163
    Propagated backward_known_eval(const std::vector<size_t>& traj, size_t depth, const Eval& eval) {
        if (depth < traj.size()) {
            Propagated p = children[traj[depth]].second.backward_known_eval(traj, depth + 1, eval);
            return propagate_child_eval(p.eval, p.variance);
        }
        return {eval, 0.0f};
    }
    // mcts.rs:171-225
    Propagated backward_network_eval(const std::vector<size_t>& traj, size_t depth,
                                     const std::vector<int>& actions, const std::vector<float>& logits,
                                     const std::vector<float>& probs, float value, float variance) {
        if (depth < traj.size()) {
            Propagated p = children[traj[depth]].second.backward_network_eval(traj, depth + 1, actions,
                                                                              logits, probs, value, variance);
            return propagate_child_eval(p.eval, p.variance);
        }
        update_mean_value(value);
        update_standard_deviation(variance);
        children.clear();
        children.reserve(actions.size());
        float parent_eval = evaluation.to_notnan();
        for (size_t i = 0; i < actions.size(); i++) {
            Node c;  // from_logit_and_probability_and_parent_value_and_std_dev, mod.rs:66-79
            c.evaluation = Eval::Value(-parent_eval);
            c.logit = logits[i];
            c.probability = probs[i];
            c.std_dev = std_dev;
            children.emplace_back(actions[i], std::move(c));
        }
        return {Eval::Value(value * TZ_DISCOUNT), variance * TZ_DISCOUNT * TZ_DISCOUNT};
    }

    // mcts.rs:235-268
    Propagated simulate_simple(Agent<E>& agent, E env, float beta) {
        std::vector<size_t> traj;
        Eval known;
        if (forward(traj, env, beta, known) == KNOWN) return backward_known_eval(traj, 0, known);
        std::vector<std::vector<int>> actions(1);
        env.populate_actions(actions[0]);
        std::vector<std::vector<float>> logits;
        std::vector<float> value, variance, probs;
        std::vector<E> envs{env};
        agent.policy_value_uncertainty(envs, actions, logits, value, variance);
        softmax(logits[0], probs);
        return backward_network_eval(traj, 0, actions[0], logits[0], probs, value[0], variance[0]);
    }

    // policy.rs:23-48
    uint32_t most_visited_count() const {
        uint32_t m = 0;
        for (auto& c : children) m = std::max(m, c.second.visit_count);
        return m;
    }
    void improved_policy(float visitations, std::vector<float>& out) const {
        std::vector<float> p(children.size());
        for (size_t i = 0; i < children.size(); i++) {
            const Node& n = children[i].second;
            float completed = (n.needs_initialization() ? evaluation : n.evaluation.negate()).to_notnan();
            p[i] = sigma_improve(completed, n.std_dev, 0.0f, visitations) + n.logit;
        }
        softmax(p, out);
    }

    // mod.rs:132-161
    int select_best_action() const {
        if (evaluation.is_known()) {
            size_t best = 0;
            for (size_t i = 1; i < children.size(); i++)
                if (eval_cmp(children[i].second.evaluation, children[best].second.evaluation) < 0) best = i;
            return children[best].first;
        }
        size_t mv = 0;
        for (size_t i = 1; i < children.size(); i++)
            if (children[i].second.visit_count >= children[mv].second.visit_count) mv = i;  // last max
        if (children[mv].second.visit_count == 0) {
            size_t bp = 0;
            for (size_t i = 1; i < children.size(); i++)
                if (!(children[i].second.probability < children[bp].second.probability)) bp = i;  // last max
            return children[bp].first;
        }
        return children[mv].first;
    }
    // mod.rs:170-207: weights handed to choose_weighted (all zero => fall back to best action);
    // returns false when the reference would not sample at all.
    bool selfplay_weights(bool sample, uint32_t threshold, float allowed_drop, std::vector<uint32_t>& w) const {
        w.assign(children.size(), 0);
        if (evaluation.is_known() || !sample) return false;
        Eval best = min_child_eval();
        Eval limit = best.tag == TZ_EVAL_VALUE ? Eval::Value(best.value + allowed_drop) : best;
        bool any = false;
        for (size_t i = 0; i < children.size(); i++) {
            const Node& c = children[i].second;
            if (c.visit_count < threshold || c.evaluation.is_win() || eval_cmp(c.evaluation, limit) > 0) continue;
            w[i] = c.visit_count;
            any = any || c.visit_count > 0;
        }
        return any;
    }
    // mod.rs:215-230
    float ube_target(float beta) const {
        if (evaluation.is_known() || needs_initialization()) return 0.0f;
        size_t best = 0;
        float best_key = 0.0f;
        for (size_t i = 0; i < children.size(); i++) {
            const Node& c = children[i].second;
            float key = c.evaluation.negate().to_notnan() + c.std_dev * beta;
            if (i == 0 || !(key < best_key)) {
                best = i;
                best_key = key;
            }
        }
        float s = children[best].second.std_dev;
        return s * s;
    }
    // noise.rs:10-26 with the Dirichlet sample supplied by the caller
    bool apply_dirichlet(const float* noise, float ratio) {
        if (needs_initialization()) return false;
        for (size_t i = 0; i < children.size(); i++) {
            Node& c = children[i].second;
            c.probability = c.probability * (1.0f - ratio) + noise[i] * ratio;
            c.logit = m_ln(c.probability);
        }
        return true;
    }
};

// ------------------------------------------------------------------ batched.rs
template <class E>
struct BatchedMCTS {
    std::vector<Node<E>> nodes;
    std::vector<E> envs;
    std::vector<std::vector<int>> replays;  // actions only; start env kept by the caller
    uint64_t simulations = 0, nn_leaf_evals = 0;

    explicit BatchedMCTS(const std::vector<E>& e) : nodes(e.size()), envs(e), replays(e.size()) {}
    size_t batch() const { return envs.size(); }

    struct Pending {
        Node<E>* node;
        std::vector<size_t> traj;
    };
    // shared body of simulate() (batched.rs:63-128) and of the inner loop of
    // gumbel_sequential_halving (batched.rs:265-335)
    void simulate_from(std::vector<Node<E>*>& roots, const std::vector<E>& root_envs, Agent<E>& agent,
                       const std::vector<float>& betas) {
        std::vector<Pending> pend;
        std::vector<E> env_batch;
        std::vector<std::vector<int>> act_batch;
        for (size_t g = 0; g < roots.size(); g++) {
            std::vector<size_t> traj;
            E env = root_envs[g];
            Eval known;
            simulations++;
            if (roots[g]->forward(traj, env, betas[g], known) == Node<E>::KNOWN) {
                roots[g]->backward_known_eval(traj, 0, known);
            } else {
                std::vector<int> acts;
                env.populate_actions(acts);
                env_batch.push_back(env);
                act_batch.push_back(std::move(acts));
                pend.push_back({roots[g], std::move(traj)});
            }
        }
        if (env_batch.empty()) return;
        std::vector<std::vector<float>> logits;
        std::vector<float> value, variance, probs;
        agent.policy_value_uncertainty(env_batch, act_batch, logits, value, variance);
        nn_leaf_evals += env_batch.size();
        for (size_t i = 0; i < pend.size(); i++) {
            softmax(logits[i], probs);
            pend[i].node->backward_network_eval(pend[i].traj, 0, act_batch[i], logits[i], probs, value[i], variance[i]);
        }
    }
    void simulate(Agent<E>& agent, const std::vector<float>& betas) {
        std::vector<Node<E>*> roots(batch());
        for (size_t g = 0; g < batch(); g++) roots[g] = &nodes[g];
        simulate_from(roots, envs, agent, betas);
    }
    // batched.rs:131-144
    void step(const std::vector<int>& actions) {
        for (size_t g = 0; g < batch(); g++) {
            if (nodes[g].is_terminal()) continue;
            nodes[g].descend(actions[g]);
            replays[g].push_back(actions[g]);
            envs[g].step(actions[g]);
        }
    }
    // batched.rs:207-409 with the Gumbel(0,1) samples supplied by the caller:
    // gumbel[g][i] for child i of root g.
    void gumbel_sequential_halving(Agent<E>& agent, const std::vector<float>& betas, size_t sampled_actions,
                                   uint32_t search_budget, const std::vector<std::vector<float>>& gumbel,
                                   std::vector<int>& selected) {
        uint32_t lg = 31 - __builtin_clz((unsigned)sampled_actions);
        simulate(agent, betas);
        struct Cand {
            float key;
            size_t child;
        };
        std::vector<std::vector<Cand>> sets(batch());
        for (size_t g = 0; g < batch(); g++) {
            auto& set = sets[g];
            for (size_t i = 0; i < nodes[g].children.size(); i++)
                set.push_back({nodes[g].children[i].second.logit + gumbel[g][i], i});
            std::stable_sort(set.begin(), set.end(), [](const Cand& a, const Cand& b) { return a.key > b.key; });
            if (set.size() > sampled_actions) set.resize(sampled_actions);
        }
        uint32_t steps = lg, visits_per_step = search_budget / steps, visits_to_most = 0;
        size_t remaining = sampled_actions;
        std::vector<float> zero_betas(batch(), 0.0f);
        for (uint32_t s = 0; s < steps; s++) {
            uint32_t visits_per_action = visits_per_step / (uint32_t)remaining;
            for (size_t i = 0; i < remaining; i++) {
                std::vector<Node<E>*> roots(batch());
                std::vector<E> cenvs;
                cenvs.reserve(batch());
                for (size_t g = 0; g < batch(); g++) {
                    size_t k = i % sets[g].size();
                    auto& ch = nodes[g].children[sets[g][k].child];
                    E env = envs[g];
                    env.step(ch.first);
                    roots[g] = &ch.second;
                    cenvs.push_back(env);
                }
                for (uint32_t v = 0; v < visits_per_action; v++) simulate_from(roots, cenvs, agent, zero_betas);
            }
            visits_to_most += visits_per_action;
            remaining /= 2;
            for (size_t g = 0; g < batch(); g++) {
                auto& set = sets[g];
                float beta = betas[g];
                std::vector<std::pair<float, Cand>> keyed;
                for (auto& c : set) {
                    const Node<E>& ch = nodes[g].children[c.child].second;
                    float k = c.key + sigma_select(ch.evaluation.negate().to_notnan(), ch.std_dev, beta, (float)visits_to_most);
                    keyed.push_back({k, c});
                }
                std::stable_sort(keyed.begin(), keyed.end(),
                                 [](const std::pair<float, Cand>& a, const std::pair<float, Cand>& b) { return a.first > b.first; });
                set.clear();
                for (size_t j = 0; j < keyed.size() && j < remaining; j++) set.push_back(keyed[j].second);
            }
        }
        selected.resize(batch());
        for (size_t g = 0; g < batch(); g++) selected[g] = nodes[g].children[sets[g][0].child].first;
        // recompute root statistics, batched.rs:373-406
        for (auto& node : nodes) {
            uint32_t sum = 0;
            bool any_loss = false, all_known = true;
            for (auto& c : node.children) {
                sum += c.second.visit_count;
                any_loss = any_loss || c.second.evaluation.is_loss();
                all_known = all_known && c.second.evaluation.is_known();
            }
            node.visit_count = sum + 1;
            if (any_loss || all_known) {
                node.evaluation = node.min_child_eval().negate();
                node.std_dev = 0.0f;
            } else {
                float sp = 0.0f, wq = 0.0f;
                for (auto& c : node.children)
                    if (c.second.visit_count > 0) sp = sp + c.second.probability;
                for (auto& c : node.children)
                    if (c.second.visit_count > 0) wq = wq + c.second.probability * c.second.evaluation.negate().to_f32();
                node.evaluation = Eval::Value(wq / sp);
            }
        }
    }
};

// ------------------------------------------------------------------ Tak environment + agents
struct TakEnv {  // Environment for Game<N,HALF_KOMI>, env.rs:33-96
    Game g;
    void populate_actions(std::vector<int>& out) const {
        std::vector<Move> mv;
        g.possible_moves(mv);
        out.clear();
        for (auto& m : mv) out.push_back(move_index(g.n, m));
    }
    void step(int action) { g.play(move_from_index(g.n, action)); }
    int terminal() const { return g.terminal(); }
    int steps() const { return g.ply; }
};

// agent.rs:16-42
template <class E>
struct DummyAgent : Agent<E> {
    void policy_value_uncertainty(const std::vector<E>&, const std::vector<std::vector<int>>& actions,
                                  std::vector<std::vector<float>>& logits, std::vector<float>& value,
                                  std::vector<float>& variance) override {
        logits.clear();
        for (auto& a : actions) logits.emplace_back(a.size(), 1.0f);
        value.assign(actions.size(), 0.0f);
        variance.assign(actions.size(), 0.0f);
    }
};
// agent.rs:44-87
struct SimpleAgent : Agent<TakEnv> {
    void policy_value_uncertainty(const std::vector<TakEnv>& envs, const std::vector<std::vector<int>>& actions,
                                  std::vector<std::vector<float>>& logits, std::vector<float>& value,
                                  std::vector<float>& variance) override {
        logits.clear();
        value.clear();
        variance.assign(envs.size(), 0.0f);
        for (size_t i = 0; i < envs.size(); i++) {
            const Game& g = envs[i].g;
            // i8 arithmetic: flat_diff - HALF_KOMI / 2 (integer division), agent.rs:67
            int v = g.flat_diff() - g.half_komi / 2;
            float fcd = (float)v / (float)(g.n * g.n);
            if (g.to_move == 1) fcd = -fcd;
            std::vector<float> l;
            for (int a : actions[i]) {
                Move m = move_from_index(g.n, a);
                l.push_back(m.spread ? 1.0f : m.piece == FLAT ? 4.0f : m.piece == CAP ? 3.0f : 2.0f);
            }
            logits.push_back(std::move(l));
            value.push_back(fcd);
        }
    }
};

// env.rs:108-209 (test-only environment of the reference)
struct SafeCrack {
    std::vector<uint8_t> key, tried;
    bool active = true;
    void populate_actions(std::vector<int>& out) const {
        out.clear();
        if (active)
            for (int i = 0; i <= 9; i++) out.push_back(i);
        else
            out.push_back(10);  // None
    }
    void step(int a) {
        if (active) tried.push_back((uint8_t)a);
        active = !active;
    }
    int terminal() const { return TZ_TERMINAL_NONE; }
    bool solved() const {
        if (tried.size() < key.size()) return false;
        for (size_t i = 0; i < key.size(); i++)
            if (tried[i] != key[i]) return false;
        return true;
    }
};
struct SafeCracker : Agent<SafeCrack> {
    void policy_value_uncertainty(const std::vector<SafeCrack>& envs, const std::vector<std::vector<int>>& actions,
                                  std::vector<std::vector<float>>& logits, std::vector<float>& value,
                                  std::vector<float>& variance) override {
        logits.clear();
        value.clear();
        variance.assign(envs.size(), 0.0f);
        for (size_t i = 0; i < envs.size(); i++) {
            logits.emplace_back(actions[i].size(), 1.0f);
            value.push_back((envs[i].active ? 1.0f : -1.0f) * (envs[i].solved() ? 1.0f : 0.0f));
        }
    }
};

}  // namespace tzo
