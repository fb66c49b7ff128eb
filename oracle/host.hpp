// oracle/host.hpp — CPU restatement of the decisions the reference's binaries take ABOVE the search:
//   Node::select_selfplay_action            takzero/src/search/node/mod.rs:170-207
//   BatchedMCTS::select_actions_in_selfplay takzero/src/search/node/batched.rs:165-183
//   selfplay::main move choice              selfplay/src/main.rs:138-153
//   take_a_step                             selfplay/src/main.rs:238-258  (policy targets: :246-250, target.rs:151-164)
//   restart_envs_and_complete_targets       selfplay/src/main.rs:263-329
//   reanalyze target creation               reanalyze/src/main.rs:184-203
//
// TEST INFRASTRUCTURE ONLY (see oracle/tak.hpp): the product's host drivers (takzero_amd/csrc/tz_host.cpp) are checked
// against this, move by move (tests/test_host_over_oracle.py); nothing under takzero_amd/ includes it.
// Randomness is an input: the uniform draw that rand's `choose_weighted` would consume is handed in (`rand 0.10.0-rc.5`,
// Cargo.lock:1178-1180, is not under /root/reference; its WeightedIndex over integer weights is restated below).
// Operates on the oracle's own Node / Eval objects (mcts.hpp), i.e. with the real Eval order of eval.rs:138-163.
#pragma once
#include <cmath>
#include <vector>

#include "mcts.hpp"

namespace tzo {

constexpr int WEIGHTED_RANDOM_PLIES = 10;   // selfplay/src/main.rs:38
constexpr float SELFPLAY_BETA = 0.25f;      // selfplay/src/main.rs:41

// Node::select_selfplay_action (node/mod.rs:170-207).  `u` in [0, 1) stands for the generator: choose_weighted builds a
// WeightedIndex over the u32 weights, draws an integer uniformly from [0, total) and takes the first child whose cumulative
// weight exceeds it.  *sampled_out = whether a draw was consumed.
template <class E>
int select_selfplay_action(const Node<E>& node, bool has_threshold, uint32_t threshold, float allowed_eval_drop, double u,
                           bool* sampled_out = nullptr) {
    if (sampled_out) *sampled_out = false;
    if (node.evaluation.is_known()) return node.select_best_action();   // :176-178
    if (!has_threshold) return node.select_best_action();               // :179-181
    // best_eval = min over the children's evaluations (:183-188)
    Eval best = node.children[0].second.evaluation;
    for (auto& c : node.children)
        if (eval_cmp(c.second.evaluation, best) < 0) best = c.second.evaluation;
    const Eval limit = best.tag == TZ_EVAL_VALUE ? Eval::Value(best.value + allowed_eval_drop) : best;   // best_eval.map(|x| x + drop)
    std::vector<uint64_t> cumulative;
    uint64_t total = 0;
    for (auto& c : node.children) {
        const Node<E>& child = c.second;
        uint32_t w = child.visit_count;
        if (child.visit_count < threshold || child.evaluation.is_win() || eval_cmp(child.evaluation, limit) > 0) w = 0;   // :192-197
        total += w;
        cumulative.push_back(total);
    }
    if (total == 0) return node.select_best_action();   // WeightError::InsufficientNonZero (:202)
    if (sampled_out) *sampled_out = true;
    uint64_t chosen = (uint64_t)(u * (double)total);
    if (chosen >= total) chosen = total - 1;
    size_t i = 0;
    while (cumulative[i] <= chosen) i++;
    return node.children[i].first;
}

struct IncompleteTarget {   // selfplay/src/main.rs:230-236
    TakEnv env;
    std::vector<int> moves;
    std::vector<float> policy;
    float ube = 0.f;
};
struct CompleteTarget {     // target.rs Target<E>
    TakEnv env;
    std::vector<int> moves;
    std::vector<float> policy;
    float value = 0.f, ube = 0.f;
};

// The per-game Vec<IncompleteTarget> of selfplay::main and the three things it does with it.
struct SelfplayHost {
    std::vector<std::vector<IncompleteTarget>> pending;

    explicit SelfplayHost(size_t batch) : pending(batch) {}

    // The action every game plays.  kind 0 = the PUCT loop: select_actions_in_selfplay(rng, WEIGHTED_RANDOM_PLIES)
    // (selfplay/src/main.rs:127-136, batched.rs:165-183); kind 1 = what the binary runs today: the sequential-halving result,
    // replaced by select_selfplay_action(Some(32), 0.5) while env.steps() < WEIGHTED_RANDOM_PLIES (:138-153).
    // draws[g] is read only where the reference would consume the generator.
    void choose(const BatchedMCTS<TakEnv>& mcts, int kind, const std::vector<int>& halving_result, const std::vector<double>& draws,
                std::vector<int>& out, std::vector<uint8_t>& sampled) const {
        out.assign(mcts.batch(), -1);
        sampled.assign(mcts.batch(), 0);
        for (size_t g = 0; g < mcts.batch(); g++) {
            const Node<TakEnv>& node = mcts.nodes[g];
            if (node.children.empty()) continue;
            const bool early = mcts.envs[g].steps() < WEIGHTED_RANDOM_PLIES;
            bool did = false;
            if (kind == 0) out[g] = select_selfplay_action(node, early, 32, 0.5f, draws[g], &did);
            else out[g] = early ? select_selfplay_action(node, true, 32, 0.5f, draws[g], &did) : halving_result[g];
            sampled[g] = did;
        }
    }

    // take_a_step before batched_mcts.step (:238-257).  kind 0: policy_target_from_proportional_visits (target.rs:151-164);
    // kind 1: improved_policy(IMPROVED_POLICY_VISITATIONS) (:246-250); root_ube_metric = ube_target(BETA).
    void record(const BatchedMCTS<TakEnv>& mcts, int kind, float visitations) {
        for (size_t g = 0; g < mcts.batch(); g++) {
            const Node<TakEnv>& node = mcts.nodes[g];
            if (node.is_terminal()) continue;   // BatchedMCTS::step leaves a terminal root alone (batched.rs:137); it is restarted below
            IncompleteTarget t;
            t.env = mcts.envs[g];
            for (auto& c : node.children) t.moves.push_back(c.first);
            if (kind == 0) {
                for (auto& c : node.children) t.policy.push_back((float)c.second.visit_count / (float)node.visit_count);
            } else {
                node.improved_policy(visitations, t.policy);
            }
            t.ube = node.ube_target(SELFPLAY_BETA);
            pending[g].push_back(std::move(t));
        }
    }

    // restart_envs_and_complete_targets (:263-329): terminal[g] = TZ_TERMINAL_* of the game that just ended (NONE otherwise);
    // the stored roots are walked newest -> oldest with value = value.negate() starting from Eval::from(terminal); targets of
    // exploratory games (beta > 0) are kept only after the initial exploration (env.ply > WEIGHTED_RANDOM_PLIES).
    void complete(const std::vector<int>& terminal, const std::vector<float>& betas, std::vector<CompleteTarget>& out) {
        for (size_t g = 0; g < pending.size(); g++) {
            if (terminal[g] == TZ_TERMINAL_NONE) continue;
            Eval value = eval_from_terminal(terminal[g]);
            for (size_t i = pending[g].size(); i-- > 0;) {
                IncompleteTarget& t = pending[g][i];
                value = value.negate();
                if (betas[g] == 0.0f || t.env.g.ply > WEIGHTED_RANDOM_PLIES) {
                    CompleteTarget c;
                    c.env = t.env;
                    c.moves = std::move(t.moves);
                    c.policy = std::move(t.policy);
                    c.value = value.to_f32();
                    c.ube = t.ube;
                    out.push_back(std::move(c));
                }
            }
            pending[g].clear();
        }
    }
};

// reanalyze::main target creation (reanalyze/src/main.rs:184-203): value = the root's evaluation if it is solved, else the
// negated evaluation of the selected child (Eval::negate bumps the ply of a proven result before the conversion to f32);
// policy = improved_policy(most_visited_count()); ube = ube_target(0.25).
inline CompleteTarget reanalyze_target(const Node<TakEnv>& node, const TakEnv& env, int selected_action) {
    CompleteTarget t;
    t.env = env;
    Eval value = node.evaluation;
    if (!node.evaluation.is_known()) {
        for (auto& c : node.children)
            if (c.first == selected_action) {
                value = c.second.evaluation.negate();
                break;
            }
    }
    t.value = value.to_f32();
    for (auto& c : node.children) t.moves.push_back(c.first);
    node.improved_policy((float)node.most_visited_count(), t.policy);
    t.ube = node.ube_target(0.25f);
    return t;
}

}  // namespace tzo
