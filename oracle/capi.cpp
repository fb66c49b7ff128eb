// oracle/capi.cpp — C entry points over the CPU oracle for ctypes (tests/, smoke(), bench.py's
// cpu_baseline leg).  TEST INFRASTRUCTURE ONLY: the product (takzero_amd/) never links this.
// The tzo_search_* surface mirrors include/takzero_hip.h one-to-one so parity tests can drive
// both engines with the same calls.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mcts.hpp"

using namespace tzo;

extern "C" {

// callback agent: logits_out is [n_envs][amax]
typedef void (*tzo_agent_fn)(void* user, int n_envs, const tz_state* states, const uint16_t* legal_idx,
                             const int32_t* legal_count, int amax, float* logits_out, float* value_out,
                             float* variance_out);

void tzo_set_use_libm(int on) { g_use_libm = on != 0; }

// ---- scalar helpers ----
float tzo_expf(float x) { return tz_expf(x); }
float tzo_logf(float x) { return tz_logf(x); }
float tzo_powif(float a, int b) { return tz_powif(a, b); }
void tzo_softmax(const float* in, int n, float* out) {
    std::vector<float> l(in, in + n), o;
    softmax(l, o);
    memcpy(out, o.data(), n * sizeof(float));
}
int tzo_eval_cmp(int tag_a, uint32_t bits_a, int tag_b, uint32_t bits_b) {
    Eval a, b;
    a.tag = tag_a;
    b.tag = tag_b;
    if (tag_a == TZ_EVAL_VALUE) a.value = tz_bits_to_float(bits_a); else a.ply = bits_a;
    if (tag_b == TZ_EVAL_VALUE) b.value = tz_bits_to_float(bits_b); else b.ply = bits_b;
    return eval_cmp(a, b);
}
float tzo_eval_to_f32(int tag, uint32_t bits) {
    Eval a;
    a.tag = tag;
    if (tag == TZ_EVAL_VALUE) a.value = tz_bits_to_float(bits); else a.ply = bits;
    return a.to_f32();
}
float tzo_exploration_rate(float n) { return exploration_rate(n); }
float tzo_ucb(float parent, float visits, float prob) { return ucb_with_predictor(parent, visits, prob); }

// ---- game helpers ----
int tzo_state_default(int n, int half_komi, tz_state* out) {
    Game g(n, half_komi);
    g.to_state(*out);
    return 0;
}
int tzo_state_from_tps(const char* tps, int n, int half_komi, tz_state* out) {
    Game g;
    if (!Game::from_tps(tps, n, half_komi, g)) return -2;
    g.to_state(*out);
    return 0;
}
int tzo_state_to_tps(const tz_state* s, char* buf, int buflen) {
    std::string t = Game::from_state(*s).to_tps();
    if ((int)t.size() + 1 > buflen) return -1;
    memcpy(buf, t.c_str(), t.size() + 1);
    return 0;
}
int tzo_possible_moves(const tz_state* s, uint16_t* out, int cap) {
    Game g = Game::from_state(*s);
    std::vector<Move> mv;
    g.possible_moves(mv);
    if ((int)mv.size() > cap) return -1;
    for (size_t i = 0; i < mv.size(); i++) out[i] = (uint16_t)move_index(g.n, mv[i]);
    return (int)mv.size();
}
int tzo_play(const tz_state* s, uint16_t move, tz_state* out) {
    Game g = Game::from_state(*s);
    Move m = move_from_index(g.n, move);
    if (!g.is_legal(m)) return -1;
    if (!g.play(m)) return -1;
    g.to_state(*out);
    return 0;
}
int tzo_terminal(const tz_state* s) { return Game::from_state(*s).terminal(); }
int tzo_result(const tz_state* s) { return Game::from_state(*s).result(); }
int tzo_flat_diff(const tz_state* s) { return Game::from_state(*s).flat_diff(); }
int tzo_input_channels(int n) { return input_channels(n); }
int tzo_output_channels(int n) { return output_channels(n); }
int tzo_game_repr(const tz_state* s, float* out) {
    Game g = Game::from_state(*s);
    int len = input_channels(g.n) * g.n * g.n;
    memset(out, 0, len * sizeof(float));
    game_repr(g, out);
    return len;
}
int tzo_move_to_ptn(int n, uint16_t idx, char* buf, int buflen) {
    std::string t = move_to_ptn(move_from_index(n, idx));
    if ((int)t.size() + 1 > buflen) return -1;
    memcpy(buf, t.c_str(), t.size() + 1);
    return 0;
}
int tzo_move_from_ptn(int n, const char* ptn, uint16_t* out) {
    Move m;
    if (!move_from_ptn(ptn, m)) return -2;
    if (m.x >= n || m.y >= n) return -2;
    if (m.spread && (m.carried() > n || m.ndrops >= n)) return -2;
    *out = (uint16_t)move_index(n, m);
    return 0;
}
int tzo_new_opening(int n, int half_komi, int choice, tz_state* out) {
    new_opening(n, half_komi, choice).to_state(*out);
    return 0;
}

// ---- search ----
struct CallbackAgent : Agent<TakEnv> {
    tzo_agent_fn fn = nullptr;
    void* user = nullptr;
    void policy_value_uncertainty(const std::vector<TakEnv>& envs, const std::vector<std::vector<int>>& actions,
                                  std::vector<std::vector<float>>& logits, std::vector<float>& value,
                                  std::vector<float>& variance) override {
        int b = (int)envs.size();
        int amax = 1;
        for (auto& a : actions) amax = std::max(amax, (int)a.size());
        std::vector<tz_state> st(b);
        std::vector<uint16_t> idx((size_t)b * amax, 0);
        std::vector<int32_t> cnt(b);
        for (int i = 0; i < b; i++) {
            envs[i].g.to_state(st[i]);
            cnt[i] = (int)actions[i].size();
            for (size_t j = 0; j < actions[i].size(); j++) idx[(size_t)i * amax + j] = (uint16_t)actions[i][j];
        }
        std::vector<float> lo((size_t)b * amax, 0.0f);
        value.assign(b, 0.0f);
        variance.assign(b, 0.0f);
        fn(user, b, st.data(), idx.data(), cnt.data(), amax, lo.data(), value.data(), variance.data());
        logits.clear();
        for (int i = 0; i < b; i++)
            logits.emplace_back(lo.begin() + (size_t)i * amax, lo.begin() + (size_t)i * amax + cnt[i]);
    }
};

struct tzo_search {
    int n, half_komi;
    std::unique_ptr<BatchedMCTS<TakEnv>> mcts;
    std::unique_ptr<Agent<TakEnv>> agent;
    std::vector<int8_t> term_reason;   // of the games the last restart_terminal found finished
    std::vector<uint8_t> term_winner;
};

tzo_search* tzo_search_create(int agent_kind, tzo_agent_fn fn, void* user, int batch, int n, int half_komi) {
    auto* s = new tzo_search();
    s->n = n;
    s->half_komi = half_komi;
    std::vector<TakEnv> envs(batch);
    for (auto& e : envs) e.g = Game(n, half_komi);
    s->mcts.reset(new BatchedMCTS<TakEnv>(envs));
    if (agent_kind == TZ_AGENT_DUMMY) s->agent.reset(new DummyAgent<TakEnv>());
    else if (agent_kind == TZ_AGENT_SIMPLE) s->agent.reset(new SimpleAgent());
    else {
        auto* c = new CallbackAgent();
        c->fn = fn;
        c->user = user;
        s->agent.reset(c);
    }
    return s;
}
void tzo_search_destroy(tzo_search* s) { delete s; }
int tzo_search_set_positions(tzo_search* s, int count, const int32_t* game_idx, const tz_state* states) {
    for (int i = 0; i < count; i++) {
        int g = game_idx[i];
        if (g < 0 || g >= (int)s->mcts->batch()) return -1;
        s->mcts->envs[g].g = Game::from_state(states[i]);
        s->mcts->nodes[g] = Node<TakEnv>();
        s->mcts->replays[g].clear();
    }
    return 0;
}
int tzo_search_get_positions(tzo_search* s, tz_state* out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) s->mcts->envs[g].g.to_state(out[g]);
    return 0;
}
int tzo_search_new_openings(tzo_search* s, const int32_t* choice) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        s->mcts->envs[g].g = new_opening(s->n, s->half_komi, choice[g]);
        s->mcts->nodes[g] = Node<TakEnv>();
        s->mcts->replays[g].clear();
    }
    return 0;
}
int tzo_search_simulate(tzo_search* s, const float* betas, int n_sims) {
    std::vector<float> b(betas, betas + s->mcts->batch());
    for (int i = 0; i < n_sims; i++) s->mcts->simulate(*s->agent, b);
    return 0;
}
int tzo_search_apply_noise(tzo_search* s, const float* noise, int amax, float ratio) {
    for (size_t g = 0; g < s->mcts->batch(); g++)
        if (!s->mcts->nodes[g].apply_dirichlet(noise + g * amax, ratio)) return -6;
    return 0;
}
int tzo_search_root_info(tzo_search* s, tz_root_info* out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        const auto& nd = s->mcts->nodes[g];
        tz_root_info& r = out[g];
        memset(&r, 0, sizeof r);
        r.visit_count = nd.visit_count;
        r.n_children = (uint32_t)nd.children.size();
        r.eval_tag = nd.evaluation.tag;
        r.is_terminal_env = s->mcts->envs[g].terminal() != TZ_TERMINAL_NONE;
        r.ply = (uint16_t)s->mcts->envs[g].g.ply;
        if (nd.evaluation.tag == TZ_EVAL_VALUE) r.eval.value = nd.evaluation.value; else r.eval.ply = nd.evaluation.ply;
        r.std_dev = nd.std_dev;
        r.logit = nd.logit;
        r.probability = nd.probability;
    }
    return 0;
}
int tzo_search_root_children(tzo_search* s, int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag,
                             uint32_t* eval_bits, float* logit, float* prob, float* std_dev) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        const auto& nd = s->mcts->nodes[g];
        if ((int)nd.children.size() > amax) return -1;
        for (size_t i = 0; i < nd.children.size(); i++) {
            const auto& c = nd.children[i].second;
            size_t o = g * amax + i;
            if (move_idx) move_idx[o] = (uint16_t)nd.children[i].first;
            if (visits) visits[o] = c.visit_count;
            if (eval_tag) eval_tag[o] = c.evaluation.tag;
            if (eval_bits) eval_bits[o] = c.evaluation.bits();
            if (logit) logit[o] = c.logit;
            if (prob) prob[o] = c.probability;
            if (std_dev) std_dev[o] = c.std_dev;
        }
    }
    return 0;
}
int tzo_search_select_best_actions(tzo_search* s, uint16_t* out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        const auto& nd = s->mcts->nodes[g];
        out[g] = nd.children.empty() ? 0xFFFF : (uint16_t)nd.select_best_action();
    }
    return 0;
}
int tzo_search_improved_policy(tzo_search* s, float visitations, int amax, float* out) {
    std::vector<float> p;
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        s->mcts->nodes[g].improved_policy(visitations, p);
        if ((int)p.size() > amax) return -1;
        for (size_t i = 0; i < p.size(); i++) out[g * amax + i] = p[i];
    }
    return 0;
}
int tzo_search_ube_target(tzo_search* s, float beta, float* out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) out[g] = s->mcts->nodes[g].ube_target(beta);
    return 0;
}
int tzo_search_selfplay_weights(tzo_search* s, int game, int sample, uint32_t threshold, float drop, uint32_t* w_out) {
    std::vector<uint32_t> w;
    bool any = s->mcts->nodes[game].selfplay_weights(sample != 0, threshold, drop, w);
    for (size_t i = 0; i < w.size(); i++) w_out[i] = w[i];
    return any ? 1 : 0;
}
int tzo_search_step(tzo_search* s, const uint16_t* actions) {
    std::vector<int> a(actions, actions + s->mcts->batch());
    s->mcts->step(a);
    return 0;
}
int tzo_search_restart_terminal(tzo_search* s, const int32_t* choice, int8_t* terminal_out) {
    s->term_reason.assign(s->mcts->batch(), 0);
    s->term_winner.assign(s->mcts->batch(), 0);
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        int t = s->mcts->envs[g].terminal();
        terminal_out[g] = (int8_t)t;
        if (t != TZ_TERMINAL_NONE) {
            const Result r = s->mcts->envs[g].g.result();
            s->term_reason[g] = (int8_t)s->mcts->envs[g].g.result_reason();
            s->term_winner[g] = r == WHITE_WINS ? 0 : r == BLACK_WINS ? 1 : 2;
            s->mcts->envs[g].g = new_opening(s->n, s->half_komi, choice[g]);
            s->mcts->nodes[g] = Node<TakEnv>();
            s->mcts->replays[g].clear();
        }
    }
    return 0;
}
int tzo_search_gumbel_sh(tzo_search* s, const float* betas, int sampled_actions, int search_budget,
                         const float* gumbel, int amax, uint16_t* selected_out) {
    size_t B = s->mcts->batch();
    {   // batched.rs:216-220 asserts that the budget is a whole number of halving rounds
        int lg = 0;
        while ((1 << (lg + 1)) <= sampled_actions) lg++;
        if (sampled_actions <= 0 || lg == 0 || search_budget % (sampled_actions * lg)) return -1;  // ilog2(k) * k
    }
    std::vector<float> b(betas, betas + B);
    std::vector<std::vector<float>> g(B);
    for (size_t i = 0; i < B; i++) g[i].assign(gumbel + i * amax, gumbel + (i + 1) * amax);
    std::vector<int> sel;
    s->mcts->gumbel_sequential_halving(*s->agent, b, (size_t)sampled_actions, (uint32_t)search_budget, g, sel);
    for (size_t i = 0; i < B; i++) selected_out[i] = (uint16_t)sel[i];
    return 0;
}
// the rest of the search ABI, so that a driver written against tz_search_* can run over this oracle unchanged
int tzo_search_shape(tzo_search* s, int* batch_out, int* n_out, int* half_komi_out, int* max_actions_out) {
    if (batch_out) *batch_out = (int)s->mcts->batch();
    if (n_out) *n_out = s->n;
    if (half_komi_out) *half_komi_out = s->half_komi;
    if (max_actions_out) *max_actions_out = s->n <= 3 ? 64 : s->n == 4 ? 192 : s->n == 5 ? 512 : 1024;
    return 0;
}
int tzo_search_terminal_details(tzo_search* s, int8_t* reason_out, uint8_t* winner_out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        reason_out[g] = g < s->term_reason.size() ? s->term_reason[g] : 0;
        winner_out[g] = g < s->term_winner.size() ? s->term_winner[g] : 0;
    }
    return 0;
}
int tzo_search_improved_policy_each(tzo_search* s, const float* visitations, int amax, float* out) {
    std::vector<float> p;
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        s->mcts->nodes[g].improved_policy(visitations[g], p);
        if ((int)p.size() > amax) return -1;
        for (int i = 0; i < amax; i++) out[g * amax + i] = i < (int)p.size() ? p[i] : 0.0f;
    }
    return 0;
}
// validated move application (Replay::from_str / Replay::states): 1 applied, 0 illegal or none, -1 already terminal
int tzo_search_play_moves(tzo_search* s, const uint16_t* actions, int8_t* ok_out) {
    for (size_t g = 0; g < s->mcts->batch(); g++) {
        Game& game = s->mcts->envs[g].g;
        s->mcts->nodes[g] = Node<TakEnv>();
        if (game.terminal() != TZ_TERMINAL_NONE) {
            ok_out[g] = -1;
            continue;
        }
        ok_out[g] = 0;
        if (actions[g] == 0xFFFF) continue;
        std::vector<Move> legal;
        game.possible_moves(legal);
        for (const Move& m : legal)
            if (move_index(s->n, m) == (int)actions[g]) {
                game.play(m);
                ok_out[g] = 1;
                break;
            }
    }
    return 0;
}
int tzo_search_counters(tzo_search* s, uint64_t* sims, uint64_t* evals) {
    *sims = s->mcts->simulations;
    *evals = s->mcts->nn_leaf_evals;
    return 0;
}
int tzo_search_replay(tzo_search* s, int game, uint16_t* out, int cap) {
    auto& r = s->mcts->replays[game];
    if ((int)r.size() > cap) return -1;
    for (size_t i = 0; i < r.size(); i++) out[i] = (uint16_t)r[i];
    return (int)r.size();
}

// ---- reference KAT drivers (mcts.rs:346-445) ----
// returns the number of simulate_simple calls until the root is proven Win, or -1.
int tzo_kat_find_tinue(int n, int half_komi, const char* const* ptn_moves, int n_moves, int agent_kind, float beta,
                       int max_visits, uint16_t* loss_child_move) {
    TakEnv env;
    env.g = Game(n, half_komi);
    for (int i = 0; i < n_moves; i++) {
        Move m;
        if (!move_from_ptn(ptn_moves[i], m) || !env.g.is_legal(m)) return -2;
        env.g.play(m);
    }
    Node<TakEnv> root;
    DummyAgent<TakEnv> dummy;
    SimpleAgent simple;
    Agent<TakEnv>& agent = agent_kind == TZ_AGENT_DUMMY ? (Agent<TakEnv>&)dummy : (Agent<TakEnv>&)simple;
    for (int i = 0; i < max_visits; i++) {
        auto p = root.simulate_simple(agent, env, beta);
        if (p.eval.is_win()) {
            for (auto& c : root.children)
                if (c.second.evaluation.is_loss()) {
                    *loss_child_move = (uint16_t)c.first;
                    return i + 1;
                }
            return -3;
        }
    }
    return -1;
}
// safe_cracker_value_propagation, mcts.rs:414-445. returns 0 if every assertion holds.
int tzo_kat_safecrack(int visits) {
    const uint8_t KEY[5] = {0, 1, 2, 3, 4};
    SafeCrack env;
    env.key.assign(KEY, KEY + 5);
    Node<SafeCrack> root;
    SafeCracker agent;
    if (root.evaluation.to_f32() != 0.0f) return 1;
    for (int i = 0; i < visits; i++) root.simulate_simple(agent, env, 0.0f);
    for (int k = 0; k < 5; k++) {
        if (!(root.evaluation.to_f32() > 0.0f)) return 10 + k;
        for (auto& c : root.children) {
            if (c.first == KEY[k]) {
                if (!(c.second.evaluation.to_f32() < 0.0f)) return 20 + k;
            } else if (c.second.evaluation.to_f32() != 0.0f) {
                return 30 + k;
            }
        }
        root.descend(KEY[k]);
        root.descend(10);
    }
    if (!(root.evaluation.to_f32() > 0.0f)) return 2;
    return 0;
}

}  // extern "C"

// ---- the decisions above the search (oracle/host.hpp), attached to a tzo_search: tests/host_over_oracle.cpp lets the
// product's native drivers run over this search and checks every decision they take against these ----
#include "host.hpp"

struct tzo_host {
    tzo_search* s;
    SelfplayHost sp;
    std::vector<CompleteTarget> done;
    explicit tzo_host(tzo_search* search) : s(search), sp(search->mcts->batch()) {}
};

extern "C" {

tzo_host* tzo_host_create(tzo_search* s) { return new tzo_host(s); }
void tzo_host_destroy(tzo_host* h) { delete h; }

// expected actions (0xFFFF for a root without children) and whether each game consumed its draw; then take_a_step's record
int tzo_host_choose_and_record(tzo_host* h, int kind, const uint16_t* halving, const double* draws, float visitations, uint16_t* expected_out,
                               uint8_t* sampled_out) {
    const size_t B = h->s->mcts->batch();
    std::vector<int> hv(B), out;
    std::vector<double> d(draws, draws + B);
    std::vector<uint8_t> sampled;
    for (size_t g = 0; g < B; g++) hv[g] = halving[g];
    h->sp.choose(*h->s->mcts, kind, hv, d, out, sampled);
    for (size_t g = 0; g < B; g++) {
        expected_out[g] = out[g] < 0 ? 0xFFFF : (uint16_t)out[g];
        sampled_out[g] = sampled[g];
    }
    h->sp.record(*h->s->mcts, kind, visitations);
    return 0;
}

// restart_envs_and_complete_targets for the games in `terminal`; returns the number of completed targets, kept for tzo_host_target
int tzo_host_complete(tzo_host* h, const int8_t* terminal, const float* betas) {
    const size_t B = h->s->mcts->batch();
    std::vector<int> t(B);
    std::vector<float> b(betas, betas + B);
    for (size_t g = 0; g < B; g++) t[g] = terminal[g];
    h->done.clear();
    h->sp.complete(t, b, h->done);
    return (int)h->done.size();
}

int tzo_host_target(tzo_host* h, int i, tz_state* state_out, int amax, uint16_t* moves_out, float* policy_out, float* value_out, float* ube_out) {
    if (i < 0 || i >= (int)h->done.size()) return -1;
    const CompleteTarget& t = h->done[i];
    if ((int)t.moves.size() > amax) return -1;
    t.env.g.to_state(*state_out);
    for (size_t k = 0; k < t.moves.size(); k++) {
        moves_out[k] = (uint16_t)t.moves[k];
        policy_out[k] = t.policy[k];
    }
    *value_out = t.value;
    *ube_out = t.ube;
    return (int)t.moves.size();
}

// reanalyze target of game g for the selected action (reanalyze/src/main.rs:184-203)
int tzo_host_reanalyze_target(tzo_search* s, int g, uint16_t selected, int amax, uint16_t* moves_out, float* policy_out, float* value_out,
                              float* ube_out) {
    const CompleteTarget t = reanalyze_target(s->mcts->nodes[g], s->mcts->envs[g], (int)selected);
    if ((int)t.moves.size() > amax) return -1;
    for (size_t k = 0; k < t.moves.size(); k++) {
        moves_out[k] = (uint16_t)t.moves[k];
        policy_out[k] = t.policy[k];
    }
    *value_out = t.value;
    *ube_out = t.ube;
    return (int)t.moves.size();
}

}  // extern "C"

// ---- a node below the root (mirror of tz_search_node) ----
extern "C" int tzo_search_node(tzo_search* s, int game, const uint16_t* path, int path_len, tz_root_info* node_out, int amax, uint16_t* move_idx,
                               uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits, float* logit, float* prob, float* std_dev) {
    const Node<TakEnv>* node = &s->mcts->nodes[game];
    for (int d = 0; d < path_len; d++) {
        const Node<TakEnv>* next = nullptr;
        for (auto& c : node->children)
            if (c.first == (int)path[d]) {
                next = &c.second;
                break;
            }
        if (!next) return -1;
        node = next;
    }
    if ((int)node->children.size() > amax) return -1;
    if (node_out) {
        memset(node_out, 0, sizeof *node_out);
        node_out->visit_count = node->visit_count;
        node_out->n_children = (uint32_t)node->children.size();
        node_out->eval_tag = node->evaluation.tag;
        node_out->eval.ply = node->evaluation.bits();
        node_out->std_dev = node->std_dev;
        node_out->logit = node->logit;
        node_out->probability = node->probability;
        node_out->ply = (uint16_t)(s->mcts->envs[game].steps() + path_len);
        node_out->is_terminal_env = node->is_terminal();
    }
    for (size_t i = 0; i < node->children.size(); i++) {
        const Node<TakEnv>& c = node->children[i].second;
        if (move_idx) move_idx[i] = (uint16_t)node->children[i].first;
        if (visits) visits[i] = c.visit_count;
        if (eval_tag) eval_tag[i] = c.evaluation.tag;
        if (eval_bits) eval_bits[i] = c.evaluation.bits();
        if (logit) logit[i] = c.logit;
        if (prob) prob[i] = c.probability;
        if (std_dev) std_dev[i] = c.std_dev;
    }
    return 0;
}
