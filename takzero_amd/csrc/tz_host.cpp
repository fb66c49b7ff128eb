// tz_host.cpp — the host side of the reference's `selfplay` binary above the search ABI, in native code:
// the outer loop of selfplay::main (selfplay/src/main.rs:63-205), take_a_step (:238-258),
// restart_envs_and_complete_targets (:263-329), save_targets_to_file / save_replays_to_file (:332-366) and
// read_buffer_lengths (:371-387).  The reference is compiled code and so is this; takzero_amd/selfplay.py is the same
// driver in Python, kept because the tests also run it over the CPU oracle.
//
// Randomness (openings, Dirichlet / Gumbel samples, early-ply move sampling) is drawn here from one seeded
// std::mt19937_64 per shard and handed to the search as input, exactly as the Python driver does with numpy.
#include <fcntl.h>
#include <unistd.h>

#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "tz_engine.h"
#include "tz_host_exchange.h"
#include "tz_math.h"

// Every call into the search goes through these two names, so that a test build can bind this same driver to another
// implementation of the search ABI (the default is the HIP engine of this library).
#ifndef TZS
#define TZS(name) tz_search_##name
#define TZ_SEARCH_T tz_search
#endif

namespace {

// OpenOptions::append(true).create(true) + ONE write of the whole text (selfplay/src/main.rs:332-366): several processes
// append to the same file - one per GPU here, 10 selfplay + 10 reanalyze processes in the reference's deployment
// (README.md:130) - and a reader must never find two writers' bytes interleaved inside a line.  stdio would split the text
// into buffer-sized writes.
bool append_whole(const std::string& path, const std::string& text) {
    if (text.empty()) return true;
    const int fd = open(path.c_str(), O_WRONLY | O_APPEND | O_CREAT, 0644);
    if (fd < 0) return false;
    size_t done = 0;
    while (done < text.size()) {   // a regular file takes it in one call; the loop is for the contract of write()
        const ssize_t w = write(fd, text.data() + done, text.size() - done);
        if (w <= 0) break;
        done += (size_t)w;
    }
    close(fd);
    return done == text.size();
}

constexpr float NOISE_ALPHA = 0.05f;          // selfplay/src/main.rs:39
constexpr float NOISE_RATIO = 0.2f;           // selfplay/src/main.rs:40
constexpr int WEIGHTED_RANDOM_PLIES = 10;     // selfplay/src/main.rs:38
constexpr float BETA = 0.25f;                 // selfplay/src/main.rs:41
constexpr uint32_t SAMPLE_THRESHOLD = 32;     // batched.rs:175
constexpr float ALLOWED_EVAL_DROP = 0.5f;     // batched.rs:176

struct MoveRecord {  // IncompleteTarget for every game of the shard at one move (selfplay/src/main.rs:230-236)
    std::vector<tz_state> states;
    std::vector<uint16_t> moves;   // [B][width]
    std::vector<float> pol;        // [B][width]
    std::vector<float> ube;        // [B]
    std::vector<uint32_t> nchild;  // [B]
    std::vector<uint8_t> stepped;  // [B]
    std::vector<uint16_t> actions; // [B]
    int width = 0;
};

// impl Ord for Eval (eval.rs:138-163) on (tag, bits) pairs as the ABI hands them over: Loss(a) < Loss(b) iff a < b; any Loss <
// any Value / Draw < any Win; Win(a) < Win(b) iff a > b; a Draw compares with a Value as CONTEMPT = -0.05 does (and is Equal
// to a Value of exactly -0.05); Draw(a) < Draw(b) iff a > b.
int eval_cmp(uint8_t ta, uint32_t ba, uint8_t tb, uint32_t bb) {
    constexpr float CONTEMPT = -0.05f;   // eval.rs:128
    auto cmpf = [](float a, float b) { return a < b ? -1 : a > b ? 1 : 0; };
    auto cmpu = [](uint32_t a, uint32_t b) { return a < b ? -1 : a > b ? 1 : 0; };
    switch (ta) {
        case TZ_EVAL_LOSS: return tb == TZ_EVAL_LOSS ? cmpu(ba, bb) : -1;
        case TZ_EVAL_WIN: return tb == TZ_EVAL_WIN ? cmpu(bb, ba) : 1;
        case TZ_EVAL_DRAW:
            switch (tb) {
                case TZ_EVAL_LOSS: return 1;
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmpu(bb, ba);
                default: return cmpf(CONTEMPT, tz_bits_to_float(bb));
            }
        default:
            switch (tb) {
                case TZ_EVAL_LOSS: return 1;
                case TZ_EVAL_WIN: return -1;
                case TZ_EVAL_DRAW: return cmpf(tz_bits_to_float(ba), CONTEMPT);
                default: return cmpf(tz_bits_to_float(ba), tz_bits_to_float(bb));
            }
    }
}

float eval_to_f32(uint8_t tag, uint32_t ply) {  // impl From<Eval> for f32 (eval.rs:95-105) for proven results
    const float base = tz_powif(TZ_DISCOUNT, (int)ply);
    return base * (tag == TZ_EVAL_WIN ? 1.0f : tag == TZ_EVAL_LOSS ? -1.0f : 0.0f);
}

const char* result_string(int reason, int winner) {  // takparse GameResult (target.rs:226-230)
    if (winner == 2) return "1/2-1/2";
    if (reason == 1) return winner == 0 ? "R-0" : "0-R";
    return winner == 0 ? "F-0" : "0-F";
}

}  // namespace

struct tz_selfplay {
    TZ_SEARCH_T* search = nullptr;
    int B = 0, n = 0, half_komi = 0, amax = 0;
    int sims = 0, kind = 0, k = 64;
    std::vector<float> betas;
    std::mt19937_64 rng;
    std::deque<MoveRecord> history;
    int64_t history_base = 0, moves_played = 0;
    std::vector<int64_t> game_start;
    std::vector<tz_state> start_states;
    // finished work, as text (drained by tz_selfplay_take_text or appended to files by the run loop)
    std::string targets_text, replays_text, exploration_text;
    uint64_t n_targets = 0, n_replays = 0, positions = 0;
    // scratch
    std::vector<tz_root_info> info;
    std::vector<uint16_t> c_moves, actions, best;
    std::vector<uint32_t> c_visits, c_bits;
    std::vector<uint8_t> c_tag;
    std::vector<float> noise, gumbel, pol, ube;
    std::vector<int32_t> choice;
    std::vector<int8_t> term, reason;
    std::vector<uint8_t> winner;
    // completed targets of one move, columnar, for tz_format_targets
    std::vector<tz_state> t_states;
    std::vector<uint16_t> t_moves;
    std::vector<float> t_pol, t_value, t_ube;
    std::vector<int32_t> t_n;
    // N shards (tz_selfplay_set_comm): the targets a move finished stay packed until tz_selfplay_exchange has gathered
    // every rank's; records = [u32 n][f32 value][f32 ube][tz_state][u16 move x n, padded to 4][f32 p x n]
    bool has_exchange = false;
    HostExchange xch;
    std::vector<unsigned char> records;
    uint64_t n_gathered = 0;
    // tests: observer of the decisions (tz_host_exchange.h HostTrace) and what it is shown
    const HostTrace* trace = nullptr;
    std::vector<double> draws;
    std::vector<uint16_t> halving;
};

namespace {

// packed target records of the exchange (tz_selfplay struct): append / parse
void pack_records(int n, int T, int amax, const std::vector<tz_state>& st, const std::vector<uint16_t>& moves, const std::vector<float>& pol,
                  const std::vector<int32_t>& cnt, const std::vector<float>& value, const std::vector<float>& ube,
                  std::vector<unsigned char>& out) {
    (void)n;
    for (int i = 0; i < T; i++) {
        const uint32_t c = (uint32_t)cnt[i];
        const size_t mv_bytes = ((size_t)c * 2 + 3) / 4 * 4;
        const size_t at = out.size();
        out.resize(at + 12 + sizeof(tz_state) + mv_bytes + (size_t)c * 4, 0);
        unsigned char* p = out.data() + at;
        memcpy(p, &c, 4);
        memcpy(p + 4, &value[i], 4);
        memcpy(p + 8, &ube[i], 4);
        memcpy(p + 12, &st[i], sizeof(tz_state));
        memcpy(p + 12 + sizeof(tz_state), moves.data() + (size_t)i * amax, (size_t)c * 2);
        memcpy(p + 12 + sizeof(tz_state) + mv_bytes, pol.data() + (size_t)i * amax, (size_t)c * 4);
    }
}

int unpack_records(const std::vector<unsigned char>& in, int amax, std::vector<tz_state>& st, std::vector<uint16_t>& moves,
                   std::vector<float>& pol, std::vector<int32_t>& cnt, std::vector<float>& value, std::vector<float>& ube) {
    size_t at = 0;
    while (at < in.size()) {
        if (at + 12 + sizeof(tz_state) > in.size()) return tz_fail(TZ_EPARSE, "exchange: truncated target record");
        uint32_t c;
        float v, u;
        memcpy(&c, in.data() + at, 4);
        memcpy(&v, in.data() + at + 4, 4);
        memcpy(&u, in.data() + at + 8, 4);
        const size_t mv_bytes = ((size_t)c * 2 + 3) / 4 * 4;
        if ((int)c > amax || at + 12 + sizeof(tz_state) + mv_bytes + (size_t)c * 4 > in.size()) return tz_fail(TZ_EPARSE, "exchange: bad target record");
        tz_state s;
        memcpy(&s, in.data() + at + 12, sizeof s);
        st.push_back(s);
        cnt.push_back((int32_t)c);
        value.push_back(v);
        ube.push_back(u);
        const size_t row = moves.size();
        moves.resize(row + amax, 0);
        pol.resize(row + amax, 0.f);
        memcpy(moves.data() + row, in.data() + at + 12 + sizeof s, (size_t)c * 2);
        memcpy(pol.data() + row, in.data() + at + 12 + sizeof s + mv_bytes, (size_t)c * 4);
        at += 12 + sizeof s + mv_bytes + (size_t)c * 4;
    }
    return TZ_OK;
}

int fetch_children(tz_selfplay* sp, int* width_out) {
    int rc = TZS(root_info)(sp->search, sp->info.data());
    if (rc) return rc;
    int w = 1;
    for (int g = 0; g < sp->B; g++) w = std::max(w, (int)sp->info[g].n_children);
    const size_t cells = (size_t)sp->B * w;
    sp->c_moves.resize(cells);
    sp->c_visits.resize(cells);
    sp->c_tag.resize(cells);
    sp->c_bits.resize(cells);
    *width_out = w;
    return TZS(root_children)(sp->search, w, sp->c_moves.data(), sp->c_visits.data(), sp->c_tag.data(), sp->c_bits.data(),
                                   nullptr, nullptr, nullptr);
}

// BatchedMCTS::select_actions_in_selfplay (batched.rs:165-183) / Node::select_selfplay_action (node/mod.rs:170-207)
int select_actions_in_selfplay(tz_selfplay* sp, std::vector<uint16_t>& out) {
    int rc = TZS(select_best_actions)(sp->search, sp->best.data());
    if (rc) return rc;
    out = sp->best;
    int w = 0;
    if ((rc = fetch_children(sp, &w))) return rc;
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    std::vector<double> weight(w);
    for (int g = 0; g < sp->B; g++) {
        const tz_root_info& ri = sp->info[g];
        const int nc = (int)ri.n_children;
        if (ri.ply >= WEIGHTED_RANDOM_PLIES || ri.eval_tag != TZ_EVAL_VALUE || nc == 0) continue;
        const size_t o = (size_t)g * w;
        int bi = 0;   // best_eval = the first minimum of the children's evaluations (Iterator::min)
        for (int i = 1; i < nc; i++)
            if (eval_cmp(sp->c_tag[o + i], sp->c_bits[o + i], sp->c_tag[o + bi], sp->c_bits[o + bi]) < 0) bi = i;
        // best_eval.map(|x| x + allowed_eval_drop): only a Value moves
        const uint8_t limit_tag = sp->c_tag[o + bi];
        const uint32_t limit_bits = limit_tag == TZ_EVAL_VALUE ? tz_float_to_bits(tz_bits_to_float(sp->c_bits[o + bi]) + ALLOWED_EVAL_DROP)
                                                               : sp->c_bits[o + bi];
        double total = 0.0;
        for (int i = 0; i < nc; i++) {
            const bool ok = sp->c_visits[o + i] >= SAMPLE_THRESHOLD && sp->c_tag[o + i] != TZ_EVAL_WIN &&
                            eval_cmp(sp->c_tag[o + i], sp->c_bits[o + i], limit_tag, limit_bits) <= 0;
            weight[i] = ok ? (double)sp->c_visits[o + i] : 0.0;
            total += weight[i];
        }
        if (total <= 0.0) continue;  // WeightError::InsufficientNonZero -> select_best_action
        const double u01 = uni(sp->rng);
        if (sp->trace) sp->draws[g] = u01;
        // rand's WeightedIndex over the u32 weights: an integer drawn uniformly from [0, total), first cumulative weight above it
        const double u = std::floor(u01 * total);
        double acc = 0.0;
        int pick = nc - 1;
        for (int i = 0; i < nc; i++) {
            acc += weight[i];
            if (acc > u) {
                pick = i;
                break;
            }
        }
        out[g] = sp->c_moves[o + pick];
    }
    return TZ_OK;
}

// one symmetric Dir(alpha) sample per root, of dimension n_children (noise.rs:17-19), zero padded to `w`.
// ~100 gamma variates for each of thousands of roots: drawn by a fixed number of worker threads, each with its own
// generator seeded from the driver's stream (fixed count, so the result does not depend on the machine).
void dirichlet_rows(tz_selfplay* sp, int w) {
    constexpr int WORKERS = 8;
    sp->noise.assign((size_t)sp->B * w, 0.0f);
    uint64_t seeds[WORKERS];
    for (auto& x : seeds) x = sp->rng();
    auto work = [&](int t) {
        std::mt19937_64 rng(seeds[t]);
        std::gamma_distribution<double> gamma(NOISE_ALPHA, 1.0);
        std::vector<double> gsamp(w);
        const int lo = (int)((int64_t)sp->B * t / WORKERS), hi = (int)((int64_t)sp->B * (t + 1) / WORKERS);
        for (int g = lo; g < hi; g++) {
            const int nc = (int)sp->info[g].n_children;
            if (nc == 0) continue;
            double sum = 0.0;
            for (int i = 0; i < nc; i++) {
                gsamp[i] = gamma(rng);
                sum += gsamp[i];
            }
            if (!(sum > 0.0)) {  // every gamma underflowed: all mass on one child (what a tiny alpha tends to)
                gsamp[0] = 1.0;
                sum = 1.0;
            }
            for (int i = 0; i < nc; i++) sp->noise[(size_t)g * w + i] = (float)(gsamp[i] / sum);
        }
    };
    if (sp->B < 256) {
        for (int t = 0; t < WORKERS; t++) work(t);
        return;
    }
    std::thread threads[WORKERS];
    for (int t = 0; t < WORKERS; t++) threads[t] = std::thread(work, t);
    for (auto& th : threads) th.join();
}

int record(tz_selfplay* sp) {  // take_a_step up to the step itself (selfplay/src/main.rs:238-257)
    int w = 0, rc;
    if ((rc = fetch_children(sp, &w))) return rc;
    sp->history.emplace_back();
    MoveRecord& m = sp->history.back();
    const int B = sp->B;
    m.width = w;
    m.states.resize(B);
    if ((rc = TZS(get_positions)(sp->search, m.states.data()))) return rc;
    m.moves = sp->c_moves;
    m.pol.assign((size_t)B * w, 0.0f);
    m.nchild.resize(B);
    m.stepped.resize(B);
    m.actions = sp->actions;
    m.ube.resize(B);
    if (sp->kind == 0) {  // policy_target_from_proportional_visits (target.rs:151-164)
        for (int g = 0; g < B; g++) {
            const float denom = (float)std::max<uint32_t>(sp->info[g].visit_count, 1u);
            for (int i = 0; i < (int)sp->info[g].n_children; i++)
                m.pol[(size_t)g * w + i] = (float)sp->c_visits[(size_t)g * w + i] / denom;
        }
    } else if (sp->kind == 2) {  // uniform policy of the pre-training games (learn/src/main.rs:451-454)
        for (int g = 0; g < B; g++) {
            const float p = 1.0f / (float)std::max<uint32_t>(sp->info[g].n_children, 1u);
            for (int i = 0; i < (int)sp->info[g].n_children; i++) m.pol[(size_t)g * w + i] = p;
        }
    } else {  // improved_policy(IMPROVED_POLICY_VISITATIONS) (selfplay/src/main.rs:47-52, 246-250)
        int lg = 0;
        while ((1 << (lg + 1)) <= sp->k) lg++;
        const float visitations = (float)((sp->sims / lg / sp->k) * ((1 << lg) - 1));
        if ((rc = TZS(improved_policy)(sp->search, visitations, w, m.pol.data()))) return rc;
    }
    if (sp->kind == 2) {
        for (int g = 0; g < B; g++) m.ube[g] = 4.0f - 1.1920929e-07f;  // MAXIMUM_VARIANCE - f32::EPSILON (learn/src/main.rs:460)
    } else if ((rc = TZS(ube_target)(sp->search, BETA, m.ube.data()))) {
        return rc;
    }
    for (int g = 0; g < B; g++) {
        m.nchild[g] = sp->info[g].n_children;
        // batched.rs:137: a root that is terminal is not stepped
        m.stepped[g] = !(sp->info[g].eval_tag != TZ_EVAL_VALUE && sp->info[g].eval.ply == 0);
    }
    return TZ_OK;
}

int complete(tz_selfplay* sp) {  // restart_envs_and_complete_targets (selfplay/src/main.rs:263-329)
    const int B = sp->B;
    bool any = false;
    for (int g = 0; g < B; g++) any = any || sp->term[g] != TZ_TERMINAL_NONE;
    sp->t_states.clear();
    sp->t_moves.clear();
    sp->t_pol.clear();
    sp->t_value.clear();
    sp->t_ube.clear();
    sp->t_n.clear();
    if (any) {
        int rc;
        std::vector<tz_state> fresh(B);
        if ((rc = TZS(get_positions)(sp->search, fresh.data()))) return rc;
        if ((rc = TZS(terminal_details)(sp->search, sp->reason.data(), sp->winner.data()))) return rc;
        const int amax = sp->amax;
        char buf[256];
        std::vector<uint16_t> acts;
        for (int g = 0; g < B; g++) {
            if (sp->term[g] == TZ_TERMINAL_NONE) continue;
            uint8_t tag = sp->term[g] == TZ_TERMINAL_WIN ? TZ_EVAL_WIN : sp->term[g] == TZ_TERMINAL_LOSS ? TZ_EVAL_LOSS : TZ_EVAL_DRAW;
            uint32_t ply = 0;
            acts.clear();
            const int64_t first = sp->game_start[g] - sp->history_base;
            for (int64_t h = (int64_t)sp->history.size() - 1; h >= first; h--) {
                const MoveRecord& m = sp->history[(size_t)h];
                if (!m.stepped[g]) continue;
                tag = tag == TZ_EVAL_WIN ? TZ_EVAL_LOSS : tag == TZ_EVAL_LOSS ? TZ_EVAL_WIN : TZ_EVAL_DRAW;  // value.negate()
                ply++;
                acts.push_back(m.actions[g]);
                const tz_state& st = m.states[g];
                // "Only generate targets from non-exploratory episodes (or after the initial exploration)"
                if (sp->betas[g] == 0.0f || st.ply > WEIGHTED_RANDOM_PLIES) {
                    const int kk = (int)m.nchild[g];
                    sp->t_states.push_back(st);
                    const size_t off = sp->t_moves.size();
                    sp->t_moves.resize(off + amax, 0);
                    sp->t_pol.resize(off + amax, 0.0f);
                    memcpy(&sp->t_moves[off], &m.moves[(size_t)g * m.width], sizeof(uint16_t) * kk);
                    memcpy(&sp->t_pol[off], &m.pol[(size_t)g * m.width], sizeof(float) * kk);
                    sp->t_n.push_back(kk);
                    sp->t_value.push_back(eval_to_f32(tag, ply));
                    sp->t_ube.push_back(m.ube[g]);
                }
            }
            // Replay: start position, the moves in playing order, PTN result (target.rs:215-232)
            std::string line = "[TPS \"";
            if ((rc = tz_state_to_tps(&sp->start_states[g], buf, sizeof buf))) return rc;
            line += buf;
            line += "\"]";
            std::string expl = line;
            int count = 0;
            for (auto it = acts.rbegin(); it != acts.rend(); ++it, ++count) {
                if ((rc = tz_move_to_ptn(sp->n, *it, buf, sizeof buf))) return rc;
                line += ' ';
                line += buf;
                if (count < WEIGHTED_RANDOM_PLIES) {
                    expl += ' ';
                    expl += buf;
                }
            }
            line += ' ';
            line += result_string(sp->reason[g], sp->winner[g]);
            line += '\n';
            sp->replays_text += line;
            sp->n_replays++;
            if (sp->betas[g] > 0.0f) {  // feature "exploration": the opening of an exploratory game (:279-290)
                expl += '\n';
                sp->exploration_text += expl;
            }
            sp->game_start[g] = sp->moves_played + 1;
            sp->start_states[g] = fresh[g];
        }
        const int T = (int)sp->t_n.size();
        if (T > 0 && sp->has_exchange) {
            pack_records(sp->n, T, amax, sp->t_states, sp->t_moves, sp->t_pol, sp->t_n, sp->t_value, sp->t_ube, sp->records);
            sp->n_targets += (uint64_t)T;
        } else if (T > 0) {
            uint64_t total_moves = 0;
            for (int v : sp->t_n) total_moves += (uint64_t)v;
            std::vector<char> out((size_t)T * 200 + total_moves * 40);
            uint64_t written = 0;
            if ((rc = tz_format_targets(sp->n, T, sp->t_states.data(), sp->t_moves.data(), sp->t_pol.data(), sp->t_n.data(), amax,
                                        sp->t_value.data(), sp->t_ube.data(), out.data(), out.size(), &written)))
                return rc;
            sp->targets_text.append(out.data(), written);
            sp->n_targets += (uint64_t)T;
        }
    }
    // drop the records no running game refers to any more
    int64_t oldest = sp->moves_played + 1;
    for (int g = 0; g < B; g++) oldest = std::min(oldest, sp->game_start[g]);
    while (sp->history_base < oldest && !sp->history.empty()) {
        sp->history.pop_front();
        sp->history_base++;
    }
    return TZ_OK;
}

}  // namespace

extern "C" {

// search_kind: 0 = PUCT + Dirichlet (the north star's loop, selfplay/src/main.rs:127-136), 1 = Gumbel sequential
// halving with `sampled_actions` (:138-153), 2 = uniformly random legal moves (learn's pre-training games).
// exploration != 0: the first half of the games search with beta = 0.25 (cargo feature "exploration", :79-86).
int tz_selfplay_create(TZ_SEARCH_T* search, int sims_per_move, uint64_t seed, int shard, int search_kind, int sampled_actions,
                       int exploration, tz_selfplay** out) {
    if (!search || !out || search_kind < 0 || search_kind > 2 || sims_per_move < 0)
        return tz_fail(TZ_EINVAL, "tz_selfplay_create: bad argument");
    *out = nullptr;
    std::unique_ptr<tz_selfplay> sp(new tz_selfplay());
    sp->search = search;
    int rc = TZS(shape)(search, &sp->B, &sp->n, &sp->half_komi, &sp->amax);
    if (rc) return rc;
    sp->sims = sims_per_move;
    sp->kind = search_kind;
    sp->k = sampled_actions > 0 ? sampled_actions : 64;
    const int B = sp->B;
    sp->betas.assign(B, 0.0f);
    if (exploration)
        for (int g = 0; g < B / 2; g++) sp->betas[g] = BETA;
    std::seed_seq seq{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)shard, 0x7a6b5c4du};
    sp->rng.seed(seq);
    sp->game_start.assign(B, 0);
    sp->start_states.resize(B);
    sp->info.resize(B);
    sp->actions.resize(B);
    sp->best.resize(B);
    sp->choice.resize(B);
    sp->term.resize(B);
    sp->reason.resize(B);
    sp->winner.resize(B);
    std::uniform_int_distribution<int> open(0, 15);
    for (int g = 0; g < B; g++) sp->choice[g] = open(sp->rng);
    if ((rc = TZS(new_openings)(search, sp->choice.data()))) return rc;
    if ((rc = TZS(get_positions)(search, sp->start_states.data()))) return rc;
    *out = sp.release();
    return TZ_OK;
}

int tz_selfplay_destroy(tz_selfplay* sp) {
    delete sp;
    return TZ_OK;
}

// One outer-loop iteration of selfplay::main: search, pick the moves, record the targets-to-be, step, restart the
// finished games and complete their targets / replays (kept as text until taken).
int tz_selfplay_play_move(tz_selfplay* sp) {
    if (!sp) return tz_fail(TZ_EINVAL, "tz_selfplay_play_move: null handle");
    const int B = sp->B;
    int rc, w = 0;
    if (sp->trace) {
        sp->draws.assign(B, std::nan(""));
        sp->halving.assign(B, 0xFFFF);
    }
    if (sp->kind == 0) {
        if ((rc = TZS(simulate)(sp->search, sp->betas.data(), 1))) return rc;            // :128
        if ((rc = TZS(root_info)(sp->search, sp->info.data()))) return rc;
        w = 1;
        for (int g = 0; g < B; g++) w = std::max(w, (int)sp->info[g].n_children);
        dirichlet_rows(sp, w);
        if ((rc = TZS(apply_noise)(sp->search, sp->noise.data(), w, NOISE_RATIO))) return rc;  // :131
        if ((rc = TZS(simulate)(sp->search, sp->betas.data(), sp->sims))) return rc;        // :134-136
        if ((rc = select_actions_in_selfplay(sp, sp->actions))) return rc;                        // batched.rs:165-183
    } else if (sp->kind == 2) {
        if ((rc = TZS(simulate)(sp->search, sp->betas.data(), 1))) return rc;
        if ((rc = fetch_children(sp, &w))) return rc;
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        for (int g = 0; g < B; g++) {
            const int nc = (int)sp->info[g].n_children;
            const int j = nc > 0 ? std::min(nc - 1, (int)(uni(sp->rng) * nc)) : 0;
            sp->actions[g] = nc > 0 ? sp->c_moves[(size_t)g * w + j] : (uint16_t)0xFFFF;
        }
    } else {
        sp->gumbel.resize((size_t)B * sp->amax);
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        for (auto& x : sp->gumbel) {
            double u = uni(sp->rng);
            if (u <= 0.0) u = 1e-300;
            x = (float)(-std::log(-std::log(u)));  // Gumbel(0, 1)
        }
        if ((rc = TZS(gumbel_sh)(sp->search, sp->betas.data(), sp->k, sp->sims, sp->gumbel.data(), sp->amax, sp->actions.data())))
            return rc;                                                                           // :138-144
        if (sp->trace) sp->halving = sp->actions;
        if ((rc = TZS(root_info)(sp->search, sp->info.data()))) return rc;
        bool early = false;
        for (int g = 0; g < B; g++) early = early || sp->info[g].ply < WEIGHTED_RANDOM_PLIES;
        if (early) {                                                                             // :145-153
            std::vector<uint16_t> sampled;
            std::vector<uint16_t> ply(B);
            for (int g = 0; g < B; g++) ply[g] = sp->info[g].ply;
            if ((rc = select_actions_in_selfplay(sp, sampled))) return rc;
            for (int g = 0; g < B; g++)
                if (ply[g] < WEIGHTED_RANDOM_PLIES) sp->actions[g] = sampled[g];
        }
    }
    if (sp->trace && sp->trace->chosen) sp->trace->chosen(sp->draws, sp->halving, sp->actions);
    if ((rc = record(sp))) return rc;
    if ((rc = TZS(step)(sp->search, sp->actions.data()))) return rc;                        // take_a_step
    std::uniform_int_distribution<int> open(0, 15);
    for (int g = 0; g < B; g++) sp->choice[g] = open(sp->rng);
    if ((rc = TZS(restart_terminal)(sp->search, sp->choice.data(), sp->term.data()))) return rc;
    if ((rc = complete(sp))) return rc;
    if (sp->trace && sp->trace->completed)
        sp->trace->completed(sp->term, sp->t_states, sp->t_moves, sp->t_pol, sp->t_n, sp->t_value, sp->t_ube, sp->amax);
    sp->moves_played++;
    sp->positions += (uint64_t)B;
    return TZ_OK;
}

// counters since creation: moves played, targets and replays completed
int tz_selfplay_counters(tz_selfplay* sp, uint64_t* moves_out, uint64_t* targets_out, uint64_t* replays_out) {
    if (!sp) return tz_fail(TZ_EINVAL, "tz_selfplay_counters: null handle");
    if (moves_out) *moves_out = (uint64_t)sp->moves_played;
    if (targets_out) *targets_out = sp->n_targets;
    if (replays_out) *replays_out = sp->n_replays;
    return TZ_OK;
}

// Moves the text accumulated since the last call out of the driver: which = 0 target lines, 1 replay lines,
// 2 exploration replay lines.  *size_out = bytes available; copied (and cleared) only if cap is large enough.
int tz_selfplay_take_text(tz_selfplay* sp, int which, char* out, uint64_t cap, uint64_t* size_out) {
    if (!sp || which < 0 || which > 2 || !size_out) return tz_fail(TZ_EINVAL, "tz_selfplay_take_text: bad argument");
    std::string& s = which == 0 ? sp->targets_text : which == 1 ? sp->replays_text : sp->exploration_text;
    *size_out = s.size();
    if (!out || cap < s.size()) return s.empty() ? TZ_OK : TZ_ECAPACITY;
    memcpy(out, s.data(), s.size());
    s.clear();
    return TZ_OK;
}

// selfplay::main on a directory (selfplay/src/main.rs:88-205) for `moves` iterations (< 0: forever): wait while
// buffer_lengths.txt says learn has more than `max_buffer_len` self-play targets (:90-105, 371-387), call `reload`
// (Net::load of model_latest, :107-121; may be null), play a move, append targets-selfplay<suffix>.txt /
// replays<suffix>.txt / replays-exploration<suffix>.txt from a writer thread.  wait_limit_s < 0: wait forever.
int tz_selfplay_set_exchange(tz_selfplay* sp, const HostExchange* x) {
    if (!sp) return tz_fail(TZ_EINVAL, "tz_selfplay_set_exchange: null handle");
    sp->has_exchange = x != nullptr && x->world > 1;
    if (x) sp->xch = *x;
    return TZ_OK;
}

// The hand-over of one move between the shards (collective: every rank calls it once per tz_selfplay_play_move): all-gather
// of the packed targets, of the replay lines and of the exploration lines; afterwards the writer rank (every rank when
// writer < 0) holds everybody's as text, in rank order, for tz_selfplay_take_text / the run loop; the other ranks hold none.
int tz_selfplay_exchange(tz_selfplay* sp) {
    if (!sp) return tz_fail(TZ_EINVAL, "tz_selfplay_exchange: null handle");
    if (!sp->has_exchange) return TZ_OK;
    const bool keep = sp->xch.writer < 0 || sp->xch.writer == sp->xch.rank;
    std::vector<std::vector<unsigned char>> all;
    int rc = sp->xch.all_gather(sp->records, all);
    sp->records.clear();
    if (rc) return rc;
    if (keep) {
        std::vector<tz_state> st;
        std::vector<uint16_t> moves;
        std::vector<float> pol, value, ube;
        std::vector<int32_t> cnt;
        for (auto& blob : all)
            if ((rc = unpack_records(blob, sp->amax, st, moves, pol, cnt, value, ube))) return rc;
        const int T = (int)cnt.size();
        if (T > 0) {
            uint64_t total_moves = 0;
            for (int v : cnt) total_moves += (uint64_t)v;
            std::vector<char> out((size_t)T * 200 + total_moves * 40);
            uint64_t written = 0;
            if ((rc = tz_format_targets(sp->n, T, st.data(), moves.data(), pol.data(), cnt.data(), sp->amax, value.data(), ube.data(), out.data(),
                                        out.size(), &written)))
                return rc;
            sp->targets_text.append(out.data(), written);
            sp->n_gathered += (uint64_t)T;
        }
    }
    for (std::string* text : {&sp->replays_text, &sp->exploration_text}) {
        std::vector<unsigned char> mine(text->begin(), text->end());
        if ((rc = sp->xch.all_gather(mine, all))) return rc;
        text->clear();
        if (keep)
            for (auto& blob : all) text->append(blob.begin(), blob.end());
    }
    return TZ_OK;
}

int tz_selfplay_run(tz_selfplay* sp, const char* directory, int moves, int max_buffer_len, const char* suffix,
                    int (*reload)(void*), void* reload_user, double wait_limit_s) {
    if (!sp || !directory) return tz_fail(TZ_EINVAL, "tz_selfplay_run: bad argument");
    const std::string dir = directory, suf = suffix ? suffix : "";
    struct Job {
        std::string targets, replays, expl;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> queue;
    bool done = false;
    std::string write_error;
    auto append = [&](const std::string& name, const std::string& text) {
        if (!append_whole(dir + "/" + name + suf + ".txt", text)) write_error = "cannot append to " + name;
    };
    std::thread writer([&] {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                job = std::move(queue.front());
                queue.pop_front();
            }
            append("targets-selfplay", job.targets);
            append("replays", job.replays);
            append("replays-exploration", job.expl);
        }
    });
    auto finish = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        writer.join();
    };
    int rc = TZ_OK;
    for (int step = 0; moves < 0 || step < moves; step++) {
        // What this rank found before the move: its own reasons to stop.  With N shards nobody leaves on its own — a rank that
        // walked out here would strand the others in the move's all-gather for good (RCCL has no timeout) — so the ranks first
        // exchange one status word and stop together, every collective of the round matched.
        int local = TZ_OK;
        std::string local_msg;
        double waited = 0.0;
        for (;;) {  // the inner `loop` of selfplay::main
            long a = -1, b = -1, c = -1;
            FILE* f = fopen((dir + "/buffer_lengths.txt").c_str(), "rb");
            const bool ok = f && fscanf(f, "%ld,%ld,%ld", &a, &b, &c) == 3 && a + b == c;
            if (f) fclose(f);
            if (ok && a <= max_buffer_len) break;
            if (wait_limit_s >= 0 && waited >= wait_limit_s) {
                local = TZ_ESTATE;
                local_msg = ok ? "tz_selfplay_run: learn's buffer stayed over its cap" : "tz_selfplay_run: buffer_lengths.txt unreadable";
                break;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(wait_limit_s >= 0 && wait_limit_s < 1 ? 10 : 1000));
            waited += wait_limit_s >= 0 && wait_limit_s < 1 ? 0.01 : 1.0;
        }
        if (reload) {   // called on every rank even when this one is about to stop: the callback may hold a collective (tz_net_broadcast)
            const int r = reload(reload_user);
            if (r && !local) {
                local = r;
                local_msg = tz_last_error();
            }
        }
        if (sp->has_exchange && sp->xch.world > 1) {
            std::vector<unsigned char> mine(sizeof(int32_t));
            const int32_t word = local;
            memcpy(mine.data(), &word, sizeof word);
            std::vector<std::vector<unsigned char>> all;
            if ((rc = sp->xch.all_gather(mine, all))) break;
            for (size_t r = 0; r < all.size() && !rc; r++) {
                int32_t theirs = 0;
                if (all[r].size() == sizeof theirs) memcpy(&theirs, all[r].data(), sizeof theirs);
                if (theirs) rc = local ? tz_fail(local, local_msg) : tz_fail(theirs, "tz_selfplay_run: rank " + std::to_string(r) + " stopped before the move");
            }
            if (rc) break;
        } else if (local) {
            rc = tz_fail(local, local_msg);
            break;
        }
        if ((rc = tz_selfplay_play_move(sp))) break;
        if ((rc = tz_selfplay_exchange(sp))) break;   // N shards: the writer rank appends everybody's lines
        if (sp->has_exchange && sp->xch.writer < 0 && sp->xch.rank != 0) {   // every rank holds them: one copy goes to the files
            sp->targets_text.clear();
            sp->replays_text.clear();
            sp->exploration_text.clear();
        }
        Job job;
        job.targets.swap(sp->targets_text);
        job.replays.swap(sp->replays_text);
        job.expl.swap(sp->exploration_text);
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(std::move(job));
        }
        cv.notify_one();
    }
    finish();
    if (!rc && !write_error.empty()) return tz_fail(TZ_ESTATE, "tz_selfplay_run: " + write_error);
    return rc;
}

}  // extern "C"

// =================================================================================================
// reanalyze::main above the search (reanalyze/src/main.rs:60-290): the position buffer fed from replays.txt
// (fill_buffer_with_positions_from_replays, :270-290; Replay::from_str / Replay::states, target.rs:205-268), sampling
// without replacement (:154-158), fresh trees on overwritten envs (:159-165), search, and the target of every position
// (:179-228) as text lines.
namespace {
constexpr int MIN_POSITIONS = 4000 * 128 / 4;  // reanalyze/src/main.rs:38
constexpr int MAX_REANALYZE_BUFFER_LEN = 32000;  // :40
}  // namespace

struct tz_reanalyze {
    TZ_SEARCH_T* search = nullptr;
    int B = 0, n = 0, half_komi = 0, amax = 0;
    int sims = 0, kind = 0, k = 64, rank = 0, world = 1;
    std::mt19937_64 rng;
    std::vector<tz_state> positions;
    uint64_t offset = 0, line_no = 0;
    std::string targets_text;
    uint64_t n_targets = 0;
    const HostTrace* trace = nullptr;   // tests: observer of the targets (tz_host_exchange.h)
};

// tests only (not in the C ABI): observers of the two drivers' decisions
void tz_selfplay_set_trace(tz_selfplay* sp, const HostTrace* t) { sp->trace = t; }
void tz_reanalyze_set_trace(tz_reanalyze* ra, const HostTrace* t) { ra->trace = t; }

namespace {

struct ParsedReplay {
    tz_state start;
    std::vector<uint16_t> moves;
};

bool parse_replay_line(const char* b, const char* e, int n, int half_komi, ParsedReplay& out) {
    static const char prefix[] = "[TPS \"";
    if (e - b < 8 || memcmp(b, prefix, 6) != 0) return false;
    const char* q = b + 6;
    const char* close = nullptr;
    for (const char* p = q; p + 1 < e; p++)
        if (p[0] == '"' && p[1] == ']') {
            close = p;
            break;
        }
    if (!close) return false;
    const std::string tps(q, close);
    if (tz_state_from_tps(tps.c_str(), n, half_komi, &out.start)) return false;
    out.moves.clear();
    const char* p = close + 2;
    while (p < e) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
        const char* t = p;
        while (p < e && *p != ' ' && *p != '\t' && *p != '\r') p++;
        if (p == t) break;
        const std::string tok(t, p);
        if (tok == "R-0" || tok == "0-R" || tok == "F-0" || tok == "0-F" || tok == "1/2-1/2" || tok == "1-0" || tok == "0-1") break;
        if (tok.back() == '.' && tok.find_first_not_of("0123456789") == tok.size() - 1) continue;  // move numbers
        uint16_t mv;
        if (tz_move_from_ptn(n, tok.c_str(), &mv)) return false;
        out.moves.push_back(mv);
    }
    return true;
}

// every pre-move state of every replay, validated on the device; a replay with an illegal move is dropped whole
// (the reference's line fails to parse, target.rs:248-268)
int expand_replays(tz_reanalyze* ra, const std::vector<ParsedReplay>& replays, uint64_t* added_out) {
    const int B = ra->B;
    std::vector<int32_t> idx(B);
    std::vector<tz_state> states(B), cur(B);
    std::vector<uint16_t> acts(B);
    std::vector<int8_t> ok(B);
    uint64_t added = 0;
    for (size_t base = 0; base < replays.size(); base += (size_t)B) {
        const int cnt = (int)std::min<size_t>(B, replays.size() - base);
        for (int g = 0; g < cnt; g++) {
            idx[g] = g;
            states[g] = replays[base + g].start;
        }
        int rc = TZS(set_positions)(ra->search, cnt, idx.data(), states.data());
        if (rc) return rc;
        std::vector<std::vector<tz_state>> per_game(cnt);
        std::vector<char> alive(B, 0), dropped(cnt, 0);
        for (int g = 0; g < cnt; g++) alive[g] = 1;
        for (size_t ply = 0;; ply++) {
            bool any = false;
            for (int g = 0; g < B; g++) {
                acts[g] = 0xFFFF;
                if (g < cnt && alive[g] && ply < replays[base + g].moves.size()) {
                    acts[g] = replays[base + g].moves[ply];
                    any = true;
                } else if (g < cnt) {
                    alive[g] = 0;
                }
            }
            if (!any) break;
            if ((rc = TZS(get_positions)(ra->search, cur.data()))) return rc;
            if ((rc = TZS(play_moves)(ra->search, acts.data(), ok.data()))) return rc;
            for (int g = 0; g < cnt; g++) {
                if (!alive[g]) continue;
                if (ok[g] == 1) {
                    per_game[g].push_back(cur[g]);
                } else {
                    if (ok[g] == 0) dropped[g] = 1;  // illegal move: the whole line is rejected
                    alive[g] = 0;
                }
            }
        }
        for (int g = 0; g < cnt; g++) {
            if (dropped[g]) continue;
            ra->positions.insert(ra->positions.end(), per_game[g].begin(), per_game[g].end());
            added += per_game[g].size();
        }
    }
    if (added_out) *added_out = added;
    return TZ_OK;
}

}  // namespace

extern "C" {

// search_kind: 0 = `sims` PUCT simulations per position (SURVEY config 5), 1 = Gumbel sequential halving with budget
// `sims` and `sampled_actions` (reanalyze/src/main.rs:171-177).  rank / world: replay line i belongs to rank i % world.
int tz_reanalyze_create(TZ_SEARCH_T* search, int sims, uint64_t seed, int rank, int world, int search_kind, int sampled_actions,
                        tz_reanalyze** out) {
    if (!search || !out || sims <= 0 || world <= 0 || rank < 0 || rank >= world || search_kind < 0 || search_kind > 1)
        return tz_fail(TZ_EINVAL, "tz_reanalyze_create: bad argument");
    *out = nullptr;
    std::unique_ptr<tz_reanalyze> ra(new tz_reanalyze());
    ra->search = search;
    int rc = TZS(shape)(search, &ra->B, &ra->n, &ra->half_komi, &ra->amax);
    if (rc) return rc;
    ra->sims = sims;
    ra->kind = search_kind;
    ra->k = sampled_actions > 0 ? sampled_actions : 64;
    ra->rank = rank;
    ra->world = world;
    std::seed_seq seq{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)rank, 0x2e7a11u};
    ra->rng.seed(seq);
    *out = ra.release();
    return TZ_OK;
}

int tz_reanalyze_destroy(tz_reanalyze* ra) {
    delete ra;
    return TZ_OK;
}

// fill_buffer_with_positions_from_replays: reads what was appended to `path` since the last call (complete lines only).
int tz_reanalyze_feed(tz_reanalyze* ra, const char* path, uint64_t* added_out, uint64_t* total_out) {
    if (!ra || !path) return tz_fail(TZ_EINVAL, "tz_reanalyze_feed: bad argument");
    if (added_out) *added_out = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return tz_fail(TZ_EPARSE, std::string("tz_reanalyze_feed: cannot open ") + path);
    std::string data;
    if (fseek(f, (long)ra->offset, SEEK_SET) == 0) {
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
    }
    fclose(f);
    const size_t last_nl = data.rfind('\n');
    std::vector<ParsedReplay> replays;
    if (last_nl != std::string::npos) {
        ra->offset += last_nl + 1;
        size_t pos = 0;
        while (pos <= last_nl) {
            const size_t nl = data.find('\n', pos);
            const bool mine = (int)(ra->line_no % (uint64_t)ra->world) == ra->rank;
            ra->line_no++;
            if (mine && nl > pos) {
                ParsedReplay r;
                if (parse_replay_line(data.data() + pos, data.data() + nl, ra->n, ra->half_komi, r)) replays.push_back(std::move(r));
            }
            pos = nl + 1;
        }
    }
    int rc = expand_replays(ra, replays, added_out);
    if (total_out) *total_out = ra->positions.size();
    return rc;
}

// One outer-loop iteration (reanalyze/src/main.rs:146-228): B positions sampled without replacement, fresh trees,
// search, one target line per position.  TZ_ESTATE if the buffer holds fewer than B positions.
int tz_reanalyze_iterate(tz_reanalyze* ra) {
    if (!ra) return tz_fail(TZ_EINVAL, "tz_reanalyze_iterate: null handle");
    const int B = ra->B;
    const size_t P = ra->positions.size();
    if (P < (size_t)B) return tz_fail(TZ_ESTATE, "tz_reanalyze_iterate: fewer positions than the batch size");
    // position_buffer.sample(rng, B): partial Fisher-Yates over the buffer itself (the order of the buffer is immaterial)
    std::vector<tz_state> states(B);
    for (int i = 0; i < B; i++) {
        std::uniform_int_distribution<size_t> pick(i, P - 1);
        std::swap(ra->positions[i], ra->positions[pick(ra->rng)]);
        states[i] = ra->positions[i];
    }
    std::vector<int32_t> idx(B);
    for (int g = 0; g < B; g++) idx[g] = g;
    int rc = TZS(set_positions)(ra->search, B, idx.data(), states.data());
    if (rc) return rc;
    std::vector<float> zero_beta(B, 0.0f);
    std::vector<uint16_t> selected(B);
    if (ra->kind == 0) {
        if ((rc = TZS(simulate)(ra->search, zero_beta.data(), ra->sims))) return rc;
        if ((rc = TZS(select_best_actions)(ra->search, selected.data()))) return rc;
    } else {
        std::vector<float> gumbel((size_t)B * ra->amax);
        std::uniform_real_distribution<double> uni(0.0, 1.0);
        for (auto& x : gumbel) {
            double u = uni(ra->rng);
            if (u <= 0.0) u = 1e-300;
            x = (float)(-std::log(-std::log(u)));
        }
        if ((rc = TZS(gumbel_sh)(ra->search, zero_beta.data(), ra->k, ra->sims, gumbel.data(), ra->amax, selected.data()))) return rc;
    }
    std::vector<tz_root_info> info(B);
    if ((rc = TZS(root_info)(ra->search, info.data()))) return rc;
    int w = 1;
    for (int g = 0; g < B; g++) w = std::max(w, (int)info[g].n_children);
    const size_t cells = (size_t)B * w;
    std::vector<uint16_t> moves(cells);
    std::vector<uint32_t> visits(cells), bits(cells);
    std::vector<uint8_t> tag(cells);
    if ((rc = TZS(root_children)(ra->search, w, moves.data(), visits.data(), tag.data(), bits.data(), nullptr, nullptr, nullptr)))
        return rc;
    std::vector<float> mvc(B), pol(cells), ube(B), value(B);
    for (int g = 0; g < B; g++) {  // most_visited_count (node/mod.rs:209-213)
        uint32_t m = 0;
        for (int i = 0; i < (int)info[g].n_children; i++) m = std::max(m, visits[(size_t)g * w + i]);
        mvc[g] = (float)m;
    }
    if ((rc = TZS(improved_policy_each)(ra->search, mvc.data(), w, pol.data()))) return rc;  // :196-202
    if ((rc = TZS(ube_target)(ra->search, BETA, ube.data()))) return rc;                     // :203
    std::vector<int32_t> nm(B);
    std::vector<uint16_t> mv_pad((size_t)B * ra->amax, 0);
    std::vector<float> pol_pad((size_t)B * ra->amax, 0.0f);
    uint64_t total_moves = 0;
    for (int g = 0; g < B; g++) {
        const int nc = (int)info[g].n_children;
        nm[g] = nc;
        total_moves += (uint64_t)nc;
        memcpy(&mv_pad[(size_t)g * ra->amax], &moves[(size_t)g * w], sizeof(uint16_t) * nc);
        memcpy(&pol_pad[(size_t)g * ra->amax], &pol[(size_t)g * w], sizeof(float) * nc);
        if (info[g].eval_tag != TZ_EVAL_VALUE) {  // solved root: its own evaluation (:184-187)
            value[g] = eval_to_f32(info[g].eval_tag, info[g].eval.ply);
        } else {                                   // the selected child's evaluation, negated, then converted (:188-195):
            float v = 0.0f;                        // Eval::negate flips a proven result AND adds a ply (eval.rs:40-47) before
            for (int i = 0; i < nc; i++)           // f32::from discounts it, so Win(p) of the child gives -0.997^(p+1)
                if (moves[(size_t)g * w + i] == selected[g]) {
                    const size_t o = (size_t)g * w + i;
                    if (tag[o] == TZ_EVAL_VALUE) {
                        v = -tz_bits_to_float(bits[o]);
                    } else {
                        const uint8_t flipped = tag[o] == TZ_EVAL_WIN ? TZ_EVAL_LOSS : tag[o] == TZ_EVAL_LOSS ? TZ_EVAL_WIN : TZ_EVAL_DRAW;
                        v = eval_to_f32(flipped, bits[o] + 1);
                    }
                    break;
                }
            value[g] = v;
        }
    }
    if (ra->trace && ra->trace->reanalyzed) ra->trace->reanalyzed(selected, mv_pad, pol_pad, nm, value, ube, ra->amax);
    std::vector<char> out((size_t)B * 200 + total_moves * 40);
    uint64_t written = 0;
    if ((rc = tz_format_targets(ra->n, B, states.data(), mv_pad.data(), pol_pad.data(), nm.data(), ra->amax, value.data(), ube.data(),
                                out.data(), out.size(), &written)))
        return rc;
    ra->targets_text.append(out.data(), written);
    ra->n_targets += (uint64_t)B;
    return TZ_OK;
}

int tz_reanalyze_take_text(tz_reanalyze* ra, char* out, uint64_t cap, uint64_t* size_out) {
    if (!ra || !size_out) return tz_fail(TZ_EINVAL, "tz_reanalyze_take_text: bad argument");
    *size_out = ra->targets_text.size();
    if (!out || cap < ra->targets_text.size()) return ra->targets_text.empty() ? TZ_OK : TZ_ECAPACITY;
    memcpy(out, ra->targets_text.data(), ra->targets_text.size());
    ra->targets_text.clear();
    return TZ_OK;
}

// reanalyze::main on a directory for `iterations` outer iterations (< 0: forever).  min_positions <= 0: the
// reference's 128 000; wait_limit_s < 0: wait forever.
int tz_reanalyze_run(tz_reanalyze* ra, const char* directory, int iterations, int min_positions, const char* suffix,
                     int (*reload)(void*), void* reload_user, double wait_limit_s) {
    if (!ra || !directory) return tz_fail(TZ_EINVAL, "tz_reanalyze_run: bad argument");
    const std::string dir = directory, suf = suffix ? suffix : "";
    const int need = std::max(min_positions > 0 ? min_positions : MIN_POSITIONS, ra->B);
    const bool fast = wait_limit_s >= 0 && wait_limit_s < 5;
    double waited = 0.0;
    auto nap = [&]() {
        std::this_thread::sleep_for(std::chrono::milliseconds(fast ? 10 : 1000));
        waited += fast ? 0.01 : 1.0;
    };
    int rc = TZ_OK;
    for (int it = 0; iterations < 0 || it < iterations;) {
        long a = -1, b = -1, c = -1;
        FILE* f = fopen((dir + "/buffer_lengths.txt").c_str(), "rb");
        const bool ok = f && fscanf(f, "%ld,%ld,%ld", &a, &b, &c) == 3 && a + b == c;
        if (f) fclose(f);
        if (!ok || b > MAX_REANALYZE_BUFFER_LEN) {
            if (wait_limit_s >= 0 && waited >= wait_limit_s)
                return tz_fail(TZ_ESTATE, ok ? "tz_reanalyze_run: learn's buffer stayed over its cap" : "tz_reanalyze_run: buffer_lengths.txt unreadable");
            nap();
            continue;
        }
        if (reload && (rc = reload(reload_user))) return rc;
        uint64_t added = 0, total = 0;
        (void)tz_reanalyze_feed(ra, (dir + "/replays.txt").c_str(), &added, &total);  // "Cannot fill position buffer": keep going
        if ((int64_t)ra->positions.size() < need) {
            if (wait_limit_s >= 0 && waited >= wait_limit_s) return tz_fail(TZ_ESTATE, "tz_reanalyze_run: not enough positions yet");
            nap();  // the reference sleeps 60 s here (:135-142)
            continue;
        }
        if ((rc = tz_reanalyze_iterate(ra))) return rc;
        if (!append_whole(dir + "/targets-reanalyze" + suf + ".txt", ra->targets_text))
            return tz_fail(TZ_ESTATE, "tz_reanalyze_run: cannot append targets-reanalyze");
        ra->targets_text.clear();
        it++;
    }
    return rc;
}

}  // extern "C"

// =================================================================================================
// The two other consumers of the search ABI, natively.
extern "C" {

// evaluation::compete (evaluation/src/main.rs:224-319): two searches (one per side, usually two networks) over the
// same `games`; the side to move picks its move by Gumbel sequential halving and BOTH trees are stepped with it.
// result_out[3] = wins, losses, draws from White's point of view.
int tz_compete(TZ_SEARCH_T* white, TZ_SEARCH_T* black, const tz_state* games, float white_beta, float black_beta, uint64_t seed,
               int sampled_actions, int search_budget, int max_moves, int32_t* result_out) {
    if (!white || !black || !games || !result_out) return tz_fail(TZ_EINVAL, "tz_compete: null argument");
    int B = 0, Bb = 0, n = 0, nb = 0, hk = 0, amax = 0;
    int rc = TZS(shape)(white, &B, &n, &hk, &amax);
    if (rc) return rc;
    if ((rc = TZS(shape)(black, &Bb, &nb, nullptr, nullptr))) return rc;
    if (B != Bb || n != nb) return tz_fail(TZ_EINVAL, "tz_compete: the two searches differ in batch or board size");
    std::seed_seq seq{(uint32_t)seed, (uint32_t)(seed >> 32), 0xc0de7e57u};
    std::mt19937_64 rng(seq);
    std::vector<int32_t> idx(B), choice(B);
    for (int g = 0; g < B; g++) idx[g] = g;
    if ((rc = TZS(set_positions)(white, B, idx.data(), games))) return rc;   // BatchedMCTS::from_envs(games) x2 (:240-241)
    if ((rc = TZS(set_positions)(black, B, idx.data(), games))) return rc;
    std::vector<float> beta_w(B, white_beta), beta_b(B, black_beta), gumbel((size_t)B * amax);
    std::vector<uint16_t> top(B);
    std::vector<int8_t> term(B);
    std::vector<char> done(B, 0);
    std::vector<tz_state> cur_states(B), moved;
    std::vector<int32_t> didx;
    int wins = 0, losses = 0, draws = 0;
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    std::uniform_int_distribution<int> open(0, 15);
    for (int mv = 0; mv < max_moves; mv++) {
        for (int side = 0; side < 2; side++) {
            bool all = true;
            for (int g = 0; g < B; g++) all = all && done[g];
            if (all) goto finished;
            {
                const bool is_white = side == 0;
                TZ_SEARCH_T* cur = is_white ? white : black;
                TZ_SEARCH_T* oth = is_white ? black : white;
                for (auto& x : gumbel) {
                    double u = uni(rng);
                    if (u <= 0.0) u = 1e-300;
                    x = (float)(-std::log(-std::log(u)));
                }
                if ((rc = TZS(gumbel_sh)(cur, (is_white ? beta_w : beta_b).data(), sampled_actions, search_budget, gumbel.data(), amax,
                                        top.data())))
                    return rc;                                               // :257-273
                if ((rc = TZS(step)(cur, top.data()))) return rc;            // :276-277
                if ((rc = TZS(step)(oth, top.data()))) return rc;
                for (int g = 0; g < B; g++) choice[g] = open(rng);
                if ((rc = TZS(restart_terminal)(cur, choice.data(), term.data()))) return rc;  // :280-288
                bool any_done = false;
                for (int g = 0; g < B; g++) {
                    if (term[g] != TZ_TERMINAL_NONE && !done[g]) {
                        // seen after the move: a Loss for the side to move is a win for the mover (:306-313)
                        if (term[g] == TZ_TERMINAL_DRAW) draws++;
                        else if ((term[g] == TZ_TERMINAL_LOSS) == is_white) wins++;
                        else losses++;
                    }
                    if (term[g] != TZ_TERMINAL_NONE) done[g] = 1;
                    any_done = any_done || done[g];
                }
                if (any_done) {  // the other side's nodes and envs of finished games are reset too (:290-299)
                    if ((rc = TZS(get_positions)(cur, cur_states.data()))) return rc;
                    didx.clear();
                    moved.clear();
                    for (int g = 0; g < B; g++)
                        if (done[g]) {
                            didx.push_back(g);
                            moved.push_back(cur_states[g]);
                        }
                    if ((rc = TZS(set_positions)(oth, (int)didx.size(), didx.data(), moved.data()))) return rc;
                }
            }
        }
    }
finished:
    result_out[0] = wins;
    result_out[1] = losses;
    result_out[2] = draws;
    return TZ_OK;
}

// puzzle `benchmark` (puzzle/src/main.rs:168-269): `count` positions solved in batches with Gumbel sequential halving at
// beta 0; solved = select_best_action equals the solution; proven = root solved to a win (win != 0) or all children but
// one solved as wins.  result_out[3] = attempted, solved, proven.
int tz_puzzle_benchmark(TZ_SEARCH_T* search, const tz_state* puzzles, const uint16_t* solutions, int count, int win, uint64_t seed,
                        int sampled_actions, int search_budget, int32_t* result_out) {
    if (!search || !puzzles || !solutions || !result_out || count < 0) return tz_fail(TZ_EINVAL, "tz_puzzle_benchmark: bad argument");
    int B = 0, n = 0, hk = 0, amax = 0;
    int rc = TZS(shape)(search, &B, &n, &hk, &amax);
    if (rc) return rc;
    std::seed_seq seq{(uint32_t)seed, (uint32_t)(seed >> 32), 0x9a221eu};
    std::mt19937_64 rng(seq);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    std::vector<int32_t> idx(B);
    for (int g = 0; g < B; g++) idx[g] = g;
    std::vector<tz_state> states(B);
    std::vector<float> zero_beta(B, 0.0f), gumbel((size_t)B * amax);
    std::vector<uint16_t> selected(B), ignored(B);
    std::vector<tz_root_info> info(B);
    int attempted = 0, solved = 0, proven = 0;
    for (int lo = 0; lo < count; lo += B) {
        const int kk = std::min(B, count - lo);
        if ((rc = TZS(get_positions)(search, states.data()))) return rc;  // a short last batch keeps the previous tail (:198-204)
        for (int g = 0; g < kk; g++) states[g] = puzzles[lo + g];
        if ((rc = TZS(set_positions)(search, B, idx.data(), states.data()))) return rc;  // every node reset (:195-197)
        for (auto& x : gumbel) {
            double u = uni(rng);
            if (u <= 0.0) u = 1e-300;
            x = (float)(-std::log(-std::log(u)));
        }
        if ((rc = TZS(gumbel_sh)(search, zero_beta.data(), sampled_actions, search_budget, gumbel.data(), amax, ignored.data()))) return rc;
        if ((rc = TZS(select_best_actions)(search, selected.data()))) return rc;         // :215
        if ((rc = TZS(root_info)(search, info.data()))) return rc;
        attempted += kk;
        for (int g = 0; g < kk; g++) solved += selected[g] == solutions[lo + g];
        if (win) {
            for (int g = 0; g < kk; g++) proven += info[g].eval_tag == TZ_EVAL_WIN;      // :237-243
        } else {
            int w = 1;
            for (int g = 0; g < B; g++) w = std::max(w, (int)info[g].n_children);
            std::vector<uint8_t> tag((size_t)B * w);
            if ((rc = TZS(root_children)(search, w, nullptr, nullptr, tag.data(), nullptr, nullptr, nullptr, nullptr))) return rc;
            for (int g = 0; g < kk; g++) {                                               // :244-258
                int wins = 0;
                for (int i = 0; i < (int)info[g].n_children; i++) wins += tag[(size_t)g * w + i] == TZ_EVAL_WIN;
                proven += wins == (int)info[g].n_children - 1;
            }
        }
    }
    result_out[0] = attempted;
    result_out[1] = solved;
    result_out[2] = proven;
    return TZ_OK;
}

}  // extern "C"
