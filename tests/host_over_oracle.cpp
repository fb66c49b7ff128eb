// Test harness: the native host drivers (takzero_amd/csrc/tz_host.cpp: selfplay::main and reanalyze::main) bound to
// the CPU oracle's search (oracle/capi.cpp exposes the same surface under tzo_*) instead of the HIP engine.  Same
// driver code, same seed => the text it produces over the oracle must equal, byte for byte, what it produces over the
// GPU engine (tests/test_gpu_native_driver.py); and being pure CPU code it also runs under ASan / UBSan.
//   host_over_oracle <n> <half_komi> <agent 1|2> <batch> <kind 0|1|2> <sims> <k> <exploration> <moves> <seed> <out prefix>
// Environment: TZH_SHARD = the shard index of the driver's random stream (default 0); TZH_COMM_DIR + TZH_RANK + TZH_WORLD
// (+ TZH_WRITER, default 0) = N such processes exchanging through the "fs" transport of csrc/tz_comm.cpp after every move
// (only the self-play part runs then); TZH_MARK = 1 writes a "#move" line after every move into the dumps.
// With -DTZ_HARNESS_WITH_NET (GPU box only, linked against libtakzero_hip.so) agent 0 = the HIP network called through
// tz_net_eval as the oracle search's Agent; four more arguments: <model.tzw> <arch> <blocks> <precision>.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "takzero_hip.h"

struct tzo_search;
typedef void (*tzo_agent_fn)(void*, int, const tz_state*, const uint16_t*, const int32_t*, int, float*, float*, float*);
extern "C" {
tzo_search* tzo_search_create(int agent_kind, tzo_agent_fn fn, void* user, int batch, int n, int half_komi);
void tzo_search_destroy(tzo_search* s);
int tzo_search_shape(tzo_search* s, int*, int*, int*, int*);
int tzo_search_set_positions(tzo_search* s, int count, const int32_t* game_idx, const tz_state* states);
int tzo_search_get_positions(tzo_search* s, tz_state* out);
int tzo_search_new_openings(tzo_search* s, const int32_t* choice);
int tzo_search_simulate(tzo_search* s, const float* betas, int n_sims);
int tzo_search_apply_noise(tzo_search* s, const float* noise, int amax, float ratio);
int tzo_search_root_info(tzo_search* s, tz_root_info* out);
int tzo_search_root_children(tzo_search* s, int amax, uint16_t* move_idx, uint32_t* visits, uint8_t* eval_tag, uint32_t* eval_bits,
                             float* logit, float* prob, float* std_dev);
int tzo_search_select_best_actions(tzo_search* s, uint16_t* out);
int tzo_search_improved_policy(tzo_search* s, float visitations, int amax, float* out);
int tzo_search_improved_policy_each(tzo_search* s, const float* visitations, int amax, float* out);
int tzo_search_ube_target(tzo_search* s, float beta, float* out);
int tzo_search_step(tzo_search* s, const uint16_t* actions);
int tzo_search_restart_terminal(tzo_search* s, const int32_t* choice, int8_t* terminal_out);
int tzo_search_terminal_details(tzo_search* s, int8_t* reason_out, uint8_t* winner_out);
int tzo_search_play_moves(tzo_search* s, const uint16_t* actions, int8_t* ok_out);
int tzo_search_gumbel_sh(tzo_search* s, const float* betas, int sampled_actions, int search_budget, const float* gumbel, int amax,
                         uint16_t* selected_out);
}

// the oracle's restatement of the decisions above the search (oracle/host.hpp through oracle/capi.cpp)
struct tzo_host;
extern "C" {
tzo_host* tzo_host_create(tzo_search* s);
void tzo_host_destroy(tzo_host* h);
int tzo_host_choose_and_record(tzo_host* h, int kind, const uint16_t* halving, const double* draws, float visitations, uint16_t* expected_out,
                               uint8_t* sampled_out);
int tzo_host_complete(tzo_host* h, const int8_t* terminal, const float* betas);
int tzo_host_target(tzo_host* h, int i, tz_state* state_out, int amax, uint16_t* moves_out, float* policy_out, float* value_out, float* ube_out);
int tzo_host_reanalyze_target(tzo_search* s, int g, uint16_t selected, int amax, uint16_t* moves_out, float* policy_out, float* value_out,
                              float* ube_out);
}

#define TZS(name) tzo_search_##name
#define TZ_SEARCH_T tzo_search
// the two constructors take the search handle: give this build's versions names of their own
#define tz_selfplay_create tzh_selfplay_create
#define tz_reanalyze_create tzh_reanalyze_create
#define tz_compete tzh_compete
#define tz_puzzle_benchmark tzh_puzzle_benchmark
#include "../takzero_amd/csrc/tz_host.cpp"

static void dump(const std::string& path, const std::string& text) {
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
}

#ifdef TZ_HARNESS_WITH_NET
static void net_agent(void* user, int n_envs, const tz_state* states, const uint16_t* legal_idx, const int32_t* legal_count, int amax,
                      float* logits, float* value, float* variance) {
    if (tz_net_eval(static_cast<tz_net*>(user), n_envs, states, legal_idx, legal_count, amax, logits, value, variance)) {
        fprintf(stderr, "tz_net_eval: %s\n", tz_last_error());
        abort();
    }
}
#endif

int main(int argc, char** argv) {
    if (argc != 12 && argc != 16) return 2;
    const int n = atoi(argv[1]), hk = atoi(argv[2]), agent = atoi(argv[3]), B = atoi(argv[4]), kind = atoi(argv[5]), sims = atoi(argv[6]),
              k = atoi(argv[7]), exploration = atoi(argv[8]), moves = atoi(argv[9]);
    const uint64_t seed = strtoull(argv[10], nullptr, 10);
    const std::string prefix = argv[11];
    tzo_agent_fn fn = nullptr;
    void* user = nullptr;
#ifdef TZ_HARNESS_WITH_NET
    if (agent == 0) {
        if (argc != 16) return 2;
        tz_net* net = nullptr;
        if (tz_net_create(n, atoi(argv[13]), 0, atoi(argv[15]), atoi(argv[14]), &net) || tz_net_load_weights(net, argv[12])) {
            fprintf(stderr, "net: %s\n", tz_last_error());
            return 8;
        }
        fn = net_agent;
        user = net;
    }
#endif
    tzo_search* s = tzo_search_create(agent, fn, user, B, n, hk);
    tz_selfplay* sp = nullptr;
    const int shard = getenv("TZH_SHARD") ? atoi(getenv("TZH_SHARD")) : 0;
    const bool mark = getenv("TZH_MARK") != nullptr;
    if (tz_selfplay_create(s, sims, seed, shard, kind, k, exploration, &sp)) return 3;
    tz_comm* comm = nullptr;
    if (getenv("TZH_COMM_DIR")) {
        if (tz_comm_create_fs(getenv("TZH_COMM_DIR"), atoi(getenv("TZH_RANK")), atoi(getenv("TZH_WORLD")), 120.0, &comm) ||
            tz_selfplay_set_comm(sp, comm, getenv("TZH_WRITER") ? atoi(getenv("TZH_WRITER")) : 0)) {
            fprintf(stderr, "comm: %s\n", tz_last_error());
            return 11;
        }
    }
    // Every decision the native driver takes above the search is checked, move by move, against the oracle's restatement of
    // the reference's code (oracle/host.hpp), which looks at the same trees through its own Node / Eval objects and is handed
    // the very draws the driver consumed: chosen actions (node/mod.rs:170-207, selfplay/src/main.rs:138-153), which games
    // consumed a draw, and every completed target - position, policy, value, UBE, order (selfplay/src/main.rs:238-329).
    unsigned long long checks = 0, mismatches = 0, sampled_games = 0;
    tzo_host* host = kind == 2 ? nullptr : tzo_host_create(s);
    HostTrace trace;
    auto bits_equal = [](float a, float b) { return memcmp(&a, &b, 4) == 0; };
    if (host) {
        int lg = 0;
        while ((1 << (lg + 1)) <= k) lg++;
        const float visitations = lg > 0 ? (float)((sims / lg / k) * ((1 << lg) - 1)) : 0.0f;   // IMPROVED_POLICY_VISITATIONS, selfplay/src/main.rs:47-52
        trace.chosen = [&, visitations](const std::vector<double>& draws, const std::vector<uint16_t>& halving, const std::vector<uint16_t>& actions) {
            std::vector<uint16_t> expected(B);
            std::vector<uint8_t> sampled(B);
            std::vector<double> d(draws);
            for (auto& x : d)
                if (x != x) x = 0.0;
            tzo_host_choose_and_record(host, kind, halving.data(), d.data(), visitations, expected.data(), sampled.data());
            for (int g = 0; g < B; g++) {
                if (expected[g] == 0xFFFF) continue;
                checks += 2;
                mismatches += expected[g] != actions[g];
                mismatches += (sampled[g] != 0) != (draws[g] == draws[g]);   // the driver consumed a draw iff the reference would
                sampled_games += sampled[g] != 0;
            }
        };
        trace.completed = [&](const std::vector<int8_t>& terminal, const std::vector<tz_state>& states, const std::vector<uint16_t>& mv,
                              const std::vector<float>& pol, const std::vector<int32_t>& count, const std::vector<float>& value,
                              const std::vector<float>& ube, int amax) {
            const int T = tzo_host_complete(host, terminal.data(), sp->betas.data());
            checks++;
            if (T != (int)count.size()) {
                mismatches++;
                return;
            }
            std::vector<uint16_t> emv(amax);
            std::vector<float> epol(amax);
            for (int i = 0; i < T; i++) {
                tz_state st;
                float v = 0, u = 0;
                const int c = tzo_host_target(host, i, &st, amax, emv.data(), epol.data(), &v, &u);
                checks++;
                bool same = c == count[i] && !memcmp(&st, &states[i], sizeof st) && bits_equal(v, value[i]) && bits_equal(u, ube[i]);
                for (int j = 0; same && j < c; j++) same = emv[j] == mv[(size_t)i * amax + j] && bits_equal(epol[j], pol[(size_t)i * amax + j]);
                mismatches += !same;
            }
        };
        tz_selfplay_set_trace(sp, &trace);
    }
    std::string targets, replays, expl;
    for (int m = 0; m < moves; m++) {
        if (tz_selfplay_play_move(sp) || tz_selfplay_exchange(sp)) {
            fprintf(stderr, "play_move: %s\n", tz_last_error());
            return 4;
        }
        targets += sp->targets_text;
        replays += sp->replays_text;
        expl += sp->exploration_text;
        if (mark) {
            targets += "#move\n";
            replays += "#move\n";
            expl += "#move\n";
        }
        sp->targets_text.clear();
        sp->replays_text.clear();
        sp->exploration_text.clear();
    }
    dump(prefix + ".targets", targets);
    dump(prefix + ".replays", replays);
    dump(prefix + ".exploration", expl);
    tz_selfplay_destroy(sp);
    if (host) tzo_host_destroy(host);
    if (comm) {
        uint64_t collectives = 0, bytes = 0;
        tz_comm_info(comm, nullptr, nullptr, nullptr, &collectives, &bytes);
        printf("positions 0 collectives %llu bytes %llu\n", (unsigned long long)collectives, (unsigned long long)bytes);
        tz_comm_destroy(comm);
        tzo_search_destroy(s);
        return 0;
    }
    // reanalyze over what was just played (searches from fresh trees, so it is independent of the loop above)
    tz_reanalyze* ra = nullptr;
    if (tz_reanalyze_create(s, kind == 1 ? sims : 32, seed + 1, 0, 1, kind == 1 ? 1 : 0, k, &ra)) return 5;
    uint64_t added = 0, total = 0;
    if (tz_reanalyze_feed(ra, (prefix + ".replays").c_str(), &added, &total)) return 6;
    HostTrace rtrace;   // reanalyze targets against reanalyze/src/main.rs:184-203 as the oracle restates it
    unsigned long long proven_children = 0;
    rtrace.reanalyzed = [&](const std::vector<uint16_t>& selected, const std::vector<uint16_t>& mv, const std::vector<float>& pol,
                            const std::vector<int32_t>& count, const std::vector<float>& value, const std::vector<float>& ube, int amax) {
        std::vector<uint16_t> emv(amax);
        std::vector<float> epol(amax);
        std::vector<tz_root_info> info(B);
        tzo_search_root_info(s, info.data());
        std::vector<uint8_t> tag((size_t)B * amax);
        std::vector<uint16_t> cm((size_t)B * amax);
        tzo_search_root_children(s, amax, cm.data(), nullptr, tag.data(), nullptr, nullptr, nullptr, nullptr);
        for (int g = 0; g < B; g++) {
            float v = 0, u = 0;
            const int c = tzo_host_reanalyze_target(s, g, selected[g], amax, emv.data(), epol.data(), &v, &u);
            checks++;
            bool same = c == count[g] && bits_equal(v, value[g]) && bits_equal(u, ube[g]);
            for (int j = 0; same && j < c; j++) same = emv[j] == mv[(size_t)g * amax + j] && bits_equal(epol[j], pol[(size_t)g * amax + j]);
            mismatches += !same;
            if (info[g].eval_tag == TZ_EVAL_VALUE)   // the case ADVICE r1 flagged: the selected child of an unsolved root is proven
                for (int j = 0; j < c; j++)
                    if (cm[(size_t)g * amax + j] == selected[g] && tag[(size_t)g * amax + j] != TZ_EVAL_VALUE) proven_children++;
        }
    };
    tz_reanalyze_set_trace(ra, &rtrace);
    std::string re;
    if (total >= (uint64_t)B) {
        for (int it = 0; it < 2; it++) {
            if (tz_reanalyze_iterate(ra)) {
                fprintf(stderr, "iterate: %s\n", tz_last_error());
                return 7;
            }
        }
        re = ra->targets_text;
    }
    dump(prefix + ".reanalyze", re);
    printf("positions %llu host_checks %llu host_mismatches %llu sampled_games %llu proven_selected_children %llu\n", (unsigned long long)total, checks,
           mismatches, sampled_games, proven_children);
    tz_reanalyze_destroy(ra);
    // evaluation::compete and the puzzle benchmark on the 16 openings (game g starts from opening g % 16)
    {
        tzo_search* other = tzo_search_create(agent == 1 ? 2 : 1, nullptr, nullptr, B, n, hk);  // the other built-in agent plays Black
        std::vector<int32_t> choice(B);
        for (int g = 0; g < B; g++) choice[g] = g % 16;
        tzo_search_new_openings(s, choice.data());
        std::vector<tz_state> games(B);
        tzo_search_get_positions(s, games.data());
        int32_t res[3] = {0, 0, 0}, pz[6] = {0, 0, 0, 0, 0, 0};
        const int budget = k >= 2 ? k * (31 - __builtin_clz((unsigned)k)) * 2 : 8;
        if (tz_compete(s, other, games.data(), 0.0f, 0.25f, seed + 2, k >= 2 ? k : 4, k >= 2 ? budget : 16, 6, res)) {
            fprintf(stderr, "compete: %s\n", tz_last_error());
            return 9;
        }
        std::vector<tz_state> puzzles(B + B / 2);
        std::vector<uint16_t> solutions(B + B / 2, 0);
        for (size_t i = 0; i < puzzles.size(); i++) puzzles[i] = games[i % B];
        if (tz_puzzle_benchmark(s, puzzles.data(), solutions.data(), (int)puzzles.size(), 1, seed + 3, k >= 2 ? k : 4, k >= 2 ? budget : 16, pz) ||
            tz_puzzle_benchmark(s, puzzles.data(), solutions.data(), (int)puzzles.size(), 0, seed + 3, k >= 2 ? k : 4, k >= 2 ? budget : 16, pz + 3)) {
            fprintf(stderr, "puzzle: %s\n", tz_last_error());
            return 10;
        }
        std::vector<tz_state> fin(B);
        tzo_search_get_positions(other, fin.data());
        std::string out = std::to_string(res[0]) + " " + std::to_string(res[1]) + " " + std::to_string(res[2]);
        for (int i = 0; i < 6; i++) out += " " + std::to_string(pz[i]);
        out += "\n";
        char buf[256];
        for (int g = 0; g < B; g++) {
            tz_state_to_tps(&fin[g], buf, sizeof buf);
            out += buf;
            out += "\n";
        }
        dump(prefix + ".consumers", out);
        tzo_search_destroy(other);
    }
    tzo_search_destroy(s);
    return 0;
}
