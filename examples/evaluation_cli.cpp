// evaluation_cli — the reference's `evaluation --model-path DIR` (evaluation/src/main.rs:131-222) as a plain C++ program over the C
// ABI of libtakzero_hip.so: the model files of DIR (*.ot, `model_latest` left out, sorted, every --step'th) are matched up two at
// a time at random; both are loaded with Network::load_partial semantics (tz_net_load_partial), 64 games start from an opening
// book (--opening-book FILE, one TPS per line, sampled without repetition) or from `new_opening_with_random_steps` (an opening
// and two or three uniformly random legal moves), and `compete` (tz_compete: one tree per side and game, Gumbel sequential
// halving 64 / 768, both trees stepped with the mover's action) plays them with either model as White.  One line per match:
//   model_a.ot vs. model_b.ot: Evaluation { wins: W, losses: L, draws: D } P%
//
//   g++ -std=c++17 -O2 examples/evaluation_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o evaluation_cli
//   ./evaluation_cli --model-path DIR --arch 4 [--step 1 --opening-book book.tps --rounds 10]
#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "takzero_hip.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != 0) {                                                 \
            fprintf(stderr, "%s failed: %s\n", #call, tz_last_error());    \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static std::vector<std::string> model_files(const std::string& dir, int step) {
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string f = e->d_name;
            if (f.size() > 3 && f.substr(f.size() - 3) == ".ot" && f != "model_latest.ot") names.push_back(f);
        }
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    std::vector<std::string> out;
    for (size_t i = 0; i < names.size(); i += (size_t)std::max(1, step)) out.push_back(names[i]);
    return out;
}

// Env::new_opening_with_random_steps for every game of a Dummy-agent search: an opening, then 2 or 3 random legal moves
static int random_openings(tz_search* dummy, int games, int amax, std::mt19937_64& rng, std::vector<tz_state>& out) {
    std::vector<int32_t> choice(games);
    for (auto& c : choice) c = (int32_t)(rng() % 16);
    CHECK(tz_search_new_openings(dummy, choice.data()));
    std::vector<int> steps(games);
    for (auto& s : steps) s = 2 + (int)(rng() % 2);
    std::vector<float> betas(games, 0.0f);
    std::vector<tz_root_info> info(games);
    std::vector<uint16_t> moves((size_t)games * amax), act(games);
    std::vector<int8_t> ok(games);
    for (int ply = 0; ply < 3; ply++) {
        CHECK(tz_search_simulate(dummy, betas.data(), 1));          // expands the roots: their children are the legal moves
        CHECK(tz_search_root_info(dummy, info.data()));
        CHECK(tz_search_root_children(dummy, amax, moves.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
        for (int g = 0; g < games; g++) {
            const int n = (int)info[g].n_children;
            act[g] = (ply < steps[g] && n > 0) ? moves[(size_t)g * amax + rng() % n] : (uint16_t)0xFFFF;
        }
        CHECK(tz_search_play_moves(dummy, act.data(), ok.data()));
    }
    out.resize(games);
    CHECK(tz_search_get_positions(dummy, out.data()));
    return 0;
}

int main(int argc, char** argv) {
    std::string model_path, book_path;
    int arch = TZ_ARCH_NET4_SIMHASH, n = 4, blocks = 0, step = 1, games = 64, k = 64, budget = 768, max_moves = 200, rounds = -1;
    int precision = TZ_PREC_F16, device = 0;
    unsigned long long seed = std::random_device{}();
    double sleep_s = 600.0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--model-path") model_path = next();
        else if (a == "--opening-book") book_path = next();
        else if (a == "--step") step = atoi(next());
        else if (a == "--arch") arch = atoi(next());
        else if (a == "--n") n = atoi(next());
        else if (a == "--blocks") blocks = atoi(next());
        else if (a == "--games") games = atoi(next());
        else if (a == "--sampled-actions") k = atoi(next());
        else if (a == "--budget") budget = atoi(next());
        else if (a == "--max-moves") max_moves = atoi(next());
        else if (a == "--rounds") rounds = atoi(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--sleep") sleep_s = atof(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--bf16") precision = TZ_PREC_BF16;
        else if (a == "--f16c8") precision = TZ_PREC_F16C8;
        else if (a == "--f16c6") precision = TZ_PREC_F16C6;
        else if (a == "--f16x2") precision = TZ_PREC_F16X2;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (model_path.empty()) {
        fprintf(stderr, "usage: evaluation_cli --model-path DIR [--step K --opening-book FILE --arch 4|5|6|100 --n N --blocks K --games 64 "
                        "--sampled-actions 64 --budget 768 --max-moves 200 --rounds R --seed X --sleep SECONDS --device G --bf16|--f16c6|--f16c8|--f16x2]\n");
        return 2;
    }
    if (arch == TZ_ARCH_NET5) n = 5;
    if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    printf("seed: %llu\n", seed);
    std::mt19937_64 rng(seed);
    tz_net *a = nullptr, *b = nullptr;
    tz_search *sa = nullptr, *sb = nullptr, *dummy = nullptr;
    CHECK(tz_net_create(n, arch, device, precision, blocks, &a));
    CHECK(tz_net_create(n, arch, device, precision, blocks, &b));
    CHECK(tz_net_init_random(a, 1));     // load_partial keeps what a file does not hold: Net::new first, as the reference does
    CHECK(tz_net_init_random(b, 2));
    CHECK(tz_search_create(a, TZ_AGENT_NET, games, n, 4, 0, &sa));
    CHECK(tz_search_create(b, TZ_AGENT_NET, games, n, 4, 0, &sb));
    CHECK(tz_search_create(nullptr, TZ_AGENT_DUMMY, games, n, 4, 1 << 10, &dummy));
    int amax = 0;
    CHECK(tz_search_shape(dummy, nullptr, nullptr, nullptr, &amax));
    std::vector<tz_state> book;
    if (!book_path.empty()) {
        std::ifstream f(book_path);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty()) continue;
            tz_state s;
            if (tz_state_from_tps(line.c_str(), n, 4, &s) != 0) {
                fprintf(stderr, "Opening book should be valid TPS, one per line: %s\n", tz_last_error());
                return 1;
            }
            book.push_back(s);
        }
        if ((int)book.size() < games) {
            fprintf(stderr, "There should be enough games in the opening book to form a unique batch\n");
            return 1;
        }
    }
    for (int round = 0; rounds < 0 || round < rounds;) {
        const std::vector<std::string> paths = model_files(model_path, step);
        if (paths.size() < 2) {
            if (rounds >= 0 && sleep_s <= 0) {
                fprintf(stderr, "Too few models.\n");
                return 1;
            }
            printf("Too few models. Sleeping for %.0f s.\n", sleep_s);
            fflush(stdout);
            usleep((useconds_t)(sleep_s * 1e6));
            continue;
        }
        size_t ia = rng() % paths.size(), ib = rng() % (paths.size() - 1);
        if (ib >= ia) ib++;
        round++;
        char missing[4096];
        int n_missing = 0;
        if (tz_net_load_partial(a, (model_path + "/" + paths[ia]).c_str(), missing, sizeof missing, &n_missing) != 0) {
            printf("Cannot load %s\n", paths[ia].c_str());
            continue;
        }
        if (tz_net_load_partial(b, (model_path + "/" + paths[ib]).c_str(), missing, sizeof missing, &n_missing) != 0) {
            printf("Cannot load %s\n", paths[ib].c_str());
            continue;
        }
        std::vector<tz_state> start;
        if (!book.empty()) {
            std::vector<size_t> idx(book.size());
            for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
            std::shuffle(idx.begin(), idx.end(), rng);
            for (int g = 0; g < games; g++) start.push_back(book[idx[g]]);
        } else if (random_openings(dummy, games, amax, rng, start) != 0) {
            return 1;
        }
        int32_t r[3];
        CHECK(tz_compete(sa, sb, start.data(), 0.0f, 0.0f, rng(), k, budget, max_moves, r));
        printf("%s vs. %s: Evaluation { wins: %d, losses: %d, draws: %d } %.1f%%\n", paths[ia].c_str(), paths[ib].c_str(), r[0], r[1], r[2],
               100.0 * r[0] / std::max(1, r[0] + r[1] + r[2]));
        CHECK(tz_compete(sb, sa, start.data(), 0.0f, 0.0f, rng(), k, budget, max_moves, r));
        printf("%s vs. %s: Evaluation { wins: %d, losses: %d, draws: %d } %.1f%%\n", paths[ib].c_str(), paths[ia].c_str(), r[0], r[1], r[2],
               100.0 * r[0] / std::max(1, r[0] + r[1] + r[2]));
        fflush(stdout);
    }
    tz_search_destroy(sa);
    tz_search_destroy(sb);
    tz_search_destroy(dummy);
    tz_net_destroy(a);
    tz_net_destroy(b);
    return 0;
}
