// selfplay_cli — the reference's `selfplay --directory DIR` as a plain C++ program over the C ABI of libtakzero_hip.so
// (no Python, no torch): creates the network and the batched search on one GPU and hands control to tz_selfplay_run
// (selfplay/src/main.rs:63-205).  Weights come from a .tzw container (takzero_amd.weights / takzero_amd.ot convert the
// reference's .ot files); DIR/model_latest.tzw is re-read whenever it changes (selfplay/src/main.rs:107-121).
//
//   g++ -std=c++17 -O2 examples/selfplay_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o selfplay_cli
//   ./selfplay_cli --directory DIR --model DIR/model_latest.tzw --arch 5 --games 128 --sims 768 --search gumbel
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "takzero_hip.h"

struct Reload {
    tz_net* net;
    std::string path;
    long long stamp_s = -1, stamp_ns = -1, size = -1, inode = -1;
    int reloads = 0;
};

static int reload_model(void* user) {
    Reload* r = static_cast<Reload*>(user);
    struct stat st;
    if (stat(r->path.c_str(), &st) != 0) return 0;  // no new model yet: keep playing with the current one
    if (st.st_mtim.tv_sec == r->stamp_s && st.st_mtim.tv_nsec == r->stamp_ns && st.st_size == r->size && (long long)st.st_ino == r->inode)
        return 0;
    if (tz_net_load_weights(r->net, r->path.c_str()) != 0) {
        fprintf(stderr, "Cannot load model: %s, not retrying.\n", tz_last_error());  // the old weights stay active
        return 0;
    }
    r->stamp_s = st.st_mtim.tv_sec;
    r->stamp_ns = st.st_mtim.tv_nsec;
    r->size = st.st_size;
    r->inode = (long long)st.st_ino;
    r->reloads++;
    return 0;
}

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != 0) {                                                 \
            fprintf(stderr, "%s failed: %s\n", #call, tz_last_error());    \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    std::string directory, model, search = "gumbel";
    int arch = TZ_ARCH_NET5, n = 5, blocks = 0, games = 128, sims = 768, moves = -1, exploration = 0, k = 64, precision = TZ_PREC_F16;
    double wait_limit = -1.0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--directory") directory = next();
        else if (a == "--model") model = next();
        else if (a == "--search") search = next();
        else if (a == "--arch") arch = atoi(next());
        else if (a == "--n") n = atoi(next());
        else if (a == "--blocks") blocks = atoi(next());
        else if (a == "--games") games = atoi(next());
        else if (a == "--sims") sims = atoi(next());
        else if (a == "--moves") moves = atoi(next());
        else if (a == "--sampled-actions") k = atoi(next());
        else if (a == "--wait-limit") wait_limit = atof(next());
        else if (a == "--f16") precision = TZ_PREC_F16;
        else if (a == "--bf16") precision = TZ_PREC_BF16;
        else if (a == "--exploration") exploration = 1;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (directory.empty() || model.empty()) {
        fprintf(stderr, "usage: selfplay_cli --directory DIR --model FILE.tzw [--arch 4|5|6|100 --n N --blocks K --games B --sims S "
                        "--search puct|gumbel --sampled-actions K --moves M --exploration --f16 --wait-limit SECONDS]\n");
        return 2;
    }
    if (arch == TZ_ARCH_NET5) n = 5;
    if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    tz_net* net = nullptr;
    tz_search* mcts = nullptr;
    tz_selfplay* sp = nullptr;
    CHECK(tz_net_create(n, arch, 0, precision, blocks, &net));
    CHECK(tz_net_load_weights(net, model.c_str()));
    CHECK(tz_search_create(net, TZ_AGENT_NET, games, n, 4, 0, &mcts));
    CHECK(tz_selfplay_create(mcts, sims, 0, 0, search == "puct" ? 0 : 1, k, exploration, &sp));
    Reload reload{net, directory + "/model_latest.tzw"};
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(tz_selfplay_run(sp, directory.c_str(), moves, 32000, "", reload_model, &reload, wait_limit));
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long played = 0, targets = 0, replays = 0, simulations = 0, evals = 0, skipped = 0;
    CHECK(tz_selfplay_counters(sp, (uint64_t*)&played, (uint64_t*)&targets, (uint64_t*)&replays));
    CHECK(tz_search_counters(mcts, (uint64_t*)&simulations, (uint64_t*)&evals));
    CHECK(tz_search_pool_overflows(mcts, (uint64_t*)&skipped));
    long rss_kb = 0;   // resident set of this process (host memory of the driver: move history, text buffers)
    if (FILE* f = fopen("/proc/self/status", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "VmRSS: %ld kB", &rss_kb) == 1) break;
        fclose(f);
    }
    printf("moves %llu targets %llu replays %llu simulations %llu nn_evals %llu model_reloads %d seconds %.3f sims_per_s %.0f rss_mb %ld skipped_expansions %llu\n",
           played, targets, replays, simulations, evals, reload.reloads, seconds, (double)simulations / seconds, rss_kb / 1024, skipped);
    tz_selfplay_destroy(sp);
    tz_search_destroy(mcts);
    tz_net_destroy(net);
    return 0;
}
