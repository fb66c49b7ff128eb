// reanalyze_cli — the reference's `reanalyze --directory DIR` (reanalyze/src/main.rs:60-244) as a plain C++ program over the C ABI of
// libtakzero_hip.so: tails DIR/replays.txt (every pre-move state of every replay, moves re-validated on the device), waits for
// --min-positions positions (128 000 in the reference), then per iteration samples B positions without replacement, resets every
// tree, searches, and appends one target per position to DIR/targets-reanalyze.txt; DIR/model_latest.ot is re-read whenever it
// changes; DIR/buffer_lengths.txt throttles.  N GPUs: --rank R --world N takes replay line i for rank i mod N; every rank appends
// to the same file (one write per iteration), no exchange is needed.
//
//   g++ -std=c++17 -O2 examples/reanalyze_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o reanalyze_cli
//   ./reanalyze_cli --directory DIR --arch 5 --games 128 --sims 768 --search gumbel
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <string>

#include "takzero_hip.h"

struct Reload {
    tz_net* net;
    std::string path;
    long long stamp_s = -1, stamp_ns = -1, size = -1, inode = -1;
    int reloads = 0;
    // --async-reload: a changed file is parsed and its device weights are built on another thread while the iteration runs
    // (tz_net_load_prepare); they are swapped in before the next one (tz_net_load_commit)
    bool async = false;
    std::future<tz_pending_weights*> preparing;
};

// reanalyze retries every kind of load failure (reanalyze/src/main.rs:93-105): a missing or torn file leaves the old net playing
// and is looked at again on the next iteration
static int reload_model(void* user) {
    Reload* r = static_cast<Reload*>(user);
    struct stat st;
    if (stat(r->path.c_str(), &st) != 0) return 0;
    const bool changed = !(st.st_mtim.tv_sec == r->stamp_s && st.st_mtim.tv_nsec == r->stamp_ns && st.st_size == r->size && (long long)st.st_ino == r->inode);
    if (r->async) {
        if (r->preparing.valid() && r->preparing.wait_for(std::chrono::seconds(0)) == std::future_status::ready) {
            bool placed = false;
            if (tz_pending_weights* p = r->preparing.get()) {
                if (tz_net_load_commit(r->net, p) == 0) {
                    r->reloads++;
                    placed = true;
                } else {
                    fprintf(stderr, "Cannot put the model in place: %s\n", tz_last_error());
                }
            }
            if (!placed) {   // a torn or missing file: forget its stamp, so that the file is prepared again below (the synchronous road's retry)
                r->stamp_s = r->stamp_ns = -1;
                r->size = -1;
                r->inode = -1;
            }
        }
        const bool again = !(st.st_mtim.tv_sec == r->stamp_s && st.st_mtim.tv_nsec == r->stamp_ns && st.st_size == r->size && (long long)st.st_ino == r->inode);
        if (again && !r->preparing.valid()) {
            tz_net* net = r->net;
            const std::string path = r->path;
            r->preparing = std::async(std::launch::async, [net, path]() -> tz_pending_weights* {
                tz_pending_weights* p = nullptr;
                if (tz_net_load_prepare(net, path.c_str(), &p) != 0) {
                    fprintf(stderr, "Cannot load model: %s, retrying.\n", tz_last_error());
                    return nullptr;
                }
                return p;
            });
            r->stamp_s = st.st_mtim.tv_sec;
            r->stamp_ns = st.st_mtim.tv_nsec;
            r->size = st.st_size;
            r->inode = (long long)st.st_ino;
        }
        return 0;
    }
    if (!changed) return 0;
    if (tz_net_load_weights(r->net, r->path.c_str()) != 0) {
        fprintf(stderr, "Cannot load model: %s, retrying.\n", tz_last_error());
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
    std::string directory, model, search = "gumbel", watch = "model_latest.ot";
    int arch = TZ_ARCH_NET5, n = 5, blocks = 0, games = 128, sims = 768, iterations = -1, k = 64, precision = TZ_PREC_F16;
    int rank = 0, world = 1, device = -1, min_positions = 0;
    unsigned long long seed = 0;
    double wait_limit = -1.0;
    bool async_reload = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--directory") directory = next();
        else if (a == "--model") model = next();
        else if (a == "--watch") watch = next();
        else if (a == "--async-reload") async_reload = true;
        else if (a == "--search") search = next();
        else if (a == "--arch") arch = atoi(next());
        else if (a == "--n") n = atoi(next());
        else if (a == "--blocks") blocks = atoi(next());
        else if (a == "--games") games = atoi(next());
        else if (a == "--sims") sims = atoi(next());
        else if (a == "--iterations") iterations = atoi(next());
        else if (a == "--min-positions") min_positions = atoi(next());
        else if (a == "--sampled-actions") k = atoi(next());
        else if (a == "--wait-limit") wait_limit = atof(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--rank") rank = atoi(next());
        else if (a == "--world") world = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--f16x2") precision = TZ_PREC_F16X2;
        else if (a == "--f16c8") precision = TZ_PREC_F16C8;
        else if (a == "--f16c6") precision = TZ_PREC_F16C6;
        else if (a == "--bf16") precision = TZ_PREC_BF16;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (directory.empty()) {
        fprintf(stderr, "usage: reanalyze_cli --directory DIR [--model FILE --watch model_latest.ot --arch 4|5|6|100 --n N --blocks K --games B "
                        "--sims S --search puct|gumbel --sampled-actions K --iterations I --min-positions P --wait-limit SECONDS --seed X "
                        "--rank R --world N --device G --async-reload --bf16|--f16c6|--f16c8|--f16x2]\n");
        return 2;
    }
    if (arch == TZ_ARCH_NET5) n = 5;
    if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    if (device < 0) device = rank;
    tz_net* net = nullptr;
    tz_search* mcts = nullptr;
    tz_reanalyze* ra = nullptr;
    CHECK(tz_net_create(n, arch, device, precision, blocks, &net));
    if (!model.empty()) CHECK(tz_net_load_weights(net, model.c_str()));
    else CHECK(tz_net_init_random(net, seed));
    CHECK(tz_search_create(net, TZ_AGENT_NET, games, n, 4, 0, &mcts));
    CHECK(tz_reanalyze_create(mcts, sims, seed, rank, world, search == "puct" ? 0 : 1, k, &ra));
    Reload reload{net, directory + "/" + watch};
    reload.async = async_reload;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = tz_reanalyze_run(ra, directory.c_str(), iterations, min_positions, "", reload_model, &reload, wait_limit);
    if (rc != 0) fprintf(stderr, "tz_reanalyze_run stopped: %s\n", tz_last_error());
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long simulations = 0, evals = 0;
    CHECK(tz_search_counters(mcts, (uint64_t*)&simulations, (uint64_t*)&evals));
    printf("rc %d simulations %llu nn_evals %llu model_reloads %d seconds %.3f sims_per_s %.0f\n", rc, simulations, evals, reload.reloads, seconds,
           (double)simulations / seconds);
    if (reload.preparing.valid())
        if (tz_pending_weights* p = reload.preparing.get()) tz_net_load_discard(p);
    tz_reanalyze_destroy(ra);
    tz_search_destroy(mcts);
    tz_net_destroy(net);
    return rc == 0 || rc == TZ_ESTATE ? 0 : 1;
}
