// selfplay_cli — the reference's `selfplay --directory DIR` as a plain C++ program over the C ABI of libtakzero_hip.so
// (no Python, no torch): creates the network and the batched search on one GPU and hands control to tz_selfplay_run
// (selfplay/src/main.rs:63-205).  DIR/model_latest.ot — the LibTorch archive `learn` writes (the reference's, or
// examples/learn_cli.cpp / tz_learn_run here) — is re-read whenever it changes (selfplay/src/main.rs:107-121); --model names
// the first model (.ot or .tzw), without it the net starts from Net::new(seed).
// N shards: --rank R --world N --comm rccl|fs [--comm-dir D] runs one process per GPU (device = rank unless --device);
// the shards hand over after every move through tz_comm (RCCL all-gather of the packed targets; rank 0 appends everybody's
// lines to the directory's files) and rank 0's model reloads are broadcast to the others.
//
//   g++ -std=c++17 -O2 examples/selfplay_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o selfplay_cli
//   ./selfplay_cli --directory DIR --arch 5 --games 128 --sims 768 --search gumbel
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>

#include "takzero_hip.h"

struct Reload {
    tz_net* net;
    std::string path;
    tz_comm* comm = nullptr;
    int rank = 0;
    long long stamp_s = -1, stamp_ns = -1, size = -1, inode = -1;
    int reloads = 0;
    // --async-reload (one process, no communicator): a changed file is parsed and its device weights are built on another thread
    // while the games go on (tz_net_load_prepare); the move after that the weights are swapped in (tz_net_load_commit)
    bool async = false;
    std::future<tz_pending_weights*> preparing;
};

// rank 0 (or the only process) looks at the file; with a communicator every rank then takes the same branch
static int reload_model(void* user) {
    Reload* r = static_cast<Reload*>(user);
    int status = 1;   // 0 = a new model is active on the root
    if (r->rank == 0) {
        struct stat st;
        const bool changed = stat(r->path.c_str(), &st) == 0 &&   // no model yet: keep playing with the current one
                             !(st.st_mtim.tv_sec == r->stamp_s && st.st_mtim.tv_nsec == r->stamp_ns && st.st_size == r->size &&
                               (long long)st.st_ino == r->inode);
        if (r->async && !r->comm) {
            if (r->preparing.valid() && r->preparing.wait_for(std::chrono::seconds(0)) == std::future_status::ready) {
                if (tz_pending_weights* p = r->preparing.get()) {
                    if (tz_net_load_commit(r->net, p) == 0) r->reloads++;
                    else fprintf(stderr, "Cannot put the model in place: %s\n", tz_last_error());
                }
            }
            if (changed && !r->preparing.valid()) {
                tz_net* net = r->net;
                const std::string path = r->path;
                r->preparing = std::async(std::launch::async, [net, path]() -> tz_pending_weights* {
                    tz_pending_weights* p = nullptr;
                    if (tz_net_load_prepare(net, path.c_str(), &p) != 0) {
                        fprintf(stderr, "Cannot load model: %s, not retrying.\n", tz_last_error());   // the old weights stay active
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
        if (changed) {
            if (tz_net_load_weights(r->net, r->path.c_str()) != 0) {
                fprintf(stderr, "Cannot load model: %s, not retrying.\n", tz_last_error());  // the old weights stay active
            } else {
                status = 0;
            }
            r->stamp_s = st.st_mtim.tv_sec;   // a file that does not parse is not looked at again until it changes
            r->stamp_ns = st.st_mtim.tv_nsec;
            r->size = st.st_size;
            r->inode = (long long)st.st_ino;
        }
    }
    if (r->comm && tz_net_broadcast(r->net, r->comm, 0, status) != 0) {
        fprintf(stderr, "tz_net_broadcast: %s\n", tz_last_error());
        return -1;
    }
    if (r->comm && r->rank != 0) {   // learn the branch the root took
        return 0;
    }
    if (status == 0) r->reloads++;
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
    std::string directory, model, search = "gumbel", watch = "model_latest.ot", comm_kind, comm_dir;
    int arch = TZ_ARCH_NET5, n = 5, blocks = 0, games = 128, sims = 768, moves = -1, exploration = 0, k = 64, precision = TZ_PREC_F16;
    bool async_reload = false;
    int rank = 0, world = 1, device = -1;
    unsigned long long seed = 0;
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
        else if (a == "--watch") watch = next();
        else if (a == "--async-reload") async_reload = true;
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--rank") rank = atoi(next());
        else if (a == "--world") world = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--comm") comm_kind = next();
        else if (a == "--comm-dir") comm_dir = next();
        else if (a == "--f16x2") precision = TZ_PREC_F16X2;
        else if (a == "--f16c8") precision = TZ_PREC_F16C8;
        else if (a == "--f16c6") precision = TZ_PREC_F16C6;
        else if (a == "--f16") precision = TZ_PREC_F16;
        else if (a == "--bf16") precision = TZ_PREC_BF16;
        else if (a == "--exploration") exploration = 1;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (directory.empty() || (world > 1 && comm_kind != "rccl" && comm_kind != "fs")) {
        fprintf(stderr, "usage: selfplay_cli --directory DIR [--model FILE.ot|.tzw --watch model_latest.ot --arch 4|5|6|100 --n N --blocks K "
                        "--games B --sims S --search puct|gumbel --sampled-actions K --moves M --exploration --async-reload --f16|--bf16|--f16c6|--f16c8|--f16x2 "
                        "--wait-limit SECONDS --seed X] [--rank R --world N --comm rccl|fs --comm-dir D --device G]\n");
        return 2;
    }
    if (device < 0) device = comm_kind == "fs" ? 0 : rank;
    if (comm_dir.empty()) comm_dir = directory;
    if (arch == TZ_ARCH_NET5) n = 5;
    if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    tz_net* net = nullptr;
    tz_search* mcts = nullptr;
    tz_selfplay* sp = nullptr;
    tz_comm* comm = nullptr;
    CHECK(tz_net_create(n, arch, device, precision, blocks, &net));
    if (!model.empty()) CHECK(tz_net_load_weights(net, model.c_str()));
    else CHECK(tz_net_init_random(net, seed));      // Net::new(DEVICE, seed), selfplay/src/main.rs:71 (same seed on every rank)
    CHECK(tz_search_create(net, TZ_AGENT_NET, games, n, 4, 0, &mcts));
    CHECK(tz_selfplay_create(mcts, sims, seed, rank, search == "puct" ? 0 : 1, k, exploration, &sp));
    if (world > 1) {
        if (comm_kind == "rccl") {
            unsigned char id[TZ_COMM_ID_BYTES];
            CHECK(tz_comm_rendezvous_id(comm_dir.c_str(), rank, id, 300.0));
            CHECK(tz_comm_create_rccl(id, rank, world, device, &comm));
        } else {
            CHECK(tz_comm_create_fs(comm_dir.c_str(), rank, world, 600.0, &comm));
        }
        CHECK(tz_selfplay_set_comm(sp, comm, 0));
    }
    Reload reload{net, directory + "/" + watch, comm, rank};
    reload.async = async_reload && !comm;
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
    if (reload.preparing.valid())
        if (tz_pending_weights* p = reload.preparing.get()) tz_net_load_discard(p);
    tz_selfplay_destroy(sp);
    tz_search_destroy(mcts);
    tz_net_destroy(net);
    if (comm) tz_comm_destroy(comm);
    return 0;
}
