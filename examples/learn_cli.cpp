// learn_cli — the reference's `learn --directory DIR` (learn/src/main.rs:99-270) as a plain C++ program over the C ABI of
// libtakzero_hip.so: no Python, no torch, no LibTorch.  Resumes from the model_<steps>.ot with most steps or initialises a
// network (Net::new) and writes model_0000000.ot; optional pre-training on uniformly random games (:425-484); writes
// model_latest.ot every --steps-per-save steps and model_<steps>.ot every --steps-per-checkpoint as LibTorch archives
// (the files the reference's selfplay / reanalyze / evaluation load), buffer_lengths.txt for back-pressure.
//
//   g++ -std=c++17 -O2 examples/learn_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o learn_cli
//   ./learn_cli --directory DIR --arch 5 --steps 100000
#include <dirent.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// get_model_path_with_most_steps (learn/src/main.rs:272-289)
static bool model_with_most_steps(const std::string& dir, long long& steps, std::string& path) {
    DIR* d = opendir(dir.c_str());
    if (!d) return false;
    bool found = false;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() < 4 || name.compare(name.size() - 3, 3, ".ot")) continue;
        const std::string stem = name.substr(0, name.size() - 3);
        const size_t us = stem.find('_');
        if (us == std::string::npos || us + 1 >= stem.size()) continue;
        const std::string tail = stem.substr(us + 1);
        if (!std::all_of(tail.begin(), tail.end(), [](char c) { return c >= '0' && c <= '9'; })) continue;
        const long long s = atoll(tail.c_str());
        if (!found || s > steps) {
            steps = s;
            path = dir + "/" + name;
            found = true;
        }
    }
    closedir(d);
    return found;
}

static int on_step(void* user, int64_t step, const float* losses, const tz_state*, int) {
    if (*static_cast<int*>(user) && step % 100 == 0)
        fprintf(stderr, "step %lld: loss_policy %.5f loss_value %.5f loss_ube %.5f\n", (long long)step, losses[0], losses[1], losses[2]);
    return 0;
}

int main(int argc, char** argv) {
    std::string directory;
    int arch = TZ_ARCH_NET5, n = 5, blocks = 0, batch = 128, verbose = 0;
    long long steps = -1, pre_training_steps = 1000, initial_targets = 128 * 2000, steps_before_reanalyze = 5000;
    int min_selfplay = 10000, min_reanalyze = 2000, steps_per_save = 100, steps_per_checkpoint = 50000;
    double read_interval = 10.0, sleep_s = 30.0, wait_limit = -1.0;
    unsigned long long seed = std::random_device{}();
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--directory") directory = next();
        else if (a == "--arch") arch = atoi(next());
        else if (a == "--n") n = atoi(next());
        else if (a == "--blocks") blocks = atoi(next());
        else if (a == "--batch") batch = atoi(next());
        else if (a == "--steps") steps = atoll(next());
        else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
        else if (a == "--pre-training-steps") pre_training_steps = atoll(next());
        else if (a == "--initial-targets") initial_targets = atoll(next());
        else if (a == "--steps-before-reanalyze") steps_before_reanalyze = atoll(next());
        else if (a == "--min-selfplay") min_selfplay = atoi(next());
        else if (a == "--min-reanalyze") min_reanalyze = atoi(next());
        else if (a == "--steps-per-save") steps_per_save = atoi(next());
        else if (a == "--steps-per-checkpoint") steps_per_checkpoint = atoi(next());
        else if (a == "--read-interval") read_interval = atof(next());
        else if (a == "--sleep") sleep_s = atof(next());
        else if (a == "--wait-limit") wait_limit = atof(next());
        else if (a == "--verbose") verbose = 1;
        else {
            fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (directory.empty()) {
        fprintf(stderr, "usage: learn_cli --directory DIR [--arch 4|5|6|100 --n N --blocks K --batch B --steps S --seed X ...]\n");
        return 2;
    }
    if (arch == TZ_ARCH_NET5) n = 5;
    if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    const bool hash = arch == TZ_ARCH_NET4_SIMHASH || arch == TZ_ARCH_NET6_SIMHASH;
    fprintf(stderr, "seed = %llu\n", seed);
    tz_trainer* trainer = nullptr;
    tz_net* net = nullptr;   // Net::new for the initial variables; for SimHash nets also the set that update_counts feeds
    CHECK(tz_trainer_create(n, arch, 0, blocks, batch, 1e-4f, &trainer));
    CHECK(tz_net_create(n, arch, 0, TZ_PREC_F16, blocks, &net));
    long long starting_steps = 0;
    std::string resume;
    if (model_with_most_steps(directory, starting_steps, resume)) {
        fprintf(stderr, "Resuming with model at %s\n", resume.c_str());
        CHECK(tz_trainer_load(trainer, resume.c_str()));
        if (hash) CHECK(tz_net_load_weights(net, resume.c_str()));   // picks up bitvec.bin beside it
    } else {
        fprintf(stderr, "Initializing a network model\n");
        CHECK(tz_net_init_random(net, seed));
        CHECK(tz_trainer_from_net(trainer, net));
        CHECK(tz_net_save(net, (directory + "/model_0000000.ot").c_str()));
        starting_steps = 0;
        if (pre_training_steps > 0) {
            // pre_training (:425-484): uniformly random games from the openings, uniform policy targets, discounted result as
            // value; every target used once; the UBE head is not trained
            tz_search* mcts = nullptr;
            tz_selfplay* sp = nullptr;
            tz_learn* pre = nullptr;
            CHECK(tz_search_create(nullptr, TZ_AGENT_DUMMY, 1024, n, 4, 1 << 10, &mcts));
            CHECK(tz_selfplay_create(mcts, 0, seed, 0, 2, 64, 0, &sp));
            CHECK(tz_learn_create(trainer, 4, seed + 1, 1, 1, &pre));
            std::vector<std::string> lines;
            std::vector<char> text(64 << 20);
            while ((long long)lines.size() < initial_targets) {
                CHECK(tz_selfplay_play_move(sp));
                uint64_t size = 0;
                CHECK(tz_selfplay_take_text(sp, 0, text.data(), text.size(), &size));
                for (uint64_t a0 = 0, k = 0; k < size; k++)
                    if (text[k] == '\n') {
                        lines.emplace_back(text.data() + a0, k + 1 - a0);
                        a0 = k + 1;
                    }
                CHECK(tz_selfplay_take_text(sp, 1, text.data(), text.size(), &size));
            }
            std::mt19937_64 rng(seed ^ 11);
            std::shuffle(lines.begin(), lines.end(), rng);
            std::string all;
            for (auto& l : lines) all += l;
            if (FILE* f = fopen((directory + "/targets-initial.txt").c_str(), "wb")) {
                fwrite(all.data(), 1, all.size(), f);
                fclose(f);
            }
            uint64_t added = 0;
            CHECK(tz_learn_add_lines(pre, 0, all.data(), all.size(), 0, &added));
            const long long todo = std::min<long long>(pre_training_steps, (long long)lines.size() / batch);
            for (long long s = 0; s < todo; s++) {
                float losses[3];
                CHECK(tz_learn_step(pre, 0, 0, 1, losses));
                if (verbose && s % 100 == 0) fprintf(stderr, "pre-training step %lld: %.5f %.5f\n", s, losses[0], losses[1]);
            }
            tz_learn_destroy(pre);
            tz_selfplay_destroy(sp);
            tz_search_destroy(mcts);
            starting_steps += pre_training_steps;
            char name[64];
            snprintf(name, sizeof name, "/model_%07lld.ot", starting_steps);
            CHECK(tz_trainer_save(trainer, (directory + name).c_str()));
        }
    }
    CHECK(tz_trainer_save(trainer, (directory + "/model_latest.ot").c_str()));
    tz_learn* loop = nullptr;
    CHECK(tz_learn_create(trainer, 4, seed, 4, 4, &loop));
    CHECK(tz_learn_set_save_points(loop, steps_per_save, steps_per_checkpoint, hash ? net : nullptr));
    int64_t model_steps = starting_steps;
    const int rc = tz_learn_run(loop, directory.c_str(), starting_steps, steps, min_selfplay, min_reanalyze, steps_before_reanalyze, read_interval,
                                sleep_s, wait_limit, on_step, &verbose, &model_steps);
    if (rc != 0) fprintf(stderr, "tz_learn_run stopped: %s\n", tz_last_error());
    printf("model_steps %lld starting_steps %lld rc %d\n", (long long)model_steps, starting_steps, rc);
    tz_learn_destroy(loop);
    tz_trainer_destroy(trainer);
    tz_net_destroy(net);
    return rc == 0 || rc == TZ_ESTATE ? 0 : 1;
}
