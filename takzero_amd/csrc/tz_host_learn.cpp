// tz_host_learn.cpp — learn::main above the native step (learn/src/main.rs:99-319, 330-374, 425-516), native host code:
// the two replay buffers with forced-use counts fed by tailing targets-selfplay.txt / targets-reanalyze.txt,
// create_batch (sampling without replacement, random board symmetry, dense policy / mask tensors), the step pipelined
// with the preparation of the next batch, buffer_lengths.txt, and the save points (the model files themselves are
// written by a callback: the LibTorch archive writer is a separate host tool).  takzero_amd/learn.py is the same loop in
// Python.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "tz_engine.h"
#include "tz_ot.h"

extern "C" int tz_trainer_snapshot(tz_trainer* t, TensorStore& out);   // tz_learn.hip

namespace {

struct Target {
    tz_state st;
    std::vector<uint16_t> moves;
    std::vector<float> pol;
    float value = 0.f, ube = 0.f;
    int uses = 0, model_steps = 0;
};

struct Buffer {  // Vec<TargetWithContext> fed from an append-only file (fill_buffer_with_targets, :291-319)
    std::vector<Target> items;
    uint64_t seek = 0;
    int forced_uses = 4;
};

// the 8 board symmetries (target.rs:41-53): index = rot*2 + mirror; the draw is uniform, the order is immaterial
void symmetry(int n, int sym, int x, int y, int& ox, int& oy) {
    if (sym & 1) x = n - 1 - x;
    for (int r = 0; r < ((sym >> 1) & 3); r++) {
        const int nx = n - 1 - y, ny = x;
        x = nx;
        y = ny;
    }
    ox = x;
    oy = y;
}

}  // namespace

struct tz_learn {
    tz_trainer* trainer = nullptr;
    int n = 0, nn = 0, B = 0, half_komi = 0, out = 0, amax = 0;
    std::mt19937_64 rng;
    Buffer buf[2];                        // 0 selfplay (exploitation), 1 reanalyze
    int perm[8][36];                      // square -> square under each symmetry
    int dirmap[8][4];                     // move_index direction (Up, Right, Down, Left) under each symmetry
    // one batch of tensors (two sets: the step of one runs while the next is being built)
    struct Tensors {
        std::vector<tz_state> states;
        std::vector<float> policy, value, ube;
        std::vector<uint8_t> mask;
    } tensors[2];
    uint64_t steps_done = 0;
    // save points of learn::main (learn/src/main.rs:247-266), written by tz_learn_run itself when set (tz_learn_set_save_points)
    int steps_per_save = 0, steps_per_checkpoint = 0;
    tz_net* hash_net = nullptr;           // SimHash nets: update_counts after every step (:418), bitvec.bin beside the model
    std::thread writer;                   // the archive of the previous save point, written behind the training loop
    std::string writer_error;
};

namespace {

void build_tables(tz_learn* l) {
    static const int DX[4] = {0, 1, 0, -1}, DY[4] = {1, 0, -1, 0};  // Up, Right, Down, Left (repr.rs:49-71)
    const int n = l->n;
    for (int sym = 0; sym < 8; sym++) {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                int ox, oy;
                symmetry(n, sym, x, y, ox, oy);
                l->perm[sym][y * n + x] = oy * n + ox;
            }
        for (int d = 0; d < 4; d++) {
            const int sx = DX[d] >= 0 ? 0 : n - 1, sy = DY[d] >= 0 ? 0 : n - 1;
            int ax, ay, bx, by;
            symmetry(n, sym, sx, sy, ax, ay);
            symmetry(n, sym, sx + DX[d], sy + DY[d], bx, by);
            for (int e = 0; e < 4; e++)
                if (DX[e] == bx - ax && DY[e] == by - ay) l->dirmap[sym][d] = e;
        }
    }
}

void augment_into(const tz_learn* l, const Target& t, int sym, tz_state& st, std::vector<uint16_t>& moves) {
    const int n = l->n, nn = l->nn, patterns = (1 << n) - 2;
    st = t.st;
    memset(st.colors, 0, sizeof st.colors);
    memset(st.height, 0, sizeof st.height);
    memset(st.top, 0, sizeof st.top);
    for (int sq = 0; sq < nn; sq++) {
        const int to = l->perm[sym][sq];
        st.colors[to] = t.st.colors[sq];
        st.height[to] = t.st.height[sq];
        st.top[to] = t.st.top[sq];
    }
    moves.resize(t.moves.size());
    for (size_t i = 0; i < t.moves.size(); i++) {
        const int idx = t.moves[i], ch = idx / nn, sq = idx % nn;
        int ch2 = ch;
        if (ch >= 3) {
            const int d = (ch - 3) / patterns, pat = (ch - 3) % patterns;
            ch2 = 3 + pat + patterns * l->dirmap[sym][d];
        }
        moves[i] = (uint16_t)(ch2 * nn + l->perm[sym][sq]);
    }
}

int parse_into(tz_learn* l, Buffer& b, const char* text, uint64_t len, int model_steps, uint64_t* consumed_out, uint64_t* added_out) {
    const int chunk = 4096, amax = l->amax;
    std::vector<tz_state> st(chunk);
    std::vector<uint16_t> mv((size_t)chunk * amax);
    std::vector<float> pol((size_t)chunk * amax), val(chunk), ube(chunk);
    std::vector<int32_t> nm(chunk);
    uint64_t pos = 0, added = 0;
    for (;;) {
        int32_t cnt = 0, skipped = 0;
        uint64_t used = 0;
        int rc = tz_parse_targets(text + pos, len - pos, l->n, l->half_komi, chunk, amax, st.data(), mv.data(), pol.data(), nm.data(),
                                  val.data(), ube.data(), &cnt, &used, &skipped);
        if (rc) return rc;
        for (int i = 0; i < cnt; i++) {
            Target t;
            t.st = st[i];
            t.moves.assign(mv.begin() + (size_t)i * amax, mv.begin() + (size_t)i * amax + nm[i]);
            t.pol.assign(pol.begin() + (size_t)i * amax, pol.begin() + (size_t)i * amax + nm[i]);
            t.value = val[i];
            t.ube = ube[i];
            t.uses = b.forced_uses;
            t.model_steps = model_steps;
            b.items.push_back(std::move(t));
        }
        pos += used;
        added += (uint64_t)cnt;
        if (cnt < chunk || used == 0) break;
    }
    if (consumed_out) *consumed_out = pos;
    if (added_out) *added_out = added;
    return TZ_OK;
}

// create_batch + create_input_and_target_tensors (:486-516, :330-374) into tensor set `slot`
int make_batch(tz_learn* l, bool using_reanalyze, bool augment, int slot) {
    const int B = l->B;
    const int from[2] = {using_reanalyze ? B / 2 : B, using_reanalyze ? B - B / 2 : 0};
    for (int w = 0; w < 2; w++)
        if ((int)l->buf[w].items.size() < from[w]) return tz_fail(TZ_ESTATE, "tz_learn: not enough targets in a buffer for a batch");
    auto& T = l->tensors[slot];
    T.states.resize(B);
    T.policy.assign((size_t)B * l->out, 0.0f);
    T.mask.assign((size_t)B * l->out, 1);
    T.value.resize(B);
    T.ube.resize(B);
    std::vector<uint16_t> moves;
    int row = 0;
    for (int w = 0; w < 2; w++) {
        Buffer& b = l->buf[w];
        std::vector<Target> reused;   // go back only after the whole batch is drawn: no target twice in one batch
        for (int i = 0; i < from[w]; i++, row++) {
            // uniformly without replacement: what shuffling the whole buffer and draining its tail does, in O(batch)
            std::uniform_int_distribution<size_t> pick(0, b.items.size() - 1);
            const size_t j = pick(l->rng);
            std::swap(b.items[j], b.items.back());
            Target t = std::move(b.items.back());
            b.items.pop_back();
            const int sym = augment ? (int)(l->rng() % 8) : 0;
            if (augment) augment_into(l, t, sym, T.states[row], moves);
            else {
                T.states[row] = t.st;
                moves = t.moves;
            }
            for (size_t k = 0; k < moves.size(); k++) {
                T.policy[(size_t)row * l->out + moves[k]] = t.pol[k];   // policy_tensor
                T.mask[(size_t)row * l->out + moves[k]] = 0;             // move_mask: 1 = not a legal move
            }
            T.value[row] = t.value;
            T.ube[row] = t.ube;
            if (t.uses > 1) {  // TargetWithContext::reuse: back into the buffer with one use less
                t.uses--;
                reused.push_back(std::move(t));
            }
        }
        for (auto& t : reused) b.items.push_back(std::move(t));
    }
    return TZ_OK;
}

int step_batch(tz_learn* l, int slot, int train_ube, float* losses) {
    auto& T = l->tensors[slot];
    return tz_trainer_step(l->trainer, T.states.data(), T.policy.data(), T.mask.data(), T.value.data(), T.ube.data(), train_ube, 1, losses);
}

}  // namespace

extern "C" {

int tz_learn_create(tz_trainer* trainer, int half_komi, uint64_t seed, int selfplay_forced_uses, int reanalyze_forced_uses,
                    tz_learn** out) {
    if (!trainer || !out) return tz_fail(TZ_EINVAL, "tz_learn_create: null argument");
    *out = nullptr;
    std::unique_ptr<tz_learn> l(new tz_learn());
    l->trainer = trainer;
    int rc = tz_trainer_shape(trainer, &l->n, &l->B, nullptr);
    if (rc) return rc;
    l->nn = l->n * l->n;
    l->half_komi = half_komi;
    l->out = tz_policy_size(l->n);
    l->amax = l->n < 6 ? 512 : 1024;
    l->buf[0].forced_uses = selfplay_forced_uses > 0 ? selfplay_forced_uses : 4;   // learn/src/main.rs:59-60
    l->buf[1].forced_uses = reanalyze_forced_uses > 0 ? reanalyze_forced_uses : 4;
    std::seed_seq seq{(uint32_t)seed, (uint32_t)(seed >> 32), 0x1ea12u};
    l->rng.seed(seq);
    build_tables(l.get());
    *out = l.release();
    return TZ_OK;
}

int tz_learn_destroy(tz_learn* l) {
    if (l && l->writer.joinable()) l->writer.join();
    delete l;
    return TZ_OK;
}

// fill_buffer_with_targets: what was appended to `path` since the last call goes into buffer `which` (0 selfplay,
// 1 reanalyze); unparsable lines are skipped, a half-written last line is left for the next call.
int tz_learn_feed(tz_learn* l, int which, const char* path, int model_steps, uint64_t* added_out) {
    if (!l || !path || which < 0 || which > 1) return tz_fail(TZ_EINVAL, "tz_learn_feed: bad argument");
    if (added_out) *added_out = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return tz_fail(TZ_EPARSE, std::string("tz_learn_feed: cannot open ") + path);
    std::string data;
    if (fseek(f, (long)l->buf[which].seek, SEEK_SET) == 0) {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) data.append(chunk, got);
    }
    fclose(f);
    uint64_t used = 0;
    const int rc = parse_into(l, l->buf[which], data.data(), data.size(), model_steps, &used, added_out);
    l->buf[which].seek += used;
    return rc;
}

// the same from memory (pre-training targets, restart targets)
int tz_learn_add_lines(tz_learn* l, int which, const char* text, uint64_t len, int model_steps, uint64_t* added_out) {
    if (!l || !text || which < 0 || which > 1) return tz_fail(TZ_EINVAL, "tz_learn_add_lines: bad argument");
    return parse_into(l, l->buf[which], text, len, model_steps, nullptr, added_out);
}

int tz_learn_buffer_len(tz_learn* l, int which, uint64_t* len_out) {
    if (!l || !len_out || which < 0 || which > 1) return tz_fail(TZ_EINVAL, "tz_learn_buffer_len: bad argument");
    *len_out = l->buf[which].items.size();
    return TZ_OK;
}

// one create_batch + compute_loss_and_take_step (:486-516, 376-423)
int tz_learn_step(tz_learn* l, int using_reanalyze, int train_ube, int augment, float* losses_out) {
    if (!l) return tz_fail(TZ_EINVAL, "tz_learn_step: null handle");
    int rc = make_batch(l, using_reanalyze != 0, augment != 0, 0);
    if (rc) return rc;
    float losses[3] = {0, 0, 0};
    rc = step_batch(l, 0, train_ube, losses);
    if (losses_out) memcpy(losses_out, losses, sizeof losses);
    if (!rc) l->steps_done++;
    return rc;
}

// the batch tensors most recently built by tz_learn_step (diagnostic: lets a test check sampling and augmentation)
int tz_learn_last_batch(tz_learn* l, tz_state* states_out, float* policy_out, uint8_t* mask_out, float* value_out, float* ube_out) {
    if (!l) return tz_fail(TZ_EINVAL, "tz_learn_last_batch: null handle");
    const auto& T = l->tensors[0];
    if ((int)T.states.size() != l->B) return tz_fail(TZ_ESTATE, "tz_learn_last_batch: no batch built yet");
    if (states_out) memcpy(states_out, T.states.data(), sizeof(tz_state) * l->B);
    if (policy_out) memcpy(policy_out, T.policy.data(), sizeof(float) * T.policy.size());
    if (mask_out) memcpy(mask_out, T.mask.data(), T.mask.size());
    if (value_out) memcpy(value_out, T.value.data(), sizeof(float) * l->B);
    if (ube_out) memcpy(ube_out, T.ube.data(), sizeof(float) * l->B);
    return TZ_OK;
}

// The main training loop of learn::main (:172-269) for `steps` steps (< 0: forever), starting after `starting_steps`:
// re-read the target files every `read_interval_s`, write buffer_lengths.txt, wait while there are not enough targets,
// take a step (pipelined: the step of batch k runs on a worker thread while batch k+1 is sampled and built), and call
// on_step(user, model_steps, losses[3], states of the batch, batch size) after every step — the host saves
// model_latest / checkpoints there (STEPS_PER_SAVE, STEPS_PER_CHECKPOINT), updates the SimHash counts of hash nets
// (net.update_counts(&tensors.input), :418) and may return non-zero to stop.
int tz_learn_run(tz_learn* l, const char* directory, int64_t starting_steps, int64_t steps, int min_selfplay, int min_reanalyze,
                 int64_t steps_before_reanalyze, double read_interval_s, double sleep_s, double wait_limit_s,
                 int (*on_step)(void*, int64_t, const float*, const tz_state*, int), void* user, int64_t* model_steps_out) {
    if (!l || !directory) return tz_fail(TZ_EINVAL, "tz_learn_run: bad argument");
    const std::string dir = directory;
    using clock = std::chrono::steady_clock;
    auto last_loaded = clock::now() - std::chrono::hours(1);
    const auto t0 = clock::now();
    int64_t model_steps = starting_steps, done = 0;
    std::future<int> pending;
    float pending_losses[3] = {0, 0, 0};
    std::string pending_error;   // the error text is per thread: carry it over from the worker
    int64_t pending_step = 0;
    int slot = 0, pending_slot = 0, rc = TZ_OK;
    // a save point: the weights are read back here (the step has finished), the archive is written by a thread while
    // training goes on; at most one archive is in flight
    auto save_point = [&](int64_t step_no) -> int {
        const bool latest = l->steps_per_save > 0 && step_no % l->steps_per_save == 0;
        const bool numbered = l->steps_per_checkpoint > 0 && step_no % l->steps_per_checkpoint == 0;
        if (!latest && !numbered) return TZ_OK;
        if (l->writer.joinable()) l->writer.join();
        if (!l->writer_error.empty()) return tz_fail(TZ_EINVAL, "tz_learn_run: writing a model file failed: " + l->writer_error);
        auto snap = std::make_shared<TensorStore>();
        int r = tz_trainer_snapshot(l->trainer, *snap);
        if (r) return r;
        if (l->hash_net && (r = tz_net_save_bitset(l->hash_net, (dir + "/bitvec.bin").c_str()))) return r;   // net6_simhash.rs:152-170
        char numbered_name[64];
        snprintf(numbered_name, sizeof numbered_name, "/model_%07lld.ot", (long long)step_no);
        const std::string a = latest ? dir + "/model_latest.ot" : std::string(), b = numbered ? dir + numbered_name : std::string();
        l->writer = std::thread([l, snap, a, b]() {
            if (!a.empty() && weights_write_file(a.c_str(), *snap)) l->writer_error = tz_last_error();
            if (!b.empty() && weights_write_file(b.c_str(), *snap)) l->writer_error = tz_last_error();
        });
        return TZ_OK;
    };
    auto finish = [&]() -> int {
        if (!pending.valid()) return TZ_OK;
        int r = pending.get();
        if (r) return tz_fail(r, pending_error);
        l->steps_done++;
        if (l->hash_net) {   // net.update_counts(&tensors.input), learn/src/main.rs:418
            std::vector<uint32_t> idx(l->B);
            if ((r = tz_net_hash_indices(l->hash_net, l->B, l->tensors[pending_slot].states.data(), idx.data(), 1))) return r;
        }
        if ((r = save_point(pending_step))) return r;
        if (on_step && on_step(user, pending_step, pending_losses, l->tensors[pending_slot].states.data(), l->B))
            return tz_fail(TZ_ESTATE, "tz_learn_run: the on_step callback asked to stop");
        return TZ_OK;
    };
    while (steps < 0 || done < steps) {
        model_steps++;
        const bool using_reanalyze = model_steps >= steps_before_reanalyze;
        for (;;) {
            if (std::chrono::duration<double>(clock::now() - last_loaded).count() >= read_interval_s) {
                uint64_t added = 0;
                (void)tz_learn_feed(l, 0, (dir + "/targets-selfplay.txt").c_str(), (int)model_steps, &added);
                if (using_reanalyze) (void)tz_learn_feed(l, 1, (dir + "/targets-reanalyze.txt").c_str(), (int)model_steps, &added);
                last_loaded = clock::now();
                if (FILE* f = fopen((dir + "/buffer_lengths.txt").c_str(), "wb")) {   // :195-209
                    const unsigned long long a = l->buf[0].items.size(), b = l->buf[1].items.size();
                    fprintf(f, "%llu,%llu,%llu", a, b, a + b);
                    fclose(f);
                }
            }
            const bool enough = (int64_t)l->buf[0].items.size() >= std::max(min_selfplay, l->B) &&
                                (!using_reanalyze || (int64_t)l->buf[1].items.size() >= std::max(min_reanalyze, l->B));
            if (enough) break;
            if (wait_limit_s >= 0 && std::chrono::duration<double>(clock::now() - t0).count() > wait_limit_s) {
                (void)finish();
                if (model_steps_out) *model_steps_out = model_steps - 1;
                return tz_fail(TZ_ESTATE, "tz_learn_run: not enough targets");
            }
            std::this_thread::sleep_for(std::chrono::duration<double>(sleep_s));
        }
        if ((rc = make_batch(l, using_reanalyze, true, slot))) break;   // while the previous step runs
        if ((rc = finish())) break;
        pending_step = model_steps;
        pending_slot = slot;
        const int use = slot;
        pending = std::async(std::launch::async, [l, use, &pending_losses, &pending_error]() {
            const int r = step_batch(l, use, 1, pending_losses);
            if (r) pending_error = tz_last_error();
            return r;
        });
        slot ^= 1;
        done++;
    }
    const int last = finish();
    if (!rc) rc = last;
    if (l->writer.joinable()) l->writer.join();
    if (!rc && !l->writer_error.empty()) rc = tz_fail(TZ_EINVAL, "tz_learn_run: writing a model file failed: " + l->writer_error);
    if (model_steps_out) *model_steps_out = rc ? model_steps - 1 : model_steps;
    return rc;
}

int tz_learn_set_save_points(tz_learn* l, int steps_per_save, int steps_per_checkpoint, tz_net* hash_net) {
    if (!l || steps_per_save < 0 || steps_per_checkpoint < 0) return tz_fail(TZ_EINVAL, "tz_learn_set_save_points: bad argument");
    l->steps_per_save = steps_per_save;
    l->steps_per_checkpoint = steps_per_checkpoint;
    l->hash_net = hash_net;
    return TZ_OK;
}

}  // extern "C"
