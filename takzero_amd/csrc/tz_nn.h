// tz_nn.h — network object of libtakzero_hip.so (Network + Agent of the reference,
// takzero/src/network/mod.rs:10-45, net5.rs:17-285, net6_simhash.rs:23-324).
#pragma once
#include <string>
#include <vector>

#include "tz_engine.h"
#include "tz_ot.h"

struct ConvW {
    int taps = 0, cin = 0, cin_pad = 0, cout = 0, cout_pad = 0;
    uint16_t* w_mfma = nullptr;  // bf16 / fp16 bits, fragment order [tap][kc][ct][lane][8]
    uint16_t* w_lo = nullptr;    // TZ_PREC_F16X2: fp16((w - hi) * 2^11) in the same order
    unsigned char* w8 = nullptr; // TZ_PREC_F16C6: E2M3 records (build_layer); TZ_PREC_F16C8: FP8 E4M3 fragments [tap][m = cin/128][ct][term: lo, hi][lane][32] of (w - hi) * 2^11 * s and hi * s
    float c8_scale = 0.0f;       //   what the correction accumulator is multiplied by: 2^-11 / (s * C8_SX), s = the layer's power of two
    float* w_f32 = nullptr;      // [tap][cout][cin]
    float* bias = nullptr;       // [cout_pad]  (BatchNorm folded in)
};

struct tz_net {
    int n = 0, nn = 0, arch = 0, device = 0, precision = 0, blocks = 0;
    int cin = 0, cin_pad = 0, pol_ch = 0, pol_stride = 0, ppt = 0;
    bool loaded = false, has_rnd = false, has_hash = false;
    TensorStore store;         // host copy of the VarStore (fp32, `.a.` / `.b.` names): tz_net_save / clone / load_partial
    uint64_t weights_gen = 0;  // bumped by every successful load: captured graphs hold weight pointers
    ConvW conv_in, policy;
    std::vector<ConvW> res;  // 2 per block
    uint16_t* tower_w = nullptr;  // all residual-tower layers back to back (fused tower kernel)
    uint16_t* tower_w_lo = nullptr;  // TZ_PREC_F16X2: their lo halves
    unsigned char* tower_w8 = nullptr;  // TZ_PREC_F16C8: their FP8 fragments
    float* c8_scales = nullptr;         // TZ_PREC_F16C8: [2*blocks + 1] correction scales of the tower layers, then of the policy conv
    float* tower_bias = nullptr;  // [2*blocks][256]
    float* heads = nullptr;  // [value conv w 256, ube conv w 256, value lin nn, ube lin nn, bv, bu, lbv, lbu]
    ConvW rnd[2][3];         // [learning, target][input, hidden, final]  (fp32 path)
    uint16_t* rndw[3] = {nullptr, nullptr, nullptr};  // bf16 path: layer 1 = both nets side by side (2048 outputs),
    float* rndb[3] = {nullptr, nullptr, nullptr};     // layers 2/3 = [net][...] for grouped launches
    float rnd_min = 0.0f, rnd_max = 1.0f;
    float* simhash = nullptr;    // [in_size][32] fp32, reference order (c*nn + px)
    uint32_t* bitset = nullptr;  // 2^32 bits, allocated for hash archs
    // work buffers, sized for max_batch positions
    int max_batch = 0;
    void *act_a = nullptr, *act_b = nullptr, *act_c = nullptr;  // [max_batch*nn][256] bf16 or f32
    float* planes = nullptr;   // [max_batch][nn][cin] fp32 (f32 path, RND, hash, debug)
    float* policy_out = nullptr;  // [max_batch*nn][pol_stride]
    float *value = nullptr, *ube = nullptr, *variance = nullptr, *aux = nullptr;  // [max_batch]
    void *rnd_in = nullptr, *rnd_h1 = nullptr, *rnd_h2 = nullptr;  // RND activations
    float* rnd_out = nullptr;                                       // [2][max_batch][512]
    void* seeds = nullptr;     // TZ_PREC_F16C6: the blocks' inputs (fp32) between the two convs of a block (tz_nn_c6.hip)
    // tz_net_eval at small batches: several CUs per board group (net_mfma_kernel SPLIT): the groups' exchange buffer and arrival
    // counters, allocated at the first such call; `eval_split` is set for the duration of a tz_net_eval call only
    void* xch = nullptr;
    unsigned* xch_count = nullptr;
    bool eval_split = false;
    hipStream_t stream_rnd = nullptr;   // ... and the stream the RND MLP runs on beside the net kernel in such a call
    hipEvent_t ev_in = nullptr, ev_rnd = nullptr;
    hipStream_t stream = nullptr;
    void* dbg_buf = nullptr;   // diagnostic builds: in-kernel stamps of the last launch (tz_debug_net_clock)
    int dbg_groups = 0;
    // staging of the Agent surface (tz_net_eval): one pinned host buffer and one device buffer, grown on demand, so that
    // a call is one host-to-device copy, the kernels, one device-to-host copy
    void* eval_host = nullptr;
    void* eval_dev = nullptr;
    size_t eval_bytes = 0;
    // profiling of the dominant kernel (residual-tower conv)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> conv_events;
    bool profile = false;
    double conv_ms = 0.0;
    uint64_t conv_launches = 0;
};

int tz_net_ensure_batch(tz_net* net, int batch);
// Evaluate `count` positions: slot i reads states_dev[game_index_dev ? game_index_dev[i] : i].
// If count_dev is non-null the number of valid slots is read on the device (<= max_positions).
int tz_net_forward_device(tz_net* net, const tz_state* states_dev, const int32_t* game_index_dev,
                          const int32_t* count_dev, int count_host, int max_positions, hipStream_t st, NetOut* out);
int tz_nn_encode_planes(int n, int cin, const tz_state* states_dev, int count, float* planes_dev, hipStream_t st);
