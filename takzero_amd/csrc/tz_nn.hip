// tz_nn.hip — policy / value / uncertainty network forward on gfx950.
//
// Replaces the LibTorch op sequence issued by the reference's Agent impls
// (takzero/src/network/net5.rs:184-285, net6_simhash.rs:194-324, residual.rs:13-63, repr.rs:169-244;
// SURVEY.md §2.2 rows a-j):
//   * conv_mfma_kernel — 3x3 / 1x1 convolution and Linear as an im2col-free implicit GEMM on
//     NHWC bf16 with v_mfma_f32_16x16x32_bf16: a workgroup owns P whole boards (P*N*N output
//     pixels), stages their activations once in LDS and walks the 9 taps as row-shifted reads of
//     that tile (out-of-board taps read a zero row); weights stream from L2 in MFMA-fragment order
//     straight into registers; BatchNorm is folded into the weights, bias / residual / ReLU are
//     fused into the epilogue; the first layer builds its input planes from the packed game
//     state inside the tile loader (game_repr fused, repr.rs:169-228).
//   * heads_kernel — value and UBE heads (1x1 conv + ReLU + Linear (+tanh)), one wave per board.
//   * RND MLP (net5.rs:122-148,193-211) through the same MFMA kernel; SimHash lookup
//     (net6_simhash.rs:208-256) with the 2^32-bit set resident in HBM.
//   * an fp32 validation path (TZ_PREC_F32) with plain FMA kernels for the 1e-3 logit gate.
#include "tz_nn.h"
#include "tz_fp8.h"
#include "tz_fp6.h"
#include <type_traits>
#include "tz_ot.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

int tz_nn_launch_split(int sp, int n, const void* net_args, int max_positions, hipStream_t st);   // the second translation unit
int tz_nn_launch_c6(int n, const void* net_args, int max_positions, hipStream_t st);                // the third (tz_nn_c6.hip)
size_t tz_nn_c6_seed_bytes(int n, int max_positions);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The 16-bit MFMA path is written once for two storage types of equal MFMA rate: bf16 (TZ_PREC_BF16, the
// benchmarked default) and IEEE fp16 (TZ_PREC_F16: 3 more mantissa bits; stays within 1e-3 of the fp32 LibTorch
// logits through the 41 stacked convs, which bf16 only just misses).
template <typename ET>
struct Elem;
template <>
struct Elem<__bf16> {
    typedef bf16x8 x8;
    typedef bf16x4 x4;
    static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ __bf16 cvt(float x) { return (__bf16)x; }
    static __device__ __forceinline__ __bf16 relu_cvt(float x) { return (__bf16)(x > 0.f ? x : 0.f); }
};
template <>
struct Elem<_Float16> {
    typedef f16x8 x8;
    typedef f16x4 x4;
    static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    // stored activations saturate at the largest finite half instead of becoming inf (fp16's range is the one thing
    // bf16 has over it; a trained net's activations are orders of magnitude below it)
    static __device__ __forceinline__ _Float16 cvt(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
    static __device__ __forceinline__ _Float16 relu_cvt(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, 0.0f, 65504.0f); }  // ReLU and saturation in one v_med3
};


// LDS activation tile, plane-major: [k-chunk kc (32 channels = 64 B)][row][64 B].  Inside a 64-B row segment the
// four 16-B pieces q are rotated by 2*((row>>2)&1):  piece q of row r sits at (q + 2*((r>>2)&1)) & 3.  With the
// 16x16x32 operand layout (lane = 16 rows x 4 pieces) and ds_read_b128's lane groups (each pairs 8 rows of piece q
// with the other 8 rows of piece q^1) every group then touches 16 distinct 16-B slots, for any row shift; padding
// alone cannot do that (it always leaves a 2-way conflict per group: 48 % of LDS cycles in the first version, PMC).
// kc is an additive plane offset, so the reads of a k-step are one base register + immediates.
// Rows RT*16 .. RT*16+7 of every plane are zero: an off-board tap reads zero row (source row & 7), i.e. the
// same slot it would have used on the board.
constexpr int LDS_ROWB = 64;
constexpr int FILTERS = 256;
__device__ __forceinline__ int lds_piece(int row, int q) { return ((q + 2 * ((row >> 2) & 1)) & 3) << 4; }
// LAYOUT 1 = the plane-major image above.  LAYOUT 0 = row-major rows of 512 B + 16 B pad (the first version,
// kept for in-process A/B: tz_debug_conv_bench).
template <int LAYOUT>
struct LdsImg {
    static constexpr int ROW_PAD = 528;
    __host__ __device__ static constexpr size_t bytes(int lrows) { return LAYOUT ? (size_t)lrows * LDS_ROWB * 8 : (size_t)lrows * ROW_PAD; }
    // byte address of 16-B chunk `ci` (0..31) of row `row`
    __device__ static __forceinline__ int store_addr(int row, int ci, int plane_bytes) {
        return LAYOUT ? (ci >> 2) * plane_bytes + row * LDS_ROWB + lds_piece(row, ci & 3) : row * ROW_PAD + ci * 16;
    }
    // fragment base of source row v (or the zero row when !ok) for lane piece q; chunk kc adds kstep(kc)
    __device__ static __forceinline__ int read_base(bool ok, int v, int q, int zrow) {
        if (LAYOUT) {
            const int sr = ok ? v : zrow + (v & 7);
            return sr * LDS_ROWB + lds_piece(sr, q);
        }
        return (ok ? v : zrow) * ROW_PAD + q * 16;
    }
};

// TZ_PREC_F16C6 (tz_nn_c6.hip): place of real channel 32 w + o in its image - plane 4 (w / 4) + o / 8, piece w % 4, element o % 8 -
// and the channel at a place; the weights' input channels are permuted to match on the host (build_layer)
__host__ __device__ constexpr int c6_position_of(int c) {
    const int w = c >> 5, o = c & 31;
    return 32 * (4 * (w >> 2) + (o >> 3)) + 8 * (w & 3) + (o & 7);
}
__host__ __device__ constexpr int c6_channel_at(int pos) {
    const int plane = pos >> 5, piece = (pos >> 3) & 3, e = pos & 7;
    return 32 * (4 * (plane >> 2) + piece) + 8 * (plane & 3) + e;
}
static_assert(c6_channel_at(c6_position_of(77)) == 77 && c6_position_of(c6_channel_at(200)) == 200 && c6_position_of(32) == 8 && c6_position_of(8) == 32, "c6 channel places");

__host__ __device__ constexpr int ppt_for(int nb) { return nb == 1 ? 64 : nb == 3 ? 16 : nb == 4 ? 12 : nb == 5 ? 8 : 4; }

__device__ __forceinline__ void default_reserves_d(int n, int& stones, int& caps) {
    stones = n == 3 ? 10 : n == 4 ? 15 : n == 5 ? 21 : 30;
    caps = n >= 5 ? 1 : 0;
}

// value of input plane `c` at square `px` of state s  (game_repr, repr.rs:169-228)
template <int NB>
__device__ __forceinline__ float plane_value(const tz_state* s, int px, int c, int flat_diff) {
    constexpr int SS = 3 + (NB - 1) + (NB + 1), NN = NB * NB;
    const int to_move = s->to_move;
    if (c < 2 * SS) {
        const int side = c / SS, k = c % SS;
        const int top = s->top[px], h = s->height[px];
        if (top == TZ_EMPTY) return 0.0f;
        const unsigned long long colors = s->colors[px];
        if (k < 3) {
            const int topc = (int)((colors >> (h - 1)) & 1ull);
            return (top == k + 1 && (topc != to_move) == (side == 1)) ? 1.0f : 0.0f;
        }
        const int idx = h - 2 - (k - 3);
        if (idx < 0) return 0.0f;
        const int color = (int)((colors >> idx) & 1ull);
        return ((color != to_move) == (side == 1)) ? 1.0f : 0.0f;
    }
    const int e = c - 2 * SS;
    int ds, dc;
    default_reserves_d(NB, ds, dc);
    const int mine = to_move, other = 1 - to_move;
    switch (e) {
        case 0: return (float)s->stones[mine] / (float)ds;
        case 1: return dc ? (float)s->caps[mine] / (float)dc : 0.0f;
        case 2: return (float)s->stones[other] / (float)ds;
        case 3: return dc ? (float)s->caps[other] / (float)dc : 0.0f;
        case 4: return to_move == 1 ? 1.0f : 0.0f;
        case 5: return ((float)flat_diff - (float)s->half_komi / 2.0f) / (float)NN;
        default: return 0.0f;
    }
}

template <int NB>
__device__ __forceinline__ int state_flat_diff(const tz_state* s) {
    int d = 0;
    for (int i = 0; i < NB * NB; i++) {
        if (s->top[i] == TZ_FLAT) {
            const int col = (int)((s->colors[i] >> (s->height[i] - 1)) & 1ull);
            d += col == 0 ? 1 : -1;
        }
    }
    return d;
}

struct ConvArgs {
    const void* in;
    const tz_state* states;
    const int32_t* game_index;
    const int32_t* count_dev;
    int count_host;
    const uint16_t* w;
    const float* bias;
    const void* residual;
    void* out;
    int cin_pad, kc_total, ct_total, out_stride, cin_real;
    int relu, out_f32, has_res;
    // grouped launch (blockIdx.z = group): independent layers of the same shape, e.g. the two RND networks
    int in_stride;          // elements per input row (>= cin_pad)
    int groups;             // grid.z
    int in_z, w_z_frags, bias_z, out_z;  // per-group offsets: input elements, weight fragments, bias floats, output elements
};

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on MFMA.  Tile = P boards x (128*RN) output channels per workgroup,
// 8 waves split the output channels (16*RN each) so each wave's weight fragments are private and
// come straight from global memory; the activation tile is shared through LDS.
// per-tap LDS byte offsets of a lane's 16-row fragments (row-shifted reads; off-board taps -> zero row)
template <int NB, int RT, int TAPS, int LAYOUT>
__device__ __forceinline__ void tap_bases(const int (&yx)[RT], int tap, int lr, int q, int zrow, int (&abase)[RT]) {
    const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int y = (yx[rt] & 0xff) + dy, x = (yx[rt] >> 8) + dx;
        const bool ok = y >= 0 && y < NB && x >= 0 && x < NB;
        abase[rt] = LdsImg<LAYOUT>::read_base(ok, rt * 16 + lr + dy * NB + dx, q, zrow);
    }
}

template <int NB, int RT, int TAPS, int LAYOUT>
__device__ __forceinline__ void tap_bases_rc(int tap, int lr, int q, int rows, int zrow, int (&abase)[RT]) {
    constexpr int NN = NB * NB;
    const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int r = rt * 16 + lr, px = r % NN;
        const int y = px / NB + dy, x = px % NB + dx;
        const bool ok = r < rows && y >= 0 && y < NB && x >= 0 && x < NB;
        abase[rt] = LdsImg<LAYOUT>::read_base(ok, r + dy * NB + dx, q, zrow);
    }
}

// ABL (diagnostic builds only, tz_debug_conv_bench): 1 = no LDS fragment reads, 2 = no weight loads, 3 = neither.
// SINGLE: the whole reduction is one 256-channel slice (tower and policy convs): unrolled tile loader,
// weight fragments prefetched two k-steps ahead.
// NW = waves per workgroup; a workgroup covers NW*RN*16 output channels of P boards.
template <int NB, int P, int RN, int TAPS, bool FROM_STATE, int ABL = 0, bool SINGLE = false, int NW = 8, int LAYOUT = 1, typename ET = __bf16, int OCC = 2>
__global__ __launch_bounds__(NW * 64, OCC) void conv_mfma_kernel(ConvArgs a) {  // OCC = 2 waves per SIMD: <= 256 registers
    typedef typename Elem<ET>::x8 ex8;
    typedef typename Elem<ET>::x4 ex4;
    constexpr int NT = NW * 64;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = (ROWS + 15) / 16, LROWS = RT * 16 + 8, ZROW = RT * 16;
    constexpr int PLANE = LROWS * LDS_ROWB;
    constexpr int KSTEP = LAYOUT ? PLANE : 64;  // byte distance between consecutive 32-channel chunks of a row
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int count = a.count_dev ? *a.count_dev : a.count_host;
    const int pos0 = blockIdx.x * P;
    if (pos0 >= count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: weight addresses stay in SGPRs
    const int q = lane >> 4, lr = lane & 15;
    const int valid_rows = min(ROWS, (count - pos0) * NN);
    const size_t m0 = (size_t)pos0 * NN;
    const int ct0 = (blockIdx.y * NW + wave) * RN;
    // fragment (tap, k-chunk, col tile) = 64 lanes x 16 B.  Buffer loads: descriptor + scalar fragment
    // offset in SGPRs, one VGPR (lane*16) for all of them -> no per-fragment 64-bit address registers.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.w), 0, (a.groups > 1 ? a.groups * a.w_z_frags : TAPS * a.kc_total * a.ct_total) * 1024, 0x00020000);
    const int lane16 = lane * 16;
    auto wload = [&](int tap, int kcg, int j) -> ex8 {
        const int frag = (int)blockIdx.z * a.w_z_frags + (tap * a.kc_total + kcg) * a.ct_total + (ct0 + j);
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, 0);
        return __builtin_bit_cast(ex8, r);
    };
    f32x4 acc[RT][RN];

    if constexpr (SINGLE) {
        static_assert(!FROM_STATE, "SINGLE is for bf16 activations");
        {   // every load of the tile is in flight before the first LDS store: one memory round trip
            const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in);
            constexpr int NLOAD = (LROWS * 32 + NT - 1) / NT;
            uint4 v[NLOAD];
#pragma unroll
            for (int i = 0; i < NLOAD; i++) {
                const int id = tid + i * NT, row = id >> 5, ci = id & 31;
                v[i] = make_uint4(0, 0, 0, 0);
                if (row < valid_rows) v[i] = *reinterpret_cast<const uint4*>(in + (m0 + row) * 256 + ci * 8);
            }
#pragma unroll
            for (int i = 0; i < NLOAD; i++) {
                const int id = tid + i * NT, row = id >> 5, ci = id & 31;
                if (row < LROWS) *reinterpret_cast<uint4*>(lds + LdsImg<LAYOUT>::store_addr(row, ci, PLANE)) = v[i];
            }
        }
        // weight ring: 4 slots, slot = kc & 3 (compile time, 8 % 4 == 0), filled two k-steps ahead
        ex8 bq[4][RN];
#pragma unroll
        for (int j = 0; j < RN; j++) {
            bq[0][j] = wload(0, 0, j);
            bq[1][j] = wload(0, 1, j);
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int j = 0; j < RN; j++) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Software pipeline over the 72 k-steps: the activation fragment of row tile rt for step s+1 is
        // loaded into the register that the MFMAs of step s have just consumed, so every ds_read has a
        // whole step to land; the weight fragments of step s+2 are issued at the top of step s.
        int abase[RT];
        tap_bases_rc<NB, RT, TAPS, LAYOUT>(0, lr, q, ROWS, ZROW, abase);
        ex8 av[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
        for (int tap = 0; tap < TAPS; tap++) {
#pragma unroll
            for (int kc = 0; kc < 8; kc++) {
                if constexpr ((ABL & 2) == 0) {
                    if (kc + 2 < 8) {
#pragma unroll
                        for (int j = 0; j < RN; j++) bq[(kc + 2) & 3][j] = wload(tap, kc + 2, j);
                    } else if (tap + 1 < TAPS) {
#pragma unroll
                        for (int j = 0; j < RN; j++) bq[(kc + 2) & 3][j] = wload(tap + 1, kc + 2 - 8, j);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the issue point: hipcc otherwise sinks the loads to their use
                }
                if (kc == 7) {
                    // opaque copy: stops LICM from hoisting the per-row y/x decomposition out of the tap loop
                    // (26 loop-invariant VGPRs that would be spilled)
                    int lr_t = lr;
                    asm volatile("" : "+v"(lr_t));
                    tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap + 1 < TAPS ? tap + 1 : tap, lr_t, q, ROWS, ZROW, abase);
                }
                if (LAYOUT == 1 && kc == 4) {  // ds_read immediates reach 64 KB: rebase once per tap for planes 5..7
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        abase[rt] += 4 * PLANE;
                        asm volatile("" : "+v"(abase[rt]));
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                    for (int j = 0; j < RN; j++)
                        acc[rt][j] = Elem<ET>::mfma(bq[kc & 3][j], av[rt], acc[rt][j]);
                    if constexpr ((ABL & 1) == 0) {
                        if (kc < 7) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1 - (LAYOUT == 1 && kc >= 4 ? 4 : 0)) * KSTEP);
                        else av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                    }
                }
            }
        }
    } else {
        int yx[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int r = rt * 16 + lr, px = r % NN;
            yx[rt] = r < ROWS ? ((px / NB) | ((px % NB) << 8)) : 0x7f7f;
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int j = 0; j < RN; j++) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nslices = (a.cin_pad + 255) / 256;
        // activation slices (256 input channels each) travel global -> registers -> LDS; the registers of slice s+1 are
        // filled while slice s is being multiplied, so only the first slice's memory round trip is exposed
        constexpr int NLOAD = FROM_STATE ? 1 : (LROWS * 32 + NT - 1) / NT;
        uint4 stage[NLOAD];
        auto fetch_slice = [&](int slice) {
            const int cpr_s = min(256, a.cin_pad - slice * 256) / 8;
            const uint16_t* in = reinterpret_cast<const uint16_t*>(a.in) + (size_t)blockIdx.z * a.in_z + slice * 256;
#pragma unroll
            for (int i = 0; i < NLOAD; i++) {
                const int id = tid + i * NT, row = id >> 5, ci = id & 31;
                stage[i] = make_uint4(0, 0, 0, 0);
                if (row < valid_rows && ci < cpr_s) stage[i] = *reinterpret_cast<const uint4*>(in + (m0 + row) * a.in_stride + ci * 8);
            }
        };
        if constexpr (!FROM_STATE) fetch_slice(0);
        for (int slice = 0; slice < nslices; slice++) {
            const int cs = min(256, a.cin_pad - slice * 256);
            const int kcs = cs / 32, cpr = cs / 8;
            __syncthreads();
            if constexpr (FROM_STATE) {
                // game_repr fused into the tile loader: one thread per (board, square)
                for (int row = tid; row < LROWS; row += NT) {
                    const bool ok = row < valid_rows;
                    const tz_state* s = nullptr;
                    int px = 0, fd = 0;
                    if (ok) {
                        const int pos = pos0 + row / NN;
                        px = row % NN;
                        s = a.states + (a.game_index ? a.game_index[pos] : pos);
                        fd = state_flat_diff<NB>(s);
                    }
                    for (int c8 = 0; c8 < cpr; c8++) {
                        ex8 v;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int c = c8 * 8 + k;
                            v[k] = (ET)((ok && c < a.cin_real) ? plane_value<NB>(s, px, c, fd) : 0.0f);
                        }
                        *reinterpret_cast<ex8*>(lds + LdsImg<LAYOUT>::store_addr(row, c8, PLANE)) = v;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NLOAD; i++) {
                    const int id = tid + i * NT, row = id >> 5, ci = id & 31;
                    if (row < LROWS && ci < cpr) *reinterpret_cast<uint4*>(lds + LdsImg<LAYOUT>::store_addr(row, ci, PLANE)) = stage[i];
                }
                if (slice + 1 < nslices) fetch_slice(slice + 1);
            }
            __syncthreads();
            ex8 bnext[RN];
#pragma unroll
            for (int j = 0; j < RN; j++) bnext[j] = wload(0, slice * 8, j);
            if constexpr (TAPS == 1) {
                // linear layers (the RND MLP): fragments of k-step kc+1 are loaded into the registers the MFMAs of
                // k-step kc have just consumed, weights one k-step ahead
                int abase[RT];
                tap_bases<NB, RT, TAPS, LAYOUT>(yx, 0, lr, q, ZROW, abase);
                ex8 av[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                for (int kc = 0; kc < kcs; kc++) {
                    ex8 bcur[RN];
#pragma unroll
                    for (int j = 0; j < RN; j++) bcur[j] = bnext[j];
                    if (kc + 1 < kcs) {
#pragma unroll
                        for (int j = 0; j < RN; j++) bnext[j] = wload(0, slice * 8 + kc + 1, j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                        for (int j = 0; j < RN; j++) acc[rt][j] = Elem<ET>::mfma(bcur[j], av[rt], acc[rt][j]);
                        if (kc + 1 < kcs) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1) * KSTEP);
                    }
                }
                continue;
            }
            for (int tap = 0; tap < TAPS; tap++) {
                int abase[RT];
                tap_bases<NB, RT, TAPS, LAYOUT>(yx, tap, lr, q, ZROW, abase);
                for (int kc = 0; kc < kcs; kc++) {
                    ex8 bcur[RN];
#pragma unroll
                    for (int j = 0; j < RN; j++) bcur[j] = bnext[j];
                    if (kc + 1 < kcs) {
#pragma unroll
                        for (int j = 0; j < RN; j++) bnext[j] = wload(tap, slice * 8 + kc + 1, j);
                    } else if (tap + 1 < TAPS) {
#pragma unroll
                        for (int j = 0; j < RN; j++) bnext[j] = wload(tap + 1, slice * 8, j);
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const ex8 av = *reinterpret_cast<const ex8*>(lds + abase[rt] + kc * KSTEP);
#pragma unroll
                        for (int j = 0; j < RN; j++)
                            acc[rt][j] = Elem<ET>::mfma(bcur[j], av, acc[rt][j]);
                    }
                }
            }
        }
    }

    // epilogue: D[row = channel (q*4 + reg), col = pixel (lr)]
#pragma unroll
    for (int j = 0; j < RN; j++) {
        const int cbase = (ct0 + j) * 16 + q * 4;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + (size_t)blockIdx.z * a.bias_z + cbase);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int r = rt * 16 + lr;
            if (r >= valid_rows) continue;
            f32x4 v = acc[rt][j] + bias;
            const size_t o = (size_t)blockIdx.z * a.out_z + (m0 + r) * a.out_stride + cbase;
            if (a.has_res) {
                const ex4 rv = *reinterpret_cast<const ex4*>(reinterpret_cast<const uint16_t*>(a.residual) + o);
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] += (float)rv[k];
            }
            if (a.relu) {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = v[k] > 0.f ? v[k] : 0.f;
            }
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + o) = v;
            } else {
                ex4 ov;
#pragma unroll
                for (int k = 0; k < 4; k++) ov[k] = Elem<ET>::cvt(v[k]);
                *reinterpret_cast<ex4*>(reinterpret_cast<uint16_t*>(a.out) + o) = ov;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The whole residual tower in ONE launch (residual.rs:13-63 x blocks).  A workgroup keeps its P boards'
// activations in LDS across all 2*blocks convolutions: after a conv the accumulators are ReLU'd, packed to
// bf16 and written back into the same LDS image as the next conv's input; the block input x is re-read from
// LDS into the accumulators (acc = x + bias) before it is overwritten, so the residual add costs no global
// traffic and no registers.  Only the tower's input tile is loaded and only its final output is stored:
// the per-layer tile-load / epilogue phases of the one-launch-per-conv form (~25 % of its time, all CUs hitting
// HBM at once) disappear, as do 2*blocks-1 launch boundaries.  K-loop identical to conv_mfma_kernel<SINGLE>.
struct TowerArgs {
    const uint16_t* in;
    uint16_t* out;
    const uint16_t* w;   // layers back to back, each [9][8][16][64 lanes][8] bf16
    const float* bias;   // [layers][256]
    const int32_t* count_dev;
    int count_host;
    int nlayers;         // 2 * blocks
};

// OPT: A/B and ablation switches for tz_debug_tower_bench (tools/tower_bench.py); tower_bf16 ships 4224 = 4096 | 128.
//   1     fetch the next layer's first weight fragments during the last two k-steps            (no gain)
//   2, 4  stagger the two waves of a SIMD by s_sleep 4 / 10                                      (no gain)
//   8     weight prefetch distance 3 instead of 2                                                (no gain)
//   16, 32, 64   ablations: no activation-fragment reads / no weight stream / no layer epilogue  (timing only)
//   128   pinned issue order: MFMAs of a row tile, then the ds_read refilling its fragment       (+1 %, shipped)
//   256, 512, 1024, 2048   other issue orders: reads one tile behind / 4:2 / VALU slots / floating weight loads (worse)
//   4096  per-tap fragment base addresses from a table in LDS instead of recomputing them        (+10 %, shipped)
//   8192  ablation: no mid-tap rebase adds                                                        (timing only, no effect)
//   16384, 32768, 65536   cache policy of the weight loads: sc0 / sc1 / nt                          (0 %, -1 %, 0 %)
template <int NB, int P, int OPT = 0, typename ET = __bf16>
__global__ __launch_bounds__(512, 2) void tower_mfma_kernel(TowerArgs a) {
    typedef typename Elem<ET>::x8 ex8;
    typedef typename Elem<ET>::x4 ex4;
    constexpr int RN = 2, TAPS = 9, LAYOUT = 1, NT = 512;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = (ROWS + 15) / 16, LROWS = RT * 16 + 8, ZROW = RT * 16;
    constexpr int PLANE = LROWS * LDS_ROWB, KSTEP = PLANE;
    constexpr int LAYER_FRAGS = TAPS * 8 * 16;  // 1-KB fragments per layer
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int count = a.count_dev ? *a.count_dev : a.count_host;
    const int pos0 = blockIdx.x * P;
    if (pos0 >= count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, lr = lane & 15;
    const int valid_rows = min(ROWS, (count - pos0) * NN);
    const size_t m0 = (size_t)pos0 * NN;
    const int ct0 = wave * RN;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.w), 0, a.nlayers * LAYER_FRAGS * 1024, 0x00020000);
    const int lane16 = lane * 16;
    // A/B: cache policy of the weight stream (aux bits of the buffer load: 1 = sc0, 2 = sc1, 4 = nt)
    constexpr int WAUX = (OPT & 16384) ? 1 : (OPT & 32768) ? 2 : (OPT & 65536) ? 4 : 0;
    auto wload = [&](int layer, int tap, int kc, int j) -> ex8 {
        const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, WAUX);
        return __builtin_bit_cast(ex8, r);
    };
    {   // tower input tile: every load in flight before the first LDS store
        constexpr int NLOAD = (LROWS * 32 + NT - 1) / NT;
        uint4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int id = tid + i * NT, row = id >> 5, ci = id & 31;
            v[i] = make_uint4(0, 0, 0, 0);
            if (row < valid_rows) v[i] = *reinterpret_cast<const uint4*>(a.in + (m0 + row) * 256 + ci * 8);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int id = tid + i * NT, row = id >> 5, ci = id & 31;
            if (row < LROWS) *reinterpret_cast<uint4*>(lds + LdsImg<LAYOUT>::store_addr(row, ci, PLANE)) = v[i];
        }
    }
    // OPT 4096 (A/B): the per-tap fragment base addresses of a lane (13 values per tap, the same in every layer) are
    // computed once into LDS behind the image, so a tap boundary costs 13 ds_read_b32 instead of ~130 VALU instructions
    int* tap_table = reinterpret_cast<int*>(lds + 8 * PLANE);  // [TAPS][RT][64 lanes]
    if constexpr (OPT & 4096) {
        if (wave == 0) {
            for (int tap = 0; tap < TAPS; tap++) {
                int tb[RT];
                tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap, lr, q, ROWS, ZROW, tb);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) tap_table[(tap * RT + rt) * 64 + lane] = tb[rt];
            }
        }
    }
    // address of this lane's 4 output channels of pixel row (rt*16 + lr) in the LDS image: its wave's 32
    // channels are plane `wave`; + rt * 1024
    int obase[RN];
#pragma unroll
    for (int j = 0; j < RN; j++)
        obase[j] = wave * PLANE + lr * LDS_ROWB + lds_piece(lr, j * 2 + (q >> 1)) + (q & 1) * 8;

    f32x4 acc[RT][RN];
    ex8 bq[4][RN];  // weight ring; slots 0/1 of the next layer are fetched during the last two k-steps of this one
    for (int layer = 0; layer < a.nlayers; layer++) {
        if ((layer & 1) == 0) {  // first conv of a block starts from its bias; the second from x + bias (below)
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + layer * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) acc[rt][j] = b4;
            }
        }
        constexpr int PD = (OPT & 8) ? 3 : 2;  // weight prefetch distance in k-steps (slot = k-step & 3)
        if (layer == 0 || !(OPT & 1)) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                bq[0][j] = wload(layer, 0, 0, j);
                bq[1][j] = wload(layer, 0, 1, j);
                if (PD == 3) bq[2][j] = wload(layer, 0, 2, j);
            }
        }
        __syncthreads();  // the LDS image of this layer's input is complete
        // The two waves that share a SIMD (w and w+4) run the same program; delayed by part of a k-step, the
        // VALU-only stretch at each tap boundary of one falls into the MFMA stretch of the other.
        if constexpr (OPT & 2) {
            if (wave >= 4) __builtin_amdgcn_s_sleep(4);
        }
        if constexpr (OPT & 4) {
            if (wave >= 4) __builtin_amdgcn_s_sleep(10);
        }
        int abase[RT];
        if constexpr (OPT & 4096) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) abase[rt] = tap_table[rt * 64 + lane];
        } else {
            int lr_t = lr;
            asm volatile("" : "+v"(lr_t));
            tap_bases_rc<NB, RT, TAPS, LAYOUT>(0, lr_t, q, ROWS, ZROW, abase);
        }
        ex8 av[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
        for (int tap = 0; tap < TAPS; tap++) {
#pragma unroll
            for (int kc = 0; kc < 8; kc++) {
                if constexpr (OPT & 32) {  // ablation: no weight stream
                } else if (kc + PD < 8) {
#pragma unroll
                    for (int j = 0; j < RN; j++) bq[(kc + PD) & 3][j] = wload(layer, tap, kc + PD, j);
                } else if (tap + 1 < TAPS) {
#pragma unroll
                    for (int j = 0; j < RN; j++) bq[(kc + PD) & 3][j] = wload(layer, tap + 1, kc + PD - 8, j);
                } else if ((OPT & 1) && layer + 1 < a.nlayers) {  // next layer's first fragments ride under this layer's epilogue
#pragma unroll
                    for (int j = 0; j < RN; j++) bq[(kc + PD) & 3][j] = wload(layer + 1, 0, kc + PD - 8, j);
                }
                if constexpr (!(OPT & 2048)) __builtin_amdgcn_sched_barrier(0);
                if (kc == 7) {
                    if constexpr (OPT & 4096) {
                        const int nt = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) abase[rt] = tap_table[(nt * RT + rt) * 64 + lane];
                    } else {
                        int lr_t = lr;
                        asm volatile("" : "+v"(lr_t));
                        tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap + 1 < TAPS ? tap + 1 : tap, lr_t, q, ROWS, ZROW, abase);
                    }
                }
                if (kc == 4 && !(OPT & 8192)) {  // OPT 8192 (timing ablation only, wrong planes): no mid-tap rebase
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        abase[rt] += 4 * PLANE;
                        asm volatile("" : "+v"(abase[rt]));
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                    for (int j = 0; j < RN; j++)
                        acc[rt][j] = Elem<ET>::mfma(bq[kc & 3][j], av[rt], acc[rt][j]);
                    if constexpr (OPT & 16) {  // ablation: no activation-fragment reads
                    } else if (kc < 7) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1 - (kc >= 4 ? 4 : 0)) * KSTEP);
                    else if (tap + 1 < TAPS) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                }
                if constexpr (OPT & 128) {  // A/B: pin the issue order of a k-step to 2 MFMA : 1 ds_read, row tile by row tile
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if constexpr (OPT & 1024) {  // A/B: 2 MFMA : 1 ds_read : up to 3 VALU (address arithmetic at tap boundaries spread out)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    }
                }
                if constexpr (OPT & 2048) {  // A/B: no hard barrier after the weight loads; they are placed a third and two thirds in
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (rt == RT / 3 || rt == 2 * RT / 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if constexpr (OPT & 512) {  // A/B: two row tiles at a time: 4 MFMA : 2 ds_read
#pragma unroll
                    for (int rt = 0; rt + 1 < RT; rt += 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    if (RT & 1) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if constexpr (OPT & 256) {  // A/B: the same with each ds_read one row tile behind the MFMAs that free its register
                    __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
#pragma unroll
                    for (int rt = 1; rt < RT; rt++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        // ---- layer epilogue.  A wave's 32 output channels are exactly one plane of the LDS image, and it is the
        // only wave that reads that plane's residual: after one barrier (all K-loops done) each lane can
        // replace its own 8 bytes in place - read x (for the residual), write ReLU(acc) - with no staging.
        if (layer + 1 == a.nlayers) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const int cbase = (ct0 + j) * 16 + q * 4;
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const int r = rt * 16 + lr;
                    ex4 pk;
#pragma unroll
                    for (int k = 0; k < 4; k++) pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                    if (r < valid_rows) *reinterpret_cast<ex4*>(a.out + (m0 + r) * FILTERS + cbase) = pk;
                }
            }
            break;
        }
        __syncthreads();  // every wave is done reading the old image
        const bool to_second = (layer & 1) == 0;  // next conv is the block's second: it starts from x + bias
        if constexpr (OPT & 64) continue;  // ablation: no epilogue
#pragma unroll
        for (int j = 0; j < RN; j++) {
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (to_second) b4 = *reinterpret_cast<const f32x4*>(a.bias + (layer + 1) * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                ex4* slot = reinterpret_cast<ex4*>(lds + obase[j] + rt * 16 * LDS_ROWB);
                ex4 pk;
#pragma unroll
                for (int k = 0; k < 4; k++) pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                if (to_second) {
                    const ex4 xv = *slot;
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[rt][j][k] = (float)xv[k] + b4[k];
                }
                *slot = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// A/B twin of the tower (tz_debug_tower_bench variant 100000): the 8 waves are 4 channel groups (64 output channels =
// 4 column tiles each) x 2 row halves (row tiles 0..6 and 7..12 for 13 tiles), so that an activation fragment read from
// LDS feeds 4 MFMAs instead of 2 (half the ds_read bytes) at 28 accumulator tiles per wave; the two waves of a channel
// group fetch the same weight fragments (the second fetch comes from the vector L1).  Waves w and w+4 share a SIMD and
// are one wave of each half, so the SIMD's MFMA count per k-step is unchanged.
#ifdef TZ_ABLATIONS   // the 4 channel groups x 2 row halves twin of the tower (measured 3 % slower): diagnostic builds only
template <int NB, int P, int RTW, typename ET>
__device__ __forceinline__ void tower4x2_body(const TowerArgs& a, unsigned char* lds, const int* tap_table, int lane, int cg, int tile0,
                                              int valid_rows, size_t m0) {
    typedef typename Elem<ET>::x8 ex8;
    typedef typename Elem<ET>::x4 ex4;
    constexpr int RN = 4, TAPS = 9;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = (ROWS + 15) / 16, LROWS = RT * 16 + 8;
    constexpr int PLANE = LROWS * LDS_ROWB;
    constexpr int LAYER_FRAGS = TAPS * 8 * 16;
    const int q = lane >> 4, lr = lane & 15;
    const int ct0 = cg * RN;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.w), 0, a.nlayers * LAYER_FRAGS * 1024, 0x00020000);
    const int lane16 = lane * 16;
    auto wload = [&](int layer, int tap, int kc, int j) -> ex8 {
        const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
        return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, 0));
    };
    int obase[RN];  // this lane's 4 channels of column tile j, row lr of row tile tile0
#pragma unroll
    for (int j = 0; j < RN; j++)
        obase[j] = (cg * 2 + (j >> 1)) * PLANE + (tile0 * 16 + lr) * LDS_ROWB + lds_piece(lr, (j & 1) * 2 + (q >> 1)) + (q & 1) * 8;
    f32x4 acc[RTW][RN];
    ex8 bq[4][RN];
    for (int layer = 0; layer < a.nlayers; layer++) {
        if ((layer & 1) == 0) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + layer * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
                for (int rt = 0; rt < RTW; rt++) acc[rt][j] = b4;
            }
        }
#pragma unroll
        for (int j = 0; j < RN; j++) {
            bq[0][j] = wload(layer, 0, 0, j);
            bq[1][j] = wload(layer, 0, 1, j);
        }
        __syncthreads();
        int abase[RTW];
#pragma unroll
        for (int rt = 0; rt < RTW; rt++) abase[rt] = tap_table[(tile0 + rt) * 64 + lane];
        ex8 av[RTW];
#pragma unroll
        for (int rt = 0; rt < RTW; rt++) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
        for (int tap = 0; tap < TAPS; tap++) {
#pragma unroll
            for (int kc = 0; kc < 8; kc++) {
                if (kc + 2 < 8) {
#pragma unroll
                    for (int j = 0; j < RN; j++) bq[(kc + 2) & 3][j] = wload(layer, tap, kc + 2, j);
                } else if (tap + 1 < TAPS) {
#pragma unroll
                    for (int j = 0; j < RN; j++) bq[(kc + 2) & 3][j] = wload(layer, tap + 1, kc + 2 - 8, j);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == 7) {
                    const int nt = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
                    for (int rt = 0; rt < RTW; rt++) abase[rt] = tap_table[(nt * RT + tile0 + rt) * 64 + lane];
                }
                if (kc == 4) {
#pragma unroll
                    for (int rt = 0; rt < RTW; rt++) {
                        abase[rt] += 4 * PLANE;
                        asm volatile("" : "+v"(abase[rt]));
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RTW; rt++) {
#pragma unroll
                    for (int j = 0; j < RN; j++) acc[rt][j] = Elem<ET>::mfma(bq[kc & 3][j], av[rt], acc[rt][j]);
                    if (kc < 7) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1 - (kc >= 4 ? 4 : 0)) * PLANE);
                    else if (tap + 1 < TAPS) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                }
#pragma unroll
                for (int rt = 0; rt < RTW; rt++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, RN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        if (layer + 1 == a.nlayers) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const int cbase = (ct0 + j) * 16 + q * 4;
#pragma unroll
                for (int rt = 0; rt < RTW; rt++) {
                    const int r = (tile0 + rt) * 16 + lr;
                    ex4 pk;
#pragma unroll
                    for (int k = 0; k < 4; k++) pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                    if (r < valid_rows) *reinterpret_cast<ex4*>(a.out + (m0 + r) * FILTERS + cbase) = pk;
                }
            }
            break;
        }
        __syncthreads();
        const bool to_second = (layer & 1) == 0;
#pragma unroll
        for (int j = 0; j < RN; j++) {
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (to_second) b4 = *reinterpret_cast<const f32x4*>(a.bias + (layer + 1) * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RTW; rt++) {
                ex4* slot = reinterpret_cast<ex4*>(lds + obase[j] + rt * 16 * LDS_ROWB);
                ex4 pk;
#pragma unroll
                for (int k = 0; k < 4; k++) pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                if (to_second) {
                    const ex4 xv = *slot;
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[rt][j][k] = (float)xv[k] + b4[k];
                }
                *slot = pk;
            }
        }
    }
}

template <int NB, int P, typename ET = __bf16>
__global__ __launch_bounds__(512, 2) void tower4x2_mfma_kernel(TowerArgs a) {
    constexpr int TAPS = 9, LAYOUT = 1, NT = 512;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = (ROWS + 15) / 16, LROWS = RT * 16 + 8, ZROW = RT * 16;
    constexpr int PLANE = LROWS * LDS_ROWB;
    constexpr int RT_A = (RT + 1) / 2, RT_B = RT - RT_A;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int count = a.count_dev ? *a.count_dev : a.count_host;
    const int pos0 = blockIdx.x * P;
    if (pos0 >= count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, lr = lane & 15;
    const int valid_rows = min(ROWS, (count - pos0) * NN);
    const size_t m0 = (size_t)pos0 * NN;
    {
        constexpr int NLOAD = (LROWS * 32 + NT - 1) / NT;
        uint4 v[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int id = tid + i * NT, row = id >> 5, ci = id & 31;
            v[i] = make_uint4(0, 0, 0, 0);
            if (row < valid_rows) v[i] = *reinterpret_cast<const uint4*>(a.in + (m0 + row) * 256 + ci * 8);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; i++) {
            const int id = tid + i * NT, row = id >> 5, ci = id & 31;
            if (row < LROWS) *reinterpret_cast<uint4*>(lds + LdsImg<LAYOUT>::store_addr(row, ci, PLANE)) = v[i];
        }
    }
    int* tap_table = reinterpret_cast<int*>(lds + 8 * PLANE);
    if (wave == 0) {
        for (int tap = 0; tap < TAPS; tap++) {
            int tb[RT];
            tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap, lr, q, ROWS, ZROW, tb);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) tap_table[(tap * RT + rt) * 64 + lane] = tb[rt];
        }
    }
    if (wave < 4) tower4x2_body<NB, P, RT_A, ET>(a, lds, tap_table, lane, wave, 0, valid_rows, m0);
    else tower4x2_body<NB, P, RT_B, ET>(a, lds, tap_table, lane, wave - 4, RT_A, valid_rows, m0);
}

#endif  // TZ_ABLATIONS

// ---------------------------------------------------------------------------------------------
// The whole trunk and its heads in ONE launch: game_repr + first conv (net5.rs:46-64), the residual tower, the policy
// conv (net5.rs:75-87) and the value / UBE heads (net5.rs:89-120).  Same structure as tower_mfma_kernel; what
// enters is the packed game states, what leaves is the policy tensor (fp32), value and UBE.  The block input of the
// first conv are the planes built in LDS planes 0..kc_in-1; the policy conv and the heads read the tower's last
// output from the LDS image.
struct NetArgs {
    const tz_state* states;
    const int32_t* game_index;
    const int32_t* count_dev;
    int count_host;
    const uint16_t* w_in;   // [9][kc_in][16][64][8]
    const float* bias_in;   // [256]
    int cin_real, kc_in;
    const uint16_t* w;      // tower layers back to back
    const float* bias;      // [nlayers][256]
    int nlayers;
    const uint16_t* w_pol;  // [9][8][8*RNP][64][8]
    const float* bias_pol;  // [pol_stride]
    float* policy_out;      // [rows][pol_stride]
    int pol_stride;
    const float* heads;     // heads_kernel layout
    float* value;
    float* ube;
    void* rnd_in;           // if set: the RND networks' input x / sum(x^2) (net5.rs:127) is written here, [pos][rnd_stride]
    int rnd_stride;         // elements; index inside a position = square * cin_real + plane (cin_real % 8 == 0)
    // split precision (SP = 1): the lo halves of the three weight buffers, same fragment order
    const uint16_t* w_in_lo;
    const uint16_t* w_lo;
    const uint16_t* w_pol_lo;
    // FP8 corrections (SP = 2): FP8 fragments of the tower and of the policy conv, and what their correction accumulators are
    // multiplied by ([nlayers] tower layers, then the policy conv)
    const unsigned char* w8;
    const unsigned char* w_pol8;
    const float* c8_scales;
    unsigned long long* dbg;   // diagnostic builds (ABL & 8): [workgroup][4] = memtime, memrealtime before / after the tower
    // TZ_PREC_F16C6 (tz_nn_c6.hip): w8 / w_pol8 hold its E2M3 records; a block's input (fp32) waits here for the block's second conv
    void* seeds;
    // SPLIT > 1 (several CUs per board group, small batches through tz_net_eval): the groups' exchange buffer [group][2][8 planes] and
    // their arrival counters (one 128-byte line each, zeroed before the launch)
    unsigned char* xch;
    unsigned* xch_count;
};

// 72 k-steps of one 256-input-channel 3x3 conv out of the LDS image: activation fragments one k-step ahead,
// weight fragments two k-steps ahead through a 4-slot ring (see tower_mfma_kernel)
// tap_table[tap][rt][lane]: the lane's fragment base address for row tile rt under tap `tap` (tap_bases_rc, computed once
// per kernel into LDS: a tap boundary then costs RT ds_read_b32 instead of ~10 VALU instructions per row tile —
// measured on the tower twin, `tools/tower_bench.py 128 4224`: 3.376 -> 3.065 ms)
// PD = weight prefetch distance in k-steps: 2 (ring of 4) for the full-size workgroups, whose k-steps are 26 MFMAs long;
// 6 (ring of 8) for the 1- and 2-board workgroups of small batches, whose k-steps are 4-8 MFMAs and which would
// otherwise wait for L2 at every step.
// `pre` (optional): the first PD weight fragments per column tile, [d][j], already requested by the caller (the several-CU form asks
// for the next layer's before it waits for its partners)
template <int NB, int RT, int RNX, int ROWS, int ZROW, int PLANE, typename ET, int PD = 2, typename WL>
__device__ __forceinline__ void k_loop_256(const unsigned char* lds, const int* tap_table, int lane, f32x4 (&acc)[RT][RNX], WL wl,
                                           const typename Elem<ET>::x8* pre = nullptr) {
    typedef typename Elem<ET>::x8 ex8;
    constexpr int TAPS = 9, RING = PD <= 2 ? 4 : PD <= 6 ? 8 : 16;
    static_assert(PD < RING && PD <= 14, "prefetch distance");
    ex8 bq[RING][RNX];
#pragma unroll
    for (int j = 0; j < RNX; j++) {
#pragma unroll
        for (int d = 0; d < PD; d++) bq[d][j] = pre ? pre[d * RNX + j] : wl(d / 8, d % 8, j);
    }
    int abase[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) abase[rt] = tap_table[rt * 64 + lane];
    ex8 av[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
    // a ring of 16 (the several-CU form: 12 fragments in flight per wave, which is what its weight stream lives on) has its slot
    // depend on the tap: the tap loop is unrolled with it so that every slot is a register picked at compile time
#pragma unroll(RING > 8 ? TAPS : 1)
    for (int tap = 0; tap < TAPS; tap++) {
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            {
                const int tn = tap + (kc + PD) / 8, kn = (kc + PD) % 8;
                if (tn < TAPS) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) bq[(tap * 8 + kc + PD) & (RING - 1)][j] = wl(tn, kn, j);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kc == 7) {
                const int nt = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
                for (int rt = 0; rt < RT; rt++) abase[rt] = tap_table[(nt * RT + rt) * 64 + lane];
            }
            if (kc == 4) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    abase[rt] += 4 * PLANE;
                    asm volatile("" : "+v"(abase[rt]));
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                for (int j = 0; j < RNX; j++) acc[rt][j] = Elem<ET>::mfma(bq[(tap * 8 + kc) & (RING - 1)][j], av[rt], acc[rt][j]);
                if (kc < 7) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1 - (kc >= 4 ? 4 : 0)) * PLANE);
                else if (tap + 1 < TAPS) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
            }
            // pin the issue order of the k-step: the MFMAs of a row tile, then the ds_read that refills its fragment
            // (measured on the tower twin of this loop, `tools/tower_bench.py 0 128`: 3.375 -> 3.347 ms; the scheduler's
            // own order, a one-tile lag of the reads, and 4 MFMA : 2 reads are all slower)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                __builtin_amdgcn_sched_group_barrier(0x008, RNX, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
}

// The same 72 k-steps for the several-CU form, whose k-steps are one or two row tiles long (1 or 2 boards, RNX column tiles of a
// quarter of the channels): with the activation fragment of step s + 1 requested after the MFMAs of step s, as above, a step is an
// LDS round trip (≈100 cycles for 32-64 cycles of MFMAs: 5-7 us per layer measured).  Here the loop is fully unrolled — every ring slot
// and every address offset a compile-time constant —, the activation fragments run AD steps ahead through a ring of four, the weight
// fragments PD steps ahead through a ring of 16, and the lane's nine tap bases per row tile are read from the table once.  Same
// order of accumulation per output: same bits.
template <int RT, int RNX, int PLANE, typename ET, int PD = 12, int AD = 3, typename WL>
__device__ __forceinline__ void k_loop_256_deep(const unsigned char* lds, const int* tap_table, int lane, f32x4 (&acc)[RT][RNX], WL wl,
                                                const typename Elem<ET>::x8* pre) {
    typedef typename Elem<ET>::x8 ex8;
    constexpr int TAPS = 9, STEPS = TAPS * 8, RING = 16, ARING = 4;
    static_assert(PD < RING && AD < ARING, "prefetch distances");
    int tb[TAPS][RT];
#pragma unroll
    for (int tap = 0; tap < TAPS; tap++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) tb[tap][rt] = tap_table[(tap * RT + rt) * 64 + lane];
    ex8 bq[RING][RNX];
#pragma unroll
    for (int d = 0; d < PD; d++)
#pragma unroll
        for (int j = 0; j < RNX; j++) bq[d][j] = pre ? pre[d * RNX + j] : wl(d / 8, d % 8, j);
    ex8 av[ARING][RT];
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) av[d][rt] = *reinterpret_cast<const ex8*>(lds + tb[d / 8][rt] + (d % 8) * PLANE);
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (st + PD < STEPS) {
#pragma unroll
            for (int j = 0; j < RNX; j++) bq[(st + PD) % RING][j] = wl((st + PD) / 8, (st + PD) % 8, j);
        }
        if (st + AD < STEPS) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
                av[(st + AD) % ARING][rt] = *reinterpret_cast<const ex8*>(lds + tb[(st + AD) / 8][rt] + ((st + AD) % 8) * PLANE);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int j = 0; j < RNX; j++) acc[rt][j] = Elem<ET>::mfma(bq[st % RING][j], av[st % ARING][rt], acc[rt][j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ... and for the split precision (TZ_PREC_F16X2: hi / lo halves of both operands, three MFMAs per product, see k_loop_split): the
// several-CU form of the arithmetic that holds the 1e-3 tolerance.  Same order per output as k_loop_split: wh xh into the main
// accumulator, then wl xh and wh xl into the correction accumulator, k-step by k-step.
template <int RT, int RNX, int PLANE, int PD = 6, int AD = (RT <= 4 ? 3 : 1), typename WL>
__device__ __forceinline__ void k_loop_split_deep(const unsigned char* lds, const int* tap_table, int lane, f32x4 (&accm)[RT][RNX],
                                                  f32x4 (&accc)[RT][RNX], WL wl) {
    typedef Elem<_Float16> E;
    typedef f16x8 ex8;
    constexpr int TAPS = 9, STEPS = TAPS * 8, RING = 8, ARING = 4, LO = 8 * PLANE;
    static_assert(PD < RING && AD < ARING, "prefetch distances");
    int tb[TAPS][RT];
#pragma unroll
    for (int tap = 0; tap < TAPS; tap++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) tb[tap][rt] = tap_table[(tap * RT + rt) * 64 + lane];
    ex8 bq[RING][RNX][2];
#pragma unroll
    for (int d = 0; d < PD; d++)
#pragma unroll
        for (int j = 0; j < RNX; j++) {
            bq[d][j][0] = wl(d / 8, d % 8, j, 0);
            bq[d][j][1] = wl(d / 8, d % 8, j, 1);
        }
    ex8 ah[ARING][RT], al[ARING][RT];
#pragma unroll
    for (int d = 0; d < AD; d++)
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            ah[d][rt] = *reinterpret_cast<const ex8*>(lds + tb[d / 8][rt] + (d % 8) * PLANE);
            al[d][rt] = *reinterpret_cast<const ex8*>(lds + tb[d / 8][rt] + (d % 8) * PLANE + LO);
        }
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (st + PD < STEPS) {
#pragma unroll
            for (int j = 0; j < RNX; j++) {
                bq[(st + PD) % RING][j][0] = wl((st + PD) / 8, (st + PD) % 8, j, 0);
                bq[(st + PD) % RING][j][1] = wl((st + PD) / 8, (st + PD) % 8, j, 1);
            }
        }
        if (st + AD < STEPS) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int off = tb[(st + AD) / 8][rt] + ((st + AD) % 8) * PLANE;
                ah[(st + AD) % ARING][rt] = *reinterpret_cast<const ex8*>(lds + off);
                al[(st + AD) % ARING][rt] = *reinterpret_cast<const ex8*>(lds + off + LO);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
#pragma unroll
            for (int j = 0; j < RNX; j++) accm[rt][j] = E::mfma(bq[st % RING][j][0], ah[st % ARING][rt], accm[rt][j]);
#pragma unroll
            for (int j = 0; j < RNX; j++) accc[rt][j] = E::mfma(bq[st % RING][j][1], ah[st % ARING][rt], accc[rt][j]);
#pragma unroll
            for (int j = 0; j < RNX; j++) accc[rt][j] = E::mfma(bq[st % RING][j][0], al[st % ARING][rt], accc[rt][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------
// Row order of the LDS image.  A 3x3 tap whose source square is off the board multiplies a zero row: on 5x5 that is 56 of
// the 225 (square, tap) pairs.  An MFMA covers 16 rows, so the zeros can only be skipped 16 rows at a time: with PERM the
// rows of a workgroup are ordered square-major — row = tile*16 + slot*P + board, where tile*PPT + slot = the square's
// place in an order that puts squares of the same board edge into the same tile — so that a row tile holds PPT = 16/P
// squares of all P boards and, for an edge tile, three of the nine taps are off the board for every row of it.  Those
// (tap, tile) pairs are left out of the k-loop at compile time (`tap_tile_mask`): 26 of 117 on 5x5 (P = 8), 12 of 81 on
// 6x6 (P = 4), 32 of 81 on 3x3 (P = 16); what is skipped adds exact zeros, the result is bit-identical.
// Without PERM rows are board-major (row = board*NN + square), the order of the activations in global memory.
template <int NB, int P, bool PERM>
struct RowMap {
    static constexpr int NN = NB * NB;
    static constexpr int PPT = PERM ? 16 / P : 1;
    static constexpr int SLOTS = PERM ? (NN + PPT - 1) / PPT * PPT : NN;
    static constexpr int RT = PERM ? SLOTS / PPT : (P * NN + 15) / 16;
    // the square at place k of the order (or -1: padding)
    __host__ __device__ static constexpr int square_at(int k) {
        if (k >= NN) return -1;
        if (PERM && NB == 5 && P == 4) {  // fours (the split-precision kernel): the same tiles as the pair order below, and inside a four
                                          // the places alternate with the checkerboard colour of the square, as on 6x6: the 4-row runs
                                          // of a lane group then never share a piece rotation (LDS bank conflicts 19.5 % -> see profiles)
            constexpr int T[25] = {2, 1, 4, 3, 10, 5, 20, 15, 22, 21, 24, 23, 14, 9, 6, 19, 8, 7, 12, 11, 16, 13, 18, 17, 0};
            return T[k];
        }
        if (PERM && NB == 5) {  // pairs along the top, left, bottom and right edges (3 taps each), one edge square left over with
                                // the interior, and a corner alone in the half-empty last tile (5 taps): 26 of 117 pairs skipped
            constexpr int T[25] = {1, 2, 3, 4, 5, 10, 15, 20, 21, 22, 23, 24, 9, 14, 19, 6, 7, 8, 11, 12, 13, 16, 17, 18, 0};
            return T[k];
        }
        if (PERM && NB == 6 && P == 8) {  // pairs: bottom row, top row, left and right columns without corners (3 taps off the board
                                          // for both squares of a pair), then the interior: 30 of the 162 (tap, tile) pairs skipped
            constexpr int T[36] = {0, 1, 2, 3, 4, 5, 30, 31, 32, 33, 34, 35, 6, 12, 18, 24, 11, 17, 23, 29,
                                   7, 8, 9, 10, 13, 14, 15, 16, 19, 20, 21, 22, 25, 26, 27, 28};
            return T[k];
        }
        if (PERM && NB == 6) {  // fours: the edges without corners, the corners, interior.  Inside a four the places alternate
                                // with the checkerboard colour of the square ((x + y) & 1 == place & 1): the 16-B piece
                                // rotation of the LDS image follows bit 2 of the row = bit 0 of the place, and any tap moves
                                // both squares of a place pair to the same colour change, so the pair never shares a bank
            constexpr int T[36] = {2, 1, 4, 3, 12, 6, 24, 18, 31, 32, 33, 34, 11, 17, 23, 29, 0, 5, 35, 30,
                                   7, 8, 9, 10, 14, 13, 16, 15, 19, 20, 21, 22, 26, 25, 28, 27};
            return T[k];
        }
        return k;
    }
    // the inverse of square_at (tables: checked against it at compile time, row_map_is_permutation)
    __host__ __device__ static constexpr int place_of(int square) {
        if (PERM && NB == 5 && P == 4) {
            constexpr int I[25] = {24, 1, 0, 3, 2, 5, 14, 17, 16, 13, 4, 19, 18, 21, 12, 7, 20, 23, 22, 15, 6, 9, 8, 11, 10};
            return I[square];
        }
        if (PERM && NB == 5) {
            constexpr int I[25] = {24, 0, 1, 2, 3, 4, 15, 16, 17, 12, 5, 18, 19, 20, 13, 6, 21, 22, 23, 14, 7, 8, 9, 10, 11};
            return I[square];
        }
        if (PERM && NB == 6 && P == 8) {
            constexpr int I[36] = {0, 1, 2, 3, 4, 5, 12, 20, 21, 22, 23, 16, 13, 24, 25, 26, 27, 17,
                                   14, 28, 29, 30, 31, 18, 15, 32, 33, 34, 35, 19, 6, 7, 8, 9, 10, 11};
            return I[square];
        }
        if (PERM && NB == 6) {
            constexpr int I[36] = {16, 1, 0, 3, 2, 17, 5, 20, 21, 22, 23, 12, 4, 25, 24, 27, 26, 13,
                                   7, 28, 29, 30, 31, 14, 6, 33, 32, 35, 34, 15, 19, 8, 9, 10, 11, 18};
            return I[square];
        }
        return square;
    }
    __host__ __device__ static constexpr int row_of(int board, int square) {
        if (!PERM) return board * NN + square;
        const int k = place_of(square);
        return (k / PPT) * 16 + (k % PPT) * P + board;
    }
    // board and square of an image row; square = -1 for padding rows
    __host__ __device__ static constexpr void decode(int row, int& board, int& square) {
        if (!PERM) {
            board = row / NN;
            square = board < P ? row % NN : -1;
            return;
        }
        const int i = row % 16;
        board = i % P;
        square = square_at((row / 16) * PPT + i / P);
    }
    // bit rt of the mask: row tile rt has at least one row whose source square under `tap` is on the board
    __host__ __device__ static constexpr unsigned tap_tile_mask(int tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        unsigned m = 0;
        for (int row = 0; row < RT * 16; row++) {
            int board = 0, sq = -1;
            decode(row, board, sq);
            if (sq < 0) continue;
            const int y = sq / NB + dy, x = sq % NB + dx;
            if (y >= 0 && y < NB && x >= 0 && x < NB) m |= 1u << (row / 16);
        }
        return m;
    }
};

// compile-time checks of the orders above: every square exactly once, and the number of (tap, tile) pairs left to issue
template <int NB, int P, bool PERM>
constexpr bool row_map_is_permutation() {
    typedef RowMap<NB, P, PERM> RM;
    for (int sq = 0; sq < NB * NB; sq++) {
        int seen = 0;
        for (int k = 0; k < NB * NB; k++) seen += RM::square_at(k) == sq;
        if (seen != 1) return false;
    }
    for (int k = 0; k < NB * NB; k++)
        if (RM::place_of(RM::square_at(k)) != k) return false;
    for (int b = 0; b < P; b++)
        for (int sq = 0; sq < NB * NB; sq++) {
            int board = -1, back = -1;
            RM::decode(RM::row_of(b, sq), board, back);
            if (board != b || back != sq) return false;
        }
    return true;
}
template <int NB, int P, bool PERM>
constexpr int row_map_pairs_issued() {
    int n = 0;
    for (int tap = 0; tap < 9; tap++)
        for (unsigned m = RowMap<NB, P, PERM>::tap_tile_mask(tap); m; m >>= 1) n += m & 1;
    return n;
}
static_assert(row_map_is_permutation<5, 8, true>() && row_map_is_permutation<6, 4, true>() && row_map_is_permutation<3, 16, true>() &&
                  row_map_is_permutation<5, 8, false>() && row_map_is_permutation<4, 12, false>(),
              "RowMap: row_of / decode must be inverse bijections");
static_assert(row_map_pairs_issued<5, 8, true>() == 91 && row_map_pairs_issued<5, 8, false>() == 117, "5x5: 26 of 117 pairs skipped");
static_assert(row_map_pairs_issued<6, 4, true>() == 69 && row_map_pairs_issued<3, 16, true>() == 49, "6x6: 12 of 81, 3x3: 32 of 81");
static_assert(row_map_is_permutation<6, 8, true>() && row_map_pairs_issued<6, 8, true>() == 132, "6x6 with 8 boards: 30 of 162 pairs skipped");
static_assert(row_map_is_permutation<5, 4, true>() && row_map_pairs_issued<5, 4, true>() == 49, "5x5 with 4 boards (split precision): 14 of 63 pairs skipped");

// tap table entries of a lane under a row map (the general form of tap_bases_rc)
template <int NB, int P, bool PERM, int LAYOUT>
__device__ __forceinline__ void tap_bases_map(int tap, int lr, int q, int zrow, int (&abase)[RowMap<NB, P, PERM>::RT]) {
    typedef RowMap<NB, P, PERM> RM;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int rt = 0; rt < RM::RT; rt++) {
        const int r = rt * 16 + lr;
        int board = 0, sq = -1;
        RM::decode(r, board, sq);
        const int y = sq / NB + dy, x = sq % NB + dx;
        const bool ok = sq >= 0 && y >= 0 && y < NB && x >= 0 && x < NB;
        // an off-board source reads the zero row with the row phase its on-board source would have had (P = 4: the colour
        // of the source square, see square_at): the same LDS slot pattern as on the board
        const int zr = (P == 4 && ((dy + dx) & 1)) ? r ^ 4 : r;
        abase[rt] = LdsImg<LAYOUT>::read_base(ok, ok ? RM::row_of(board, y * NB + x) : zr, q, zrow);
    }
}

template <int N>
struct IntC {
    static constexpr int value = N;
};

// k_loop_256 with the tap loop unrolled and the all-zero (tap, row tile) pairs of the row map left out
// ABL (diagnostic builds only, TZ_ABLATIONS): 1 = the activation fragments are read once and reused (no ds_read stream),
// 2 = the weight fragments are fetched once and reused (no L2 stream), 4 = operands stream but no MFMA is issued.
// ABL bit 32 (diagnostic builds): fairness between the two waves of a SIMD - every k-step a wave posts its step number in LDS, reads
// its partner's (wave ^ 4) and lowers its own priority while it is ahead (fair_slots: 8 ints; fair_step0: the conv's first step)
template <int NB, int P, int RNX, int PLANE, typename ET, int ABL = 0, typename WL>
__device__ __forceinline__ void k_loop_256_skip(const unsigned char* lds, const int* tap_table, int lane,
                                                f32x4 (&acc)[RowMap<NB, P, true>::RT][RNX], WL wl, int* fair_slots = nullptr, int fair_wave = 0,
                                                int fair_step0 = 0) {
    typedef typename Elem<ET>::x8 ex8;
    typedef RowMap<NB, P, true> RM;
    constexpr int TAPS = 9, RT = RM::RT;
    ex8 bq[4][RNX];
#pragma unroll
    for (int j = 0; j < RNX; j++) {
        bq[0][j] = wl(0, 0, j);
        bq[1][j] = wl(0, 1, j);
    }
    int abase[RT];
    ex8 av[RT];
    {
        constexpr unsigned M0 = RM::tap_tile_mask(0);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((M0 >> rt) & 1) abase[rt] = tap_table[rt * 64 + lane];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((M0 >> rt) & 1) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
    }
    auto one_tap = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr unsigned NOW = RM::tap_tile_mask(tap);
        constexpr unsigned NEXT = tap + 1 < TAPS ? RM::tap_tile_mask(tap + 1 < TAPS ? tap + 1 : tap) : 0u;
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            if constexpr ((ABL & 96) != 0) {
                // bit 32: the decision is taken once per tap and held for its eight k-steps; bit 64: once per three taps
                if (kc == 0 && ((ABL & 32) || tap % 3 == 0)) {
                    const int step = fair_step0 + tap * 8 + kc;
                    fair_slots[fair_wave] = step;
                    const int other = __builtin_amdgcn_readfirstlane(fair_slots[fair_wave ^ 4]);
                    if (other < step) __builtin_amdgcn_s_setprio(0);
                    else __builtin_amdgcn_s_setprio(3);
                }
            }
            if constexpr (!(ABL & 2)) {
                if (kc + 2 < 8) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap, kc + 2, j);
                } else if (tap + 1 < TAPS) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap + 1, kc + 2 - 8, j);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kc == 7) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    if ((NEXT >> rt) & 1) abase[rt] = tap_table[((tap + 1) * RT + rt) * 64 + lane];
            }
            if (kc == 4) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    if ((NOW >> rt) & 1) {
                        abase[rt] += 4 * PLANE;
                        asm volatile("" : "+v"(abase[rt]));
                    }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if ((NOW >> rt) & 1) {
                    if constexpr (ABL & 4) {   // keep the operands live without issuing the MFMAs
#pragma unroll
                        for (int j = 0; j < RNX; j++) asm volatile("" ::"v"(bq[kc & ((ABL & 2) ? 1 : 3)][j]), "v"(av[rt]));
                    } else {
#pragma unroll
                        for (int j = 0; j < RNX; j++) acc[rt][j] = Elem<ET>::mfma(bq[kc & ((ABL & 2) ? 1 : 3)][j], av[rt], acc[rt][j]);
                    }
                }
                if constexpr (!(ABL & 1)) {
                    if (kc < 7) {
                        if ((NOW >> rt) & 1) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + (kc + 1 - (kc >= 4 ? 4 : 0)) * PLANE);
                    } else if ((NEXT >> rt) & 1) {
                        av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                    }
                } else if (kc == 7 && tap == 0) {   // one refill per conv so that every tile holds a fragment
                    if (((NEXT & ~NOW) >> rt) & 1) av[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                }
            }
            // the issue order of k_loop_256: the MFMAs of a row tile, then the ds_read that refills its fragment
            if constexpr ((ABL & 7) == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    if ((NOW >> rt) & 1) __builtin_amdgcn_sched_group_barrier(0x008, RNX, 0);
                    if (kc < 7 ? ((NOW >> rt) & 1) : ((NEXT >> rt) & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    };
    one_tap(IntC<0>{});
    one_tap(IntC<1>{});
    one_tap(IntC<2>{});
    one_tap(IntC<3>{});
    one_tap(IntC<4>{});
    one_tap(IntC<5>{});
    one_tap(IntC<6>{});
    one_tap(IntC<7>{});
    one_tap(IntC<8>{});
}

// ---------------------------------------------------------------------------------------------
// The same conv for workgroups whose rows do not fit the register file of k_loop_256_skip: 6x6 with 8 boards is 18 row tiles,
// 144 accumulator registers, and a whole k-step of activation fragments ahead (72 more) does not fit beside them.  Here the
// fragments go through a ring of D registers that runs D (tap-tile, k-chunk) pairs ahead of the MFMAs, restarted at every tap;
// the lane's fragment addresses come from the compact table  tap_c[tap][rt][slot] + lane_const  (all rows of a square's 8-row run
// move together under a tap, so one entry per (tap, tile, square slot) is enough: 1.3 KB instead of 41 KB per-lane entries, which
// is what lets 8 boards' image fit the 160 KB at all).  Same k order as the other loops: bit-identical results.
template <int NB, int P, int RNX, int PLANE, typename ET, int D, typename WL>
__device__ __forceinline__ void k_loop_256_ring(const unsigned char* lds, const int* tap_c, int slot, int lane_const,
                                                f32x4 (&acc)[RowMap<NB, P, true>::RT][RNX], WL wl) {
    typedef typename Elem<ET>::x8 ex8;
    typedef RowMap<NB, P, true> RM;
    constexpr int TAPS = 9, RT = RM::RT, PPT = RM::PPT;
    ex8 bq[4][RNX];
#pragma unroll
    for (int j = 0; j < RNX; j++) {
        bq[0][j] = wl(0, 0, j);
        bq[1][j] = wl(0, 1, j);
    }
    auto one_tap = [&](auto tap_c_) {
        constexpr int tap = decltype(tap_c_)::value;
        constexpr unsigned NOW = RM::tap_tile_mask(tap);
        constexpr int NA = __builtin_popcount(NOW), NP = 8 * NA;
        // the rt of the i-th active tile of this tap
        auto nth = [](int i) constexpr -> int {
            int seen = 0;
            for (int rt = 0; rt < RT; rt++)
                if ((NOW >> rt) & 1) {
                    if (seen == i) return rt;
                    seen++;
                }
            return 0;
        };
        int abase[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((NOW >> rt) & 1) abase[rt] = tap_c[(tap * RT + rt) * PPT + slot] + lane_const;
        ex8 rg[D];
#pragma unroll
        for (int p = 0; p < D && p < NP; p++) rg[p] = *reinterpret_cast<const ex8*>(lds + abase[nth(p % NA)] + (p / NA) * PLANE);
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            if (kc + 2 < 8) {
#pragma unroll
                for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap, kc + 2, j);
            } else if (tap + 1 < TAPS) {
#pragma unroll
                for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap + 1, kc + 2 - 8, j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int p = kc * NA + i;          // this pair; its fragment sits in ring slot p % D
                const int rt = nth(i);
#pragma unroll
                for (int j = 0; j < RNX; j++) acc[rt][j] = Elem<ET>::mfma(bq[kc & 3][j], rg[p % D], acc[rt][j]);
                const int pn = p + D;               // the pair D ahead refills the slot
                if (pn < NP) {
                    const int kn = pn / NA, rn = nth(pn % NA);
                    if (pn % NA == 0 && kn == 4) {  // the read stream enters plane 4: rebase so that the offsets stay 16-bit immediates
#pragma unroll
                        for (int r2 = 0; r2 < RT; r2++)
                            if ((NOW >> r2) & 1) {
                                abase[r2] += 4 * PLANE;
                                asm volatile("" : "+v"(abase[r2]));
                            }
                    }
                    rg[p % D] = *reinterpret_cast<const ex8*>(lds + abase[rn] + (kn - (kn >= 4 ? 4 : 0)) * PLANE);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, RNX, 0);
                if (pn < NP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    one_tap(IntC<0>{});
    one_tap(IntC<1>{});
    one_tap(IntC<2>{});
    one_tap(IntC<3>{});
    one_tap(IntC<4>{});
    one_tap(IntC<5>{});
    one_tap(IntC<6>{});
    one_tap(IntC<7>{});
    one_tap(IntC<8>{});
}

// ---------------------------------------------------------------------------------------------
// Split precision (TZ_PREC_F16X2).  Every operand is a pair of halves: hi = fp16(x) and lo = fp16((x - hi) * 2^11), a
// 22-bit significand; the image holds 16 planes (0..7 hi, 8..15 lo), the weights two buffers in the same fragment order.
// A product is three MFMAs with fp32 accumulation: wh*xh into the main accumulator, wl*xh and wh*xl into a correction
// accumulator that is scaled by 2^-11 once per layer (the lo*lo term is below fp32's own rounding); fp16 x fp16 products
// are exact in fp32, so the layer is an fp32 convolution of 22-bit operands.  Same k order, tap table, row maps and
// skipped (tap, tile) pairs as k_loop_256_skip; weight fragments one k-step ahead (a k-step is 6 MFMAs per row tile).
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;
// A tap-table entry as a fragment address.  32- and 16-bit entries are the address; 8-bit entries (6x6 with 4 boards in a split
// precision: the table has to fit beside 16 planes) are the image row, and the lane's 16-byte piece follows from the row
template <typename TapT>
__device__ __forceinline__ int tap_addr(TapT e, int lane) {
    if constexpr (sizeof(TapT) == 1) {
        const int row = e;
        return row * LDS_ROWB + lds_piece(row, lane >> 4);
    } else {
        return (int)e;
    }
}
template <int NB, int P, bool PERM, int RNX, int PLANE, typename TapT, typename WL>
__device__ __forceinline__ void k_loop_split(const unsigned char* lds, const TapT* tap_table, int lane,
                                             f32x4 (&accm)[RowMap<NB, P, PERM>::RT][RNX], f32x4 (&accc)[RowMap<NB, P, PERM>::RT][RNX], WL wl) {
    typedef Elem<_Float16> E;
    typedef f16x8 ex8;
    typedef RowMap<NB, P, PERM> RM;
    constexpr int TAPS = 9, RT = RM::RT, LO = 8 * PLANE;
    ex8 bq[2][RNX][2];
#pragma unroll
    for (int j = 0; j < RNX; j++) {
        bq[0][j][0] = wl(0, 0, j, 0);
        bq[0][j][1] = wl(0, 0, j, 1);
    }
    int abase[RT];
    ex8 ah[RT], al[RT];
    {
        constexpr unsigned M0 = RM::tap_tile_mask(0);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((M0 >> rt) & 1) abase[rt] = tap_addr(tap_table[rt * 64 + lane], lane);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((M0 >> rt) & 1) {
                ah[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                al[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + LO);
            }
    }
    auto one_tap = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr unsigned NOW = RM::tap_tile_mask(tap);
        constexpr unsigned NEXT = tap + 1 < TAPS ? RM::tap_tile_mask(tap + 1 < TAPS ? tap + 1 : tap) : 0u;
#pragma unroll
        for (int kc = 0; kc < 8; kc++) {
            if (kc + 1 < 8) {
#pragma unroll
                for (int j = 0; j < RNX; j++) {
                    bq[(kc + 1) & 1][j][0] = wl(tap, kc + 1, j, 0);
                    bq[(kc + 1) & 1][j][1] = wl(tap, kc + 1, j, 1);
                }
            } else if (tap + 1 < TAPS) {
#pragma unroll
                for (int j = 0; j < RNX; j++) {
                    bq[(kc + 1) & 1][j][0] = wl(tap + 1, 0, j, 0);
                    bq[(kc + 1) & 1][j][1] = wl(tap + 1, 0, j, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kc == 7) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    if ((NEXT >> rt) & 1) abase[rt] = tap_addr(tap_table[((tap + 1) * RT + rt) * 64 + lane], lane);
            }
            if (kc == 4) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    if ((NOW >> rt) & 1) {
                        abase[rt] += 4 * PLANE;
                        asm volatile("" : "+v"(abase[rt]));
                    }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if ((NOW >> rt) & 1) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) accm[rt][j] = E::mfma(bq[kc & 1][j][0], ah[rt], accm[rt][j]);
#pragma unroll
                    for (int j = 0; j < RNX; j++) accc[rt][j] = E::mfma(bq[kc & 1][j][1], ah[rt], accc[rt][j]);
#pragma unroll
                    for (int j = 0; j < RNX; j++) accc[rt][j] = E::mfma(bq[kc & 1][j][0], al[rt], accc[rt][j]);
                }
                if (kc < 7) {
                    if ((NOW >> rt) & 1) {
                        const int off = abase[rt] + (kc + 1 - (kc >= 4 ? 4 : 0)) * PLANE;
                        ah[rt] = *reinterpret_cast<const ex8*>(lds + off);
                        al[rt] = *reinterpret_cast<const ex8*>(lds + off + LO);
                    }
                } else if ((NEXT >> rt) & 1) {
                    ah[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt]);
                    al[rt] = *reinterpret_cast<const ex8*>(lds + abase[rt] + LO);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if ((NOW >> rt) & 1) __builtin_amdgcn_sched_group_barrier(0x008, 3 * RNX, 0);
                if (kc < 7 ? ((NOW >> rt) & 1) : ((NEXT >> rt) & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
    };
    one_tap(IntC<0>{});
    one_tap(IntC<1>{});
    one_tap(IntC<2>{});
    one_tap(IntC<3>{});
    one_tap(IntC<4>{});
    one_tap(IntC<5>{});
    one_tap(IntC<6>{});
    one_tap(IntC<7>{});
    one_tap(IntC<8>{});
}


// ---------------------------------------------------------------------------------------------
// FP16 products with FP8 corrections (TZ_PREC_F16C8).  The split form above spends two of its three MFMAs on the correction
// wl*xh + wh*xl, a quantity 2^-11 of the product that only has to be right to a few bits for the sum to be right to 15.  Here
// those two products run on OCP FP8 (E4M3) copies of the four operands through v_mfma_f32_16x16x128_f8f6f4 - 128 input
// channels per instruction at twice the fp16 rate (32 cycles for 4x the K of the 16-cycle 16x16x32, tools/mfma_f8_probe.hip)
// - so a tap of 256 channels costs 8 fp16 + 4 FP8 MFMAs = 256 cycles per (row tile, 16 outputs) instead of 24 x 16 = 384.
// Image: planes 0..7 hi (fp16, as TZ_PREC_F16), 8..11 FP8 of hi * C8_SX, 12..15 FP8 of (x - hi) * 2^11 * C8_SX, 16..19 FP8 of
// what that second byte still misses (* 16): only the residual connection reads it (the block input is carried to 19 bits;
// carried at 15 the logits lose a factor 2, tools/fp8_correction_study.py).  An FP8 plane is [row][64 channels] with the
// same 16-B piece rotation as an fp16 plane, so a lane's tap-table address serves all of them: its 32 operand bytes of
// K-half m are piece q of planes 2m and 2m + 1.  Steps of a tap: fp16 chunks 0..3, FP8 half 0, chunks 4..7, FP8 half 1.
constexpr float C8_SX = 4.0f;                       // activations times 4: E4M3 saturates at 112, keeps 4 bits down to 2^-8
constexpr float C8_LO = 2048.0f * C8_SX;            // scale of the lo byte
constexpr float C8_LO_INV = 1.0f / C8_LO;
constexpr float C8_RS = 16.0f, C8_RS_INV = 1.0f / 16.0f;   // the remainder byte, relative to the lo byte
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_f8(i32x8 a, i32x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);   // both scales 0: the unscaled instruction
}
__device__ __forceinline__ i32x8 lds_f8_frag(const unsigned char* lds, int addr, int plane_bytes) {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(lds + addr);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(lds + addr + plane_bytes);
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// four values -> four E4M3 bytes (clamped: the instruction turns overflow into NaN)
__device__ __forceinline__ int pack_e4m3(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f), w, true);
    return w;
}
__device__ __forceinline__ float e4m3_byte(int w, int k) {
    return k == 0 ? __builtin_amdgcn_cvt_f32_fp8(w, 0) : k == 1 ? __builtin_amdgcn_cvt_f32_fp8(w, 1) : k == 2 ? __builtin_amdgcn_cvt_f32_fp8(w, 2)
                                                                                                        : __builtin_amdgcn_cvt_f32_fp8(w, 3);
}
// an activation's stored parts: hi (fp16) and the FP8 words of 4 channels; want_r: also the remainder word.  v is post-ReLU and
// at most 65504.  The scaled conversions divide by the power of two in `scale` and do not saturate (overflow is NaN,
// tools/cvt_scale_probe.hip), so the hi copy is capped at 112 = 448 / C8_SX and the lo part at +-448 / C8_LO first.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void c8_parts(const f32x4& v, f16x4& hi, int& h8, int& l8, int& r8, bool want_r) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        hi[k] = (_Float16)v[k];
        t[k] = __builtin_amdgcn_fmed3f(v[k] - (float)hi[k], -448.0f / C8_LO, 448.0f / C8_LO);
    }
    const f16x2 cap = {(_Float16)(448.0f / C8_SX), (_Float16)(448.0f / C8_SX)};
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_elementwise_min(f16x2{hi[0], hi[1]}, cap), 1.0f / C8_SX, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_elementwise_min(f16x2{hi[2], hi[3]}, cap), 1.0f / C8_SX, true);
    h8 = __builtin_bit_cast(int, w);
    s16x2 x = {0, 0};
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(x, t[0], t[1], C8_LO_INV, false);
    x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(x, t[2], t[3], C8_LO_INV, true);
    l8 = __builtin_bit_cast(int, x);
    r8 = 0;
    if (want_r) {   // what the lo byte misses, 16 times finer: |.| <= 16 * half an E4M3 step <= 256, no cap needed
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = __builtin_fmaf(e4m3_byte(l8, k), -C8_LO_INV, t[k]);
        s16x2 y = {0, 0};
        y = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(y, r[0], r[1], C8_LO_INV * C8_RS_INV, false);
        y = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(y, r[2], r[3], C8_LO_INV * C8_RS_INV, true);
        r8 = __builtin_bit_cast(int, y);
    }
}
__device__ __forceinline__ float c8_value(_Float16 hi, int l8, int r8, int k) {
    return (float)hi + (e4m3_byte(l8, k) + e4m3_byte(r8, k) * C8_RS_INV) * C8_LO_INV;
}

// One tap is a stream of 12 x NA items (NA = row tiles the tap does not skip): fp16 chunk kc of tile rt (one ds_read_b128, RNX
// fp16 MFMAs) or FP8 term of half m of tile rt (two ds_read_b128, RNX FP8 MFMAs; term 0 = wl8 * hi bytes, term 1 = wh8 * lo
// bytes).  The operands run D items ahead of the MFMAs through a ring of D 32-byte slots, across tap boundaries (the next
// tap's fragment addresses are read from the tap table two chunks before it starts); consecutive MFMAs never share an
// accumulator (all tiles of one chunk or term, then the next).
// FAIR (A/B): once per tap a wave posts its progress in LDS and lowers its priority while it is ahead of the SIMD's other wave
template <int NB, int P, bool PERM, int RNX, int PLANE, int FAIR = 0, typename TapT, typename WL, typename WL8>
__device__ __forceinline__ void k_loop_c8(const unsigned char* lds, const TapT* tap_table, int lane,
                                          f32x4 (&accm)[RowMap<NB, P, PERM>::RT][RNX], f32x4 (&accc)[RowMap<NB, P, PERM>::RT][RNX], WL wl, WL8 wl8,
                                          int* fair_slots = nullptr, int fair_wave = 0, int fair_step0 = 0) {
    typedef Elem<_Float16> E;
    typedef f16x8 ex8;
    typedef RowMap<NB, P, PERM> RM;
    constexpr int TAPS = 9, RT = RM::RT, D = 4;
    // plane offsets up to 7 planes: ds_read immediates on 5x5 (7 x 7680 B); the 6x6 four-board image (9728 B planes) pays an add on the last plane
    ex8 bq[4][RNX];
#pragma unroll
    for (int j = 0; j < RNX; j++) {
        bq[0][j] = wl(0, 0, j);
        bq[1][j] = wl(0, 1, j);
    }
    i32x8 b8[RNX][2];      // FP8 weight fragments of the coming half: [j][term: lo, hi]
    i32x8 rg[D];
    int abase[RT], abase8[RT], abn[RT];   // fragment addresses of the tap: fp16 planes, FP8 planes; of the next tap
    // item u of a tap with mask M: segment, tile, byte offset of its operand from the tile's fragment address
    // (fp16 planes from abase, FP8 planes from abase + 8 planes: both offsets stay below 7 planes)
    auto load_item = [&](auto mask_c, auto u_c, int (&ab)[RT], int (&ab8)[RT], i32x8& dst) {
        constexpr unsigned M = decltype(mask_c)::value;
        constexpr int u = decltype(u_c)::value;
        constexpr int NA = __builtin_popcount(M);
        constexpr int seg = u < 4 * NA ? 0 : u < 6 * NA ? 1 : u < 10 * NA ? 2 : 3;
        constexpr int v = u - (seg == 0 ? 0 : seg == 1 ? 4 * NA : seg == 2 ? 6 * NA : 10 * NA);
        constexpr int ti = v % NA, grp = v / NA;
        int rt = 0;
        {
            int seen = 0;
#pragma unroll
            for (int r = 0; r < RT; r++)
                if ((M >> r) & 1) {
                    if (seen == ti) rt = r;
                    seen++;
                }
        }
        if constexpr (seg == 0 || seg == 2) {
            const i32x4 x = *reinterpret_cast<const i32x4*>(lds + ab[rt] + ((seg == 2 ? 4 : 0) + grp) * PLANE);
            dst[0] = x[0];
            dst[1] = x[1];
            dst[2] = x[2];
            dst[3] = x[3];
        } else {
            constexpr int m = seg == 1 ? 0 : 1;
            dst = lds_f8_frag(lds, ab8[rt] + (2 * m + 4 * grp) * PLANE, PLANE);
        }
    };
    {
        constexpr unsigned M0 = RM::tap_tile_mask(0);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
            if ((M0 >> rt) & 1) {
                abase[rt] = tap_addr(tap_table[rt * 64 + lane], lane);
                abase8[rt] = abase[rt] + 8 * PLANE;
            }
        load_item(IntC<(int)M0>{}, IntC<0>{}, abase, abase8, rg[0]);
        load_item(IntC<(int)M0>{}, IntC<1>{}, abase, abase8, rg[1]);
        load_item(IntC<(int)M0>{}, IntC<2>{}, abase, abase8, rg[2]);
        load_item(IntC<(int)M0>{}, IntC<3>{}, abase, abase8, rg[3]);
    }
    auto one_tap = [&](auto tap_c) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr unsigned NOW = RM::tap_tile_mask(tap);
        constexpr unsigned NEXT = tap + 1 < TAPS ? RM::tap_tile_mask(tap + 1 < TAPS ? tap + 1 : tap) : 0u;
        constexpr int NA = __builtin_popcount(NOW), NI = 12 * NA;
        static_assert(NA >= 1 && 12 * __builtin_popcount(NEXT ? NEXT : 1u) >= D, "ring depth");
        if constexpr (FAIR != 0) {
            const int step = fair_step0 + tap;
            fair_slots[fair_wave] = step;
            const int other = __builtin_amdgcn_readfirstlane(fair_slots[fair_wave ^ 4]);
            if (other < step) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(3);
        }
        auto item = [&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            constexpr int seg = u < 4 * NA ? 0 : u < 6 * NA ? 1 : u < 10 * NA ? 2 : 3;
            constexpr int v = u - (seg == 0 ? 0 : seg == 1 ? 4 * NA : seg == 2 ? 6 * NA : 10 * NA);
            constexpr int ti = v % NA, grp = v / NA;
            int rt = 0;
            {
                int seen = 0;
#pragma unroll
                for (int r = 0; r < RT; r++)
                    if ((NOW >> r) & 1) {
                        if (seen == ti) rt = r;
                        seen++;
                    }
            }
            if constexpr ((seg == 0 || seg == 2) && ti == 0) {   // a new fp16 chunk starts: weights two chunks ahead
                constexpr int kc = (seg == 2 ? 4 : 0) + grp;
                if (kc + 2 < 8) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap, kc + 2, j);
                } else if (tap + 1 < TAPS) {
#pragma unroll
                    for (int j = 0; j < RNX; j++) bq[(kc + 2) & 3][j] = wl(tap + 1, kc + 2 - 8, j);
                }
                if (kc == 1 || kc == 5) {   // the FP8 fragments of the half that follows chunk 3 / 7
#pragma unroll
                    for (int j = 0; j < RNX; j++) {
                        b8[j][0] = wl8(tap, kc >> 2, j, 0);
                        b8[j][1] = wl8(tap, kc >> 2, j, 1);
                    }
                }
                if (kc == 6) {   // the next tap's fragment addresses
#pragma unroll
                    for (int r = 0; r < RT; r++)
                        if ((NEXT >> r) & 1) abn[r] = tap_addr(tap_table[((tap + 1) * RT + r) * 64 + lane], lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (seg == 0 || seg == 2) {
                constexpr int kc = (seg == 2 ? 4 : 0) + grp;
                const i32x4 x = i32x4{rg[u % D][0], rg[u % D][1], rg[u % D][2], rg[u % D][3]};
                const ex8 av = __builtin_bit_cast(ex8, x);
#pragma unroll
                for (int j = 0; j < RNX; j++) accm[rt][j] = E::mfma(bq[kc & 3][j], av, accm[rt][j]);
            } else {
#pragma unroll
                for (int j = 0; j < RNX; j++) accc[rt][j] = mfma_f8(b8[j][grp], rg[u % D], accc[rt][j]);
            }
            // refill the slot with the item D ahead: of this tap, or of the next
            if constexpr (u + D < NI) {
                load_item(IntC<(int)NOW>{}, IntC<u + D>{}, abase, abase8, rg[u % D]);
                constexpr int un = u + D;
                constexpr bool f8n = (un >= 4 * NA && un < 6 * NA) || un >= 10 * NA;
                __builtin_amdgcn_sched_group_barrier(0x008, RNX, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, f8n ? 2 : 1, 0);
            } else if constexpr (NEXT != 0u) {
                load_item(IntC<(int)NEXT>{}, IntC<u + D - NI>{}, abn, abn, rg[u % D]);   // an fp16 item: the FP8 base is not read
                __builtin_amdgcn_sched_group_barrier(0x008, RNX, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        // the items in order (at most 12 x 7: the index sequence is spelled out through a recursive lambda)
        auto run = [&](auto self, auto u_c) -> void {
            constexpr int u = decltype(u_c)::value;
            if constexpr (u < NI) {
                item(u_c);
                self(self, IntC<u + 1>{});
            }
        };
        run(run, IntC<0>{});
        if constexpr (NEXT != 0u) {
#pragma unroll
            for (int r = 0; r < RT; r++)
                if ((NEXT >> r) & 1) {
                    abase[r] = abn[r];
                    abase8[r] = abn[r] + 8 * PLANE;
                    asm volatile("" : "+v"(abase8[r]));   // keep it a register: the plane offsets then fold into the ds_read immediates
                }
        }
    };
    one_tap(IntC<0>{});
    one_tap(IntC<1>{});
    one_tap(IntC<2>{});
    one_tap(IntC<3>{});
    one_tap(IntC<4>{});
    one_tap(IntC<5>{});
    one_tap(IntC<6>{});
    one_tap(IntC<7>{});
    one_tap(IntC<8>{});
}

// hi / lo halves of an fp32 value (both saturating)
__device__ __forceinline__ void split_halves(float v, _Float16& hi, _Float16& lo) {
    hi = Elem<_Float16>::cvt(v);
    lo = Elem<_Float16>::cvt((v - (float)hi) * SPLIT_SCALE);
}

// SP = 1: split precision (k_loop_split): 16 image planes (hi 0..7, lo 8..15), two accumulator sets, hi / lo weight buffers.
// SP = 2: fp16 products with FP8 corrections (k_loop_c8): 20 image planes (hi 0..7, FP8 hi 8..11, FP8 lo 12..15, FP8 remainder
//         16..19), 16-bit tap table entries; the first conv runs the split form on planes 8.. as fp16 lo halves of its input.
// ABL: ablation bits of k_loop_256_skip for the tower (diagnostic builds); bit 8 = stamp s_memtime / s_memrealtime around the tower
// into a.dbg (the in-kernel clock: MI355X_MICROARCH.md, DVFS give-back item 6) - no output depends on the stamps.
// TT = 1: compact tap table + ring loop (k_loop_256_ring): the form for 18 row tiles (6x6, 8 boards); needs PERM and 16 % P == 0.
// NW = 4: four waves of 64 output channels, one per SIMD (A/B form, TZ_NET_W4=1): every activation fragment read from LDS feeds four
// MFMAs instead of two; a wave then owns 208 accumulator registers on 5x5 and nothing fills its gaps.
// SPLIT = 2 | 4: that many workgroups (CUs) share a board group, each computing 256 / SPLIT output channels of every conv (4 waves of
// 16 / (4 SPLIT) column tiles) from the whole image, which every member holds; after a layer the members hand each other their planes
// (a member's channels are 8 / SPLIT whole planes of the image) through a buffer in global memory.  For the Agent surface at the
// reference's batch of 128, where one workgroup per board streams all 47 MB of weights through one CU's L1 (0.45 ms): with four
// CUs each streams a quarter.  The hand-over is written access by access — stores with sc0 sc1 (written through the L2), s_waitcnt
// vmcnt(0), a system-scope atomic without a fence, loads with sc0 sc1 (past the L1 and any stale L2 line): 1.9-3.4 us per layer
// whether the members share an XCD or not (tools/cu_exchange_probe.hip: no stale read in either placement) — because agent-scope
// release / acquire fences write back and invalidate the whole L2 on gfx950 (23-68 us per layer).  Members are blocks b, b + 8,
// b + 16, b + 24: one XCD under the dispatcher's round-robin, where the exchange is fastest (the L2 serves it), but nothing depends
// on that.  A member that waits seconds for its partners poisons its outputs with NaN instead of hanging.  The accumulation order of
// an output does not depend on SPLIT (same k-loop, other RN): same bits as every other form.
template <int NB, int P, int RNP, typename ET, bool PERM = false, int SP = 0, int ABL = 0, int TT = 0, int NW = 8, int SPLIT = 1>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void net_mfma_kernel(NetArgs a) {
    typedef typename Elem<ET>::x8 ex8;
    typedef typename Elem<ET>::x4 ex4;
    static_assert(!SP || sizeof(ET) == 2, "split precision runs on fp16 halves");
    constexpr int RN = 16 / (NW * SPLIT), TAPS = 9, LAYOUT = 1, NT = NW * 64;
    static_assert((NW == 8 || NW == 4) && (NW == 8 || SP != 2), "four waves: not the FP8-correction form");
    static_assert(SPLIT == 1 || ((SPLIT == 2 || SPLIT == 4) && NW == 4 && SP != 2 && TT == 0 && ABL == 0 && !PERM), "several CUs per board group: the four-wave fp16 / bf16 / hi-lo forms, board-major rows");
    typedef RowMap<NB, P, PERM> RM;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = RM::RT, LROWS = RT * 16 + 8, ZROW = RT * 16;
    constexpr int PLANE = LROWS * LDS_ROWB;
    // 6x6 with 4 boards in a split precision: 16 planes and a table of 8-bit rows is what fits the 160 KB; the FP8-correction form then
    // has no remainder planes and carries a block input as hi + one FP8 byte (15 bits: twice the logit error, tools/fp8_correction_study.py)
    constexpr bool ROWTAB = SP != 0 && NB == 6 && P == 4;
    constexpr bool R8 = SP == 2 && NB != 6;   // every 6x6 form alike, so that a position's outputs do not depend on the batch it came in
    constexpr int NPL = SP == 2 ? (R8 ? 20 : 16) : SP ? 16 : 8, LO = 8 * PLANE;   // image planes; byte offset of the lo half of a plane
    typedef typename std::conditional<ROWTAB, uint8_t, typename std::conditional<SP == 2, uint16_t, int>::type>::type tap_t;
    static_assert(SP != 2 || PLANE < 65536, "16-bit tap table");
    static_assert(!ROWTAB || LROWS <= 256, "8-bit tap table");
    constexpr int LAYER_FRAGS = TAPS * 8 * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* hscratch = reinterpret_cast<float*>(lds + NPL * PLANE);  // [2][RT*16] head pre-activations
    tap_t* tap_table = reinterpret_cast<tap_t*>(lds + NPL * PLANE + 2 * RT * 16 * sizeof(float));  // [TAPS][RT][64 lanes], TT = 1: [TAPS][RT][PPT]
    static_assert(!TT || (PERM && !SP && 16 % P == 0 && P >= 8), "compact tap table: square-major rows with whole 8-row runs per square");
    const int count = a.count_dev ? *a.count_dev : a.count_host;
    // SPLIT: members of a group are 8 blocks apart (one XCD); the grid is a multiple of 8 SPLIT blocks
    const int group = SPLIT == 1 ? (int)blockIdx.x : ((int)blockIdx.x / (8 * SPLIT)) * 8 + (int)blockIdx.x % 8;
    const int member = SPLIT == 1 ? 0 : ((int)blockIdx.x % (8 * SPLIT)) / 8;
    const int pos0 = group * P;
    if (pos0 >= count) return;   // the whole group leaves together
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, lr = lane & 15;
    const int valid_boards = min(P, count - pos0);
    const size_t m0 = (size_t)pos0 * NN;
    const int ct0 = (member * NW + wave) * RN;
    const int lane16 = lane * 16;
    int& xch_failed = *reinterpret_cast<int*>(lds + NPL * PLANE + 2 * RT * 16 * sizeof(float) + (size_t)TAPS * RT * 64 * sizeof(tap_t));   // SPLIT: 16 bytes behind the tap table
    if (SPLIT > 1 && tid == 0) xch_failed = 0;

    // ---- the workgroup's packed states, copied into LDS in one round trip (plane 7 of the image is free until the first
    // conv's epilogue): game_repr walks every square of a state with data-dependent branches, which from global memory
    // is a chain of dependent loads per thread
    static_assert(sizeof(tz_state) % 4 == 0 && P * sizeof(tz_state) <= (size_t)RT * 16 * LDS_ROWB, "state staging fits a plane");
    {
        constexpr int DW = sizeof(tz_state) / 4;
        uint32_t* stage = reinterpret_cast<uint32_t*>(lds + 7 * PLANE);
        for (int i = tid; i < valid_boards * DW; i += NT) {
            const int b = i / DW, d = i - b * DW, pos = pos0 + b;
            stage[i] = reinterpret_cast<const uint32_t*>(a.states + (a.game_index ? a.game_index[pos] : pos))[d];
        }
        __syncthreads();
    }
    const tz_state* staged = reinterpret_cast<const tz_state*>(lds + 7 * PLANE);
    // ---- game_repr into planes 0..kc_in-1; zero rows of every plane
    for (int row = tid; row < LROWS; row += NT) {
        int board = 0, px = -1;
        if (row < RT * 16) RM::decode(row, board, px);
        const bool ok = px >= 0 && board < valid_boards;
        const tz_state* s = nullptr;
        int fd = 0;
        if (ok) {
            s = staged + board;
            fd = state_flat_diff<NB>(s);
        }
        float ssq = 0.0f;   // sum of squares of this square's planes, in plane order (for the RND input below)
        for (int c8 = 0; c8 < a.kc_in * 4; c8++) {
            ex8 v, vl;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = c8 * 8 + k;
                const float pv = (ok && c < a.cin_real) ? plane_value<NB>(s, px, c, fd) : 0.0f;
                ssq += pv * pv;
                v[k] = (ET)pv;
                if constexpr (SP) vl[k] = (ET)((pv - (float)v[k]) * SPLIT_SCALE);
            }
            *reinterpret_cast<ex8*>(lds + LdsImg<LAYOUT>::store_addr(row, c8, PLANE)) = v;
            if constexpr (SP) *reinterpret_cast<ex8*>(lds + LO + LdsImg<LAYOUT>::store_addr(row, c8, PLANE)) = vl;
        }
        if (row < RT * 16) hscratch[row] = ssq;
    }
    for (int i = tid; i < NPL * 8 * 4; i += NT) {  // planes x 8 zero rows x 4 pieces of 16 B
        const int plane = i >> 5, zr = (i >> 2) & 7, pc = i & 3;
        *reinterpret_cast<uint4*>(lds + plane * PLANE + (ZROW + zr) * LDS_ROWB + pc * 16) = make_uint4(0, 0, 0, 0);
    }
    // TT = 1: a lane's fragment address = entry of its (tap, tile, square slot) + its own constant: board row and piece rotation
    const int tslot = TT ? lr / P : 0;
    const int lane_const = TT ? (lr % P) * LDS_ROWB + lds_piece(lr % P, q) : 0;
    if constexpr (TT) {
        for (int i = tid; i < TAPS * RT * RM::PPT; i += NT) {
            const int tap = i / (RT * RM::PPT), rt = (i / RM::PPT) % RT, sl = i % RM::PPT;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int sq = RM::square_at(rt * RM::PPT + sl);
            const int y = sq / NB + dy, x = sq % NB + dx;
            const bool ok = sq >= 0 && y >= 0 && y < NB && x >= 0 && x < NB;
            tap_table[i] = (ok ? RM::row_of(0, y * NB + x) : ZROW) * LDS_ROWB;   // off the board: the zero rows, same board phase
        }
    } else {   // the per-tap fragment base addresses of a lane, once for all layers (read after the next barrier): wave w computes
        // taps w and w + 8
        for (int tap = wave; tap < TAPS; tap += NW) {
            int tb[RT];
            if constexpr (PERM) tap_bases_map<NB, P, true, LAYOUT>(tap, lr, q, ZROW, tb);
            else tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap, lr, q, ROWS, ZROW, tb);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) tap_table[(tap * RT + rt) * 64 + lane] = (tap_t)(ROWTAB ? tb[rt] / LDS_ROWB : tb[rt]);
        }
    }
    auto ta = [&](int tap, int rt) -> int {   // the lane's fragment base address of row tile rt under `tap`
        if constexpr (TT) return tap_table[(tap * RT + rt) * RM::PPT + tslot] + lane_const;
        else return tap_addr(tap_table[(tap * RT + rt) * 64 + lane], lane);
    };
    int obase[RN];
#pragma unroll
    for (int j = 0; j < RN; j++) obase[j] = ((ct0 + j) >> 1) * PLANE + lr * LDS_ROWB + lds_piece(lr, ((ct0 + j) & 1) * 2 + (q >> 1)) + (q & 1) * 8;   // channels 16 (ct0 + j) + 4 q ..: plane of 32, piece of 8
    // SP = 2: where the lane's four channels (16 * (2 wave + j) + 4 q ..) sit in the FP8 hi plane: plane 8 + wave / 2 of 64 channels,
    // piece 2 (wave & 1) + j, byte 4 q; the lo and remainder planes are 4 and 8 planes further
    int obase8[RN];
#pragma unroll
    for (int j = 0; j < RN; j++) obase8[j] = (8 + (wave >> 1)) * PLANE + lr * LDS_ROWB + lds_piece(lr, (wave & 1) * 2 + j) + q * 4;
    // stores one output tile's values (post-ReLU) in the parts the image of this SP holds
    auto store_c8 = [&](const f32x4& v, int j, int rt, bool block_output) {
        f16x4 hi;
        int h8, l8, r8;
        c8_parts(v, hi, h8, l8, r8, R8 && block_output);
        *reinterpret_cast<f16x4*>(lds + obase[j] + rt * 16 * LDS_ROWB) = hi;
        *reinterpret_cast<int*>(lds + obase8[j] + rt * 16 * LDS_ROWB) = h8;
        *reinterpret_cast<int*>(lds + obase8[j] + 4 * PLANE + rt * 16 * LDS_ROWB) = l8;
        if (R8 && block_output) *reinterpret_cast<int*>(lds + obase8[j] + 8 * PLANE + rt * 16 * LDS_ROWB) = r8;
    };

    // SPLIT: after a layer's epilogue every member publishes its planes and takes the others' (see the template's comment)
    int xround = 0;
    auto exchange = [&]() {
        if constexpr (SPLIT > 1) {
            constexpr int MYPL = 8 / SPLIT, PL16 = PLANE / 16, HALVES = SP ? 2 : 1;   // split precision: the hi planes 0..7 and the lo planes 8..15
            __syncthreads();   // the member's own planes are complete in LDS
            unsigned char* gbase = a.xch + (size_t)(group * 2 + (xround & 1)) * 8 * HALVES * PLANE;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, 8 * HALVES * PLANE, 0x00020000);
            for (int i = tid; i < HALVES * MYPL * PL16; i += NT) {
                const int pl = i / PL16, off = ((pl / MYPL) * 8 + member * MYPL + pl % MYPL) * PLANE + (i % PL16) * 16;
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(lds + off), rs, off, 0, 17);   // sc0 sc1: written through the L2
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have been acknowledged
            __syncthreads();
            if (tid == 0) {
                unsigned* counter = a.xch_count + 32 * group;
                const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(counter, 0, 128, 0x00020000);
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // executed past the L2: no fence, the waitcnt above orders it
                const unsigned want = (unsigned)SPLIT * (unsigned)(xround + 1);
                int spins = 0;
                while (!xch_failed && (unsigned)__builtin_amdgcn_raw_buffer_load_b32(crs, 0, 0, 17) < want) {   // sc0 sc1: past the L1
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 21)) xch_failed = 1;   // seconds: a partner is not coming; no more waiting, the outputs are poisoned below
                }
            }
            __syncthreads();
            for (int i = tid; i < HALVES * (8 - MYPL) * PL16; i += NT) {
                const int pl = i / PL16, off = ((pl / (8 - MYPL)) * 8 + (member * MYPL + MYPL + pl % (8 - MYPL)) % 8) * PLANE + (i % PL16) * 16;
                *reinterpret_cast<u32x4*>(lds + off) = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17);
            }
            xround++;
        }
    };

    f32x4 acc[RT][RN];
    f32x4 accc[SP ? RT : 1][RN];   // split precision: the correction accumulator (wl*xh + wh*xl), scaled by 2^-11 per layer
    // ---- first conv: cin_pad = 32*kc_in channels, 9*kc_in k-steps
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_in), 0, TAPS * a.kc_in * 16 * 1024, 0x00020000);
#pragma unroll
        for (int j = 0; j < RN; j++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias_in + (ct0 + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[rt][j] = b4;
        }
        // one 32-channel chunk of input planes (3x3 .. 5x5): all 18 weight fragments of the wave are requested before the
        // barrier, so that the nine taps do not each wait for their own round trip to L2
        const bool one_chunk = !SP && a.kc_in == 1;
        ex8 bw[SP ? 1 : TAPS][RN];
        if constexpr (!SP) {
            if (one_chunk) {
#pragma unroll
                for (int tap = 0; tap < TAPS; tap++)
#pragma unroll
                    for (int j = 0; j < RN; j++)
                        bw[tap][j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (tap * 16 + ct0 + j) * 1024, 0));
            }
        }
        __syncthreads();
        // ---- RND input (RndNetwork::normalize, net5.rs:127): x / sum(x^2) over the whole position, written once per
        // position by the threads that built its squares; the sum adds the squares' partial sums in square order, so it
        // does not depend on the row order or on the boards per workgroup
        if (a.rnd_in && member == 0) {
            for (int row = tid; row < RT * 16; row += NT) {
                int board = 0, px = -1;
                RM::decode(row, board, px);
                if (px < 0 || board >= valid_boards) continue;
                float ss = 0.0f;
                for (int sq = 0; sq < NN; sq++) ss += hscratch[RM::row_of(board, sq)];
                const tz_state* s = staged + board;
                const int fd = state_flat_diff<NB>(s);
                ET* out = reinterpret_cast<ET*>(a.rnd_in) + (size_t)(pos0 + board) * a.rnd_stride + px * a.cin_real;
                for (int c8 = 0; c8 < a.cin_real / 8; c8++) {
                    ex8 v;
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = (ET)(plane_value<NB>(s, px, c8 * 8 + k, fd) / ss);
                    *reinterpret_cast<ex8*>(out + c8 * 8) = v;
                }
            }
        }
        if constexpr (!SP) {
            if (one_chunk) {
#pragma unroll
                for (int tap = 0; tap < TAPS; tap++) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const ex8 av = *reinterpret_cast<const ex8*>(lds + ta(tap, rt));
#pragma unroll
                        for (int j = 0; j < RN; j++) acc[rt][j] = Elem<ET>::mfma(bw[tap][j], av, acc[rt][j]);
                    }
                }
            }
        }
        if constexpr (SP) {
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_in_lo), 0, TAPS * a.kc_in * 16 * 1024, 0x00020000);
#pragma unroll
            for (int j = 0; j < RN; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) accc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int tap = 0; tap < TAPS; tap++) {
                int abase[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) abase[rt] = ta(tap, rt);
                for (int kc = 0; kc < a.kc_in; kc++) {
                    ex8 bh[RN], bl[RN];
#pragma unroll
                    for (int j = 0; j < RN; j++) {
                        const int fo = ((tap * a.kc_in + kc) * 16 + ct0 + j) * 1024;
                        bh[j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, fo, 0));
                        bl[j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rsl, lane16, fo, 0));
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const ex8 avh = *reinterpret_cast<const ex8*>(lds + abase[rt] + kc * PLANE);
                        const ex8 avl = *reinterpret_cast<const ex8*>(lds + abase[rt] + kc * PLANE + LO);
#pragma unroll
                        for (int j = 0; j < RN; j++) {
                            acc[rt][j] = Elem<ET>::mfma(bh[j], avh, acc[rt][j]);
                            accc[rt][j] = Elem<ET>::mfma(bl[j], avh, accc[rt][j]);
                            accc[rt][j] = Elem<ET>::mfma(bh[j], avl, accc[rt][j]);
                        }
                    }
                }
            }
        } else {
            for (int tap = 0; tap < (one_chunk ? 0 : TAPS); tap++) {
                int abase[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) abase[rt] = ta(tap, rt);
                for (int kc = 0; kc < a.kc_in; kc++) {
                    ex8 b[RN];
#pragma unroll
                    for (int j = 0; j < RN; j++)
                        b[j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, ((tap * a.kc_in + kc) * 16 + ct0 + j) * 1024, 0));
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const ex8 av = *reinterpret_cast<const ex8*>(lds + abase[rt] + kc * PLANE);
#pragma unroll
                        for (int j = 0; j < RN; j++) acc[rt][j] = Elem<ET>::mfma(b[j], av, acc[rt][j]);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if constexpr (SP == 2) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        v[k] = __builtin_amdgcn_fmed3f(acc[rt][j][k] + accc[rt][j][k] * SPLIT_INV, 0.0f, 65504.0f);
                    }
                    store_c8(v, j, rt, true);
                    continue;
                }
                ex4 pk, pl;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (SP) {
                        const float v = acc[rt][j][k] + accc[rt][j][k] * SPLIT_INV;
                        _Float16 h, l;
                        split_halves(v > 0.f ? v : 0.f, h, l);
                        pk[k] = h;
                        pl[k] = l;
                    } else {
                        pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                    }
                }
                *reinterpret_cast<ex4*>(lds + obase[j] + rt * 16 * LDS_ROWB) = pk;
                if constexpr (SP) *reinterpret_cast<ex4*>(lds + LO + obase[j] + rt * 16 * LDS_ROWB) = pl;
            }
    }
    // ---- residual tower
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, a.nlayers * LAYER_FRAGS * 1024, 0x00020000);
    // SPLIT: the first weight fragments of the next layer are requested before the wait for the partners: their way from L2 overlaps the
    // hand-over, and a layer's k-loop does not begin with a round trip of its own (41 of them per forward)
    constexpr int KPD = SPLIT > 1 ? 12 : P <= 2 ? 6 : 2;
    ex8 wpre[SPLIT > 1 ? KPD : 1][RN];
    auto preload_weights = [&](int layer) {
        if constexpr (SPLIT > 1 && SP == 0) {
#pragma unroll
            for (int d = 0; d < KPD; d++)
#pragma unroll
                for (int j = 0; j < RN; j++)
                    wpre[d][j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, (layer * LAYER_FRAGS + d * 16 + (ct0 + j)) * 1024, 0));
        }
    };
    if (a.nlayers > 0) preload_weights(0);
    exchange();
    const __amdgpu_buffer_rsrc_t wrsrc_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SP == 1 ? a.w_lo : a.w), 0, a.nlayers * LAYER_FRAGS * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc8 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(SP == 2 ? a.w8 : reinterpret_cast<const unsigned char*>(a.w)), 0,
                                                                            a.nlayers * TAPS * 2 * 16 * 2 * 2048, 0x00020000);
    if constexpr (ABL & 8) {
        if (tid == 0 && a.dbg) {
            a.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
            a.dbg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
    for (int layer = 0; layer < a.nlayers; layer++) {
        if ((layer & 1) == 0) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + layer * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) acc[rt][j] = b4;
            }
        }
        if constexpr (SP) {
#pragma unroll
            for (int j = 0; j < RN; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) accc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if constexpr (ABL & 16) {   // phase stamps of the middle layer: [0] barrier passed, [1] k-loop done, [2] barrier passed, [3] epilogue done
            if (layer == a.nlayers / 2 && lane == 0 && wave == 0 && a.dbg) a.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        }
        if constexpr (SP == 2) {
            auto wl = [&](int tap, int kc, int j) -> ex8 {
                const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, 0));
            };
            auto wl8 = [&](int tap, int m, int j, int term) -> i32x8 {
                const int frag = ((layer * TAPS * 2 + tap * 2 + m) * 16 + (ct0 + j)) * 2 + term;   // 2 KB each
                const i32x4 lo = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc8, lane * 32, frag * 2048, 0));
                const i32x4 hi = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc8, lane * 32 + 16, frag * 2048, 0));
                return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            k_loop_c8<NB, P, PERM, RN, PLANE, ((ABL & 32) ? 1 : 0)>(lds, tap_table, lane, acc, accc, wl, wl8, reinterpret_cast<int*>(hscratch), wave, layer * 9);
        } else if constexpr (SP) {
            auto wl2 = [&](int tap, int kc, int j, int part) -> ex8 {
                const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(part ? wrsrc_lo : wrsrc, lane16, frag * 1024, 0));
            };
            if constexpr (SPLIT > 1) k_loop_split_deep<RT, RN, PLANE>(lds, tap_table, lane, acc, accc, wl2);
            else k_loop_split<NB, P, PERM, RN, PLANE>(lds, tap_table, lane, acc, accc, wl2);
        } else {
            auto wl = [&](int tap, int kc, int j) -> ex8 {
                const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, 0));
            };
            if constexpr (TT) k_loop_256_ring<NB, P, RN, PLANE, ET, 6>(lds, tap_table, tslot, lane_const, acc, wl);
            else if constexpr (PERM) k_loop_256_skip<NB, P, RN, PLANE, ET, (ABL & 103)>(lds, tap_table, lane, acc, wl, reinterpret_cast<int*>(hscratch), wave, layer * 72);
            else if constexpr (SPLIT > 1 && RT <= 4) k_loop_256_deep<RT, RN, PLANE, ET, KPD>(lds, tap_table, lane, acc, wl, &wpre[0][0]);
            else if constexpr (SPLIT > 1) k_loop_256<NB, RT, RN, ROWS, ZROW, PLANE, ET, KPD>(lds, tap_table, lane, acc, wl, &wpre[0][0]);   // 4 and 8 boards: k-steps long enough for one fragment ahead
            else k_loop_256<NB, RT, RN, ROWS, ZROW, PLANE, ET, KPD>(lds, tap_table, lane, acc, wl);
        }
        if constexpr (ABL & 16) {
            if (layer == a.nlayers / 2 && lane == 0 && wave == 0 && a.dbg) a.dbg[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        }
        __syncthreads();
        if constexpr (ABL & 16) {
            if (layer == a.nlayers / 2 && lane == 0 && wave == 0 && a.dbg) a.dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
        }
        const bool to_second = (layer & 1) == 0;
#pragma unroll
        for (int j = 0; j < RN; j++) {
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (to_second) b4 = *reinterpret_cast<const f32x4*>(a.bias + (layer + 1) * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                ex4* slot = reinterpret_cast<ex4*>(lds + obase[j] + rt * 16 * LDS_ROWB);
                ex4 pk;
                if constexpr (SP == 2) {
                    const float cs = a.c8_scales[layer];
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        v[k] = __builtin_amdgcn_fmed3f(acc[rt][j][k] + accc[rt][j][k] * cs, 0.0f, 65504.0f);
                    }
                    if (to_second) {   // the block input, carried to 19 bits, starts the second conv's accumulator
                        const f16x4 xh = *reinterpret_cast<const f16x4*>(slot);
                        const int xl8 = *reinterpret_cast<const int*>(lds + obase8[j] + 4 * PLANE + rt * 16 * LDS_ROWB);
                        const int xr8 = R8 ? *reinterpret_cast<const int*>(lds + obase8[j] + 8 * PLANE + rt * 16 * LDS_ROWB) : 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[rt][j][k] = c8_value(xh[k], xl8, xr8, k) + b4[k];
                    }
                    store_c8(v, j, rt, !to_second);
                } else if constexpr (SP) {
                    ex4* slot_lo = reinterpret_cast<ex4*>(lds + LO + obase[j] + rt * 16 * LDS_ROWB);
                    ex4 pl;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float v = acc[rt][j][k] + accc[rt][j][k] * SPLIT_INV;
                        _Float16 h, l;
                        split_halves(v > 0.f ? v : 0.f, h, l);
                        pk[k] = h;
                        pl[k] = l;
                    }
                    if (to_second) {
                        const ex4 xv = *slot, xl = *slot_lo;
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[rt][j][k] = ((float)xv[k] + (float)xl[k] * SPLIT_INV) + b4[k];
                    }
                    *slot = pk;
                    *slot_lo = pl;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) pk[k] = Elem<ET>::relu_cvt(acc[rt][j][k]);
                    if (to_second) {
                        const ex4 xv = *slot;
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[rt][j][k] = (float)xv[k] + b4[k];
                    }
                    *slot = pk;
                }
            }
        }
        if constexpr (ABL & 16) {
            if (layer == a.nlayers / 2 && lane == 0 && wave == 0 && a.dbg) a.dbg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
        }
        if (layer + 1 < a.nlayers) preload_weights(layer + 1);
        exchange();
    }
    __syncthreads();  // the image now holds the tower's output
    if constexpr (ABL & 8) {
        if (tid == 0 && a.dbg) {
            a.dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
            a.dbg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
    // ---- value / UBE heads: conv1x1(256->1)+bias, ReLU over the image rows, then Linear(nn->1) per board
    {
        const float* hw = a.heads;
        float wv[4], wu[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            wv[k] = hw[lane * 4 + k];
            wu[k] = hw[FILTERS + lane * 4 + k];
        }
        const float* lv = hw + 2 * FILTERS;
        const float* lu = lv + NN;
        const float bv = lu[NN], bu = lu[NN + 1], lbv = lu[NN + 2], lbu = lu[NN + 3];
        const int hplane = lane >> 3, hpiece = (lane & 7) >> 1, hhalf = lane & 1;  // channels 4*lane .. 4*lane+3
        constexpr int HROWS = PERM ? RT * 16 : ROWS;
        // four rows per step: the 64-lane reductions of a row are a chain of six dependent cross-lane moves, four rows
        // give the pipeline four independent chains (same reduction tree per row, so the same bits)
        constexpr int HU = 4;
        for (int row0 = wave; row0 < HROWS; row0 += NW * HU) {
            float dv[HU], du[HU];
#pragma unroll
            for (int u = 0; u < HU; u++) {
                const int row = min(row0 + NW * u, HROWS - 1);
                const int haddr = hplane * PLANE + row * LDS_ROWB + lds_piece(row, hpiece) + hhalf * 8;
                const ex4 xv = *reinterpret_cast<const ex4*>(lds + haddr);
                float xf[4];
#pragma unroll
                for (int k = 0; k < 4; k++) xf[k] = (float)xv[k];
                if constexpr (SP == 2) {   // channels 4 lane ..: FP8 plane lane / 16, piece (lane & 15) / 4, byte 4 (lane & 3)
                    const int h8addr = (12 + (lane >> 4)) * PLANE + row * LDS_ROWB + lds_piece(row, (lane & 15) >> 2) + (lane & 3) * 4;
                    const int xl8 = *reinterpret_cast<const int*>(lds + h8addr), xr8 = R8 ? *reinterpret_cast<const int*>(lds + h8addr + 4 * PLANE) : 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) xf[k] = c8_value(xv[k], xl8, xr8, k);
                } else if constexpr (SP) {
                    const ex4 xl = *reinterpret_cast<const ex4*>(lds + LO + haddr);
#pragma unroll
                    for (int k = 0; k < 4; k++) xf[k] += (float)xl[k] * SPLIT_INV;
                }
                dv[u] = 0.f;
                du[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    dv[u] += xf[k] * wv[k];
                    du[u] += xf[k] * wu[k];
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
                for (int u = 0; u < HU; u++) {
                    dv[u] += __shfl_xor(dv[u], d);
                    du[u] += __shfl_xor(du[u], d);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < HU; u++) {
                    const int row = row0 + NW * u;
                    if (row < HROWS) {
                        const float a = dv[u] + bv, b = du[u] + bu;
                        hscratch[row] = a > 0.f ? a : 0.f;
                        hscratch[RT * 16 + row] = b > 0.f ? b : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        for (int pos = wave; pos < P; pos += NW) {
            if (pos0 + pos >= count) break;
            float sv = 0.f, su = 0.f;
            for (int px = lane; px < NN; px += 64) {
                const int hr = RM::row_of(pos, px);
                sv += hscratch[hr] * lv[px];
                su += hscratch[RT * 16 + hr] * lu[px];
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sv += __shfl_xor(sv, d);
                su += __shfl_xor(su, d);
            }
            if (lane == 0 && member == 0) {
                const float poison = SPLIT > 1 && xch_failed ? __builtin_nanf("") : 0.0f;
                a.value[pos0 + pos] = tanhf(sv + lbv) + poison;
                a.ube[pos0 + pos] = su + lbu + poison;
            }
        }
    }
    // ---- policy conv: 16*RNPW output channels per wave, fp32 out
    constexpr int PM = SPLIT == 1 ? 1 : (8 * RNP / NW < SPLIT ? 8 * RNP / NW : SPLIT);   // members that take part in the policy conv
    if (member < PM) {
        constexpr int RNPW = RNP * 8 / (NW * PM);   // column tiles of a wave
        const int ctp = (member * NW + wave) * RNPW;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_pol), 0, TAPS * 8 * 8 * RNP * 1024, 0x00020000);
        f32x4 pacc[RT][RNPW];
#pragma unroll
        for (int j = 0; j < RNPW; j++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias_pol + (ctp + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) pacc[rt][j] = b4;
        }
        if constexpr (SP == 2) {
            const __amdgpu_buffer_rsrc_t rs8 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.w_pol8), 0, TAPS * 2 * 8 * RNP * 2 * 2048, 0x00020000);
            f32x4 paccc[RT][RNPW];
#pragma unroll
            for (int j = 0; j < RNPW; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) paccc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto wlp = [&](int tap, int kc, int j) -> ex8 {
                const int frag = (tap * 8 + kc) * (8 * RNP) + (ctp + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, frag * 1024, 0));
            };
            auto wlp8 = [&](int tap, int m, int j, int term) -> i32x8 {
                const int frag = ((tap * 2 + m) * (8 * RNP) + (ctp + j)) * 2 + term;
                const i32x4 lo = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs8, lane * 32, frag * 2048, 0));
                const i32x4 hi = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs8, lane * 32 + 16, frag * 2048, 0));
                return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            k_loop_c8<NB, P, PERM, RNPW, PLANE>(lds, tap_table, lane, pacc, paccc, wlp, wlp8);
            const float cs = a.c8_scales[a.nlayers];
#pragma unroll
            for (int j = 0; j < RNPW; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) pacc[rt][j][k] += paccc[rt][j][k] * cs;
        } else if constexpr (SP) {
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_pol_lo), 0, TAPS * 8 * 8 * RNP * 1024, 0x00020000);
            f32x4 paccc[RT][RNPW];
#pragma unroll
            for (int j = 0; j < RNPW; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) paccc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto wlp2 = [&](int tap, int kc, int j, int part) -> ex8 {
                const int frag = (tap * 8 + kc) * (8 * RNP) + (ctp + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(part ? rsl : rs, lane16, frag * 1024, 0));
            };
            if constexpr (SPLIT > 1) k_loop_split_deep<RT, RNPW, PLANE>(lds, tap_table, lane, pacc, paccc, wlp2);
            else k_loop_split<NB, P, PERM, RNPW, PLANE>(lds, tap_table, lane, pacc, paccc, wlp2);
#pragma unroll
            for (int j = 0; j < RNPW; j++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int k = 0; k < 4; k++) pacc[rt][j][k] += paccc[rt][j][k] * SPLIT_INV;
        } else {
            auto wlp = [&](int tap, int kc, int j) -> ex8 {
                const int frag = (tap * 8 + kc) * (8 * RNP) + (ctp + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, frag * 1024, 0));
            };
            if constexpr (TT) k_loop_256_ring<NB, P, RNPW, PLANE, ET, 6>(lds, tap_table, tslot, lane_const, pacc, wlp);
            else if constexpr (PERM) k_loop_256_skip<NB, P, RNPW, PLANE, ET>(lds, tap_table, lane, pacc, wlp);
            else if constexpr (SPLIT > 1 && RT <= 4) k_loop_256_deep<RT, RNPW, PLANE, ET>(lds, tap_table, lane, pacc, wlp, nullptr);
            else if constexpr (SPLIT > 1) k_loop_256<NB, RT, RNPW, ROWS, ZROW, PLANE, ET, 12>(lds, tap_table, lane, pacc, wlp);
            else k_loop_256<NB, RT, RNPW, ROWS, ZROW, PLANE, ET>(lds, tap_table, lane, pacc, wlp);
        }
#pragma unroll
        for (int j = 0; j < RNPW; j++) {
            const int cbase = (ctp + j) * 16 + q * 4;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                int board = 0, sq = -1;
                RM::decode(rt * 16 + lr, board, sq);
                if (sq >= 0 && board < valid_boards) {
                    if (SPLIT > 1 && xch_failed) pacc[rt][j][0] = __builtin_nanf("");
                    *reinterpret_cast<f32x4*>(a.policy_out + (m0 + board * NN + sq) * a.pol_stride + cbase) = pacc[rt][j];
                }
            }
        }
    }
}

#ifndef TZ_NN_SPLIT_TU   // the second translation unit (tz_nn_split.hip) holds the split-precision instantiations of the net kernel only
// ---------------------------------------------------------------------------------------------
// fp32 validation path: one thread per (row, output channel); weights [tap][cin][cout].
template <int NB>
__global__ void conv_f32_kernel(const float* in, const float* w, const float* bias, const float* residual, float* out,
                                const int32_t* count_dev, int count_host, int taps, int cin, int in_stride, int cout,
                                int out_stride, int relu) {
    constexpr int NN = NB * NB;
    const int count = count_dev ? *count_dev : count_host;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)count * NN;
    if (idx >= rows * cout) return;
    const size_t row = idx / cout;
    const int co = (int)(idx % cout);
    const int px = (int)(row % NN), y = px / NB, x = px % NB;
    float acc = 0.0f;
    for (int t = 0; t < taps; t++) {
        const int dy = taps == 9 ? t / 3 - 1 : 0, dx = taps == 9 ? t % 3 - 1 : 0;
        if (y + dy < 0 || y + dy >= NB || x + dx < 0 || x + dx >= NB) continue;
        const float* ip = in + (row + dy * NB + dx) * in_stride;
        const float* wp = w + (size_t)t * cin * cout + co;
        for (int c = 0; c < cin; c++) acc = fmaf(ip[c], wp[(size_t)c * cout], acc);
    }
    acc += bias[co];
    if (residual) acc += residual[row * out_stride + co];
    if (relu) acc = acc > 0.f ? acc : 0.f;
    out[row * out_stride + co] = acc;
}

// input planes NHWC fp32: planes[slot][px][cin]
template <int NB>
__global__ void encode_kernel(const tz_state* states, const int32_t* game_index, const int32_t* count_dev,
                              int count_host, int cin, float* planes) {
    constexpr int NN = NB * NB;
    const int count = count_dev ? *count_dev : count_host;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * NN) return;
    const int pos = idx / NN, px = idx % NN;
    const tz_state* s = states + (game_index ? game_index[pos] : pos);
    const int fd = state_flat_diff<NB>(s);
    for (int c = 0; c < cin; c++) planes[(size_t)idx * cin + c] = plane_value<NB>(s, px, c, fd);
}

// value + UBE heads: conv1x1(256->1)+bias, ReLU, Linear(nn->1) (+tanh for value). One wave per board.
template <typename T>
__global__ __launch_bounds__(64) void heads_kernel(const T* act, const float* hw, const int32_t* count_dev,
                                                   int count_host, int nn, float* value, float* ube) {
    const int count = count_dev ? *count_dev : count_host;
    const int pos = blockIdx.x, l = threadIdx.x;
    if (pos >= count) return;
    float wv[4], wu[4];
    for (int k = 0; k < 4; k++) {
        wv[k] = hw[l * 4 + k];
        wu[k] = hw[FILTERS + l * 4 + k];
    }
    const float* lv = hw + 2 * FILTERS;
    const float* lu = lv + nn;
    const float bv = lu[nn], bu = lu[nn + 1], lbv = lu[nn + 2], lbu = lu[nn + 3];
    float sv = 0.f, su = 0.f;
    for (int px = 0; px < nn; px++) {
        const T* p = act + ((size_t)pos * nn + px) * FILTERS + l * 4;
        float dv = 0.f, du = 0.f;
        for (int k = 0; k < 4; k++) {
            const float x = (float)p[k];
            dv += x * wv[k];
            du += x * wu[k];
        }
        for (int d = 32; d >= 1; d >>= 1) {
            dv += __shfl_xor(dv, d);
            du += __shfl_xor(du, d);
        }
        dv += bv;
        du += bu;
        sv += (dv > 0.f ? dv : 0.f) * lv[px];
        su += (du > 0.f ? du : 0.f) * lu[px];
    }
    if (l == 0) {
        value[pos] = tanhf(sv + lbv);
        ube[pos] = su + lbu;
    }
}

// RND input: x / sum(x^2) (net5.rs:127), written in NHWC plane order (weights are permuted to match)
template <typename T>
__global__ __launch_bounds__(64) void rnd_prep_kernel(const float* planes, const int32_t* count_dev, int count_host,
                                                      int in_size, int out_stride, T* out) {
    const int count = count_dev ? *count_dev : count_host;
    const int pos = blockIdx.x, l = threadIdx.x;
    if (pos >= count) return;
    const float* x = planes + (size_t)pos * in_size;
    float ss = 0.f;
    for (int i = l; i < in_size; i += 64) ss += x[i] * x[i];
    for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
    for (int i = l; i < out_stride; i += 64) out[(size_t)pos * out_stride + i] = (T)(i < in_size ? x[i] / ss : 0.f);
}

// the same straight from the packed state (game_repr fused): one wave per board
template <int NB, typename ET>
__global__ __launch_bounds__(64) void rnd_prep_state_kernel(const tz_state* states, const int32_t* game_index,
                                                            const int32_t* count_dev, int count_host, int cin,
                                                            int out_stride, ET* out) {
    constexpr int NN = NB * NB;
    const int count = count_dev ? *count_dev : count_host;
    const int pos = blockIdx.x, l = threadIdx.x;
    if (pos >= count) return;
    __shared__ tz_state sh;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(states + (game_index ? game_index[pos] : pos));
        uint32_t* dst = reinterpret_cast<uint32_t*>(&sh);
        for (int i = l; i < (int)(sizeof(tz_state) / 4); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const tz_state* s = &sh;
    const int fd = state_flat_diff<NB>(s);
    const int in_size = NN * cin;
    constexpr int PER = (NN * 40 + 63) / 64;  // cin <= 40
    float x[PER];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = l + k * 64;
        x[k] = i < in_size ? plane_value<NB>(s, i / cin, i % cin, fd) : 0.f;
        ss += x[k] * x[k];
    }
    for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = l + k * 64;
        if (i < out_stride) out[(size_t)pos * out_stride + i] = (ET)(i < in_size ? x[k] / ss : 0.f);
    }
}

// variance = clamp(max(exp(ube), local), 0, 4)  (net5.rs:271-278, net6_simhash.rs:311-318)
__global__ __launch_bounds__(64) void rnd_finish_kernel(const float* learn, const float* target, const float* ube,
                                                        const int32_t* count_dev, int count_host, int dim, int stride,
                                                        float rmin, float rmax, float* variance) {
    const int count = count_dev ? *count_dev : count_host;
    const int pos = blockIdx.x, l = threadIdx.x;
    if (pos >= count) return;
    float s = 0.f;
    for (int i = l; i < dim; i += 64) {
        const float d = learn[(size_t)pos * stride + i] - target[(size_t)pos * stride + i];
        s += d * d;
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (l == 0) {
        float norm = (s - rmin) / (rmax - rmin);
        norm = fminf(fmaxf(norm, 0.f), 1.f) * 4.0f;
        variance[pos] = fminf(fmaxf(fmaxf(expf(ube[pos]), norm), 0.f), 4.f);
    }
}

__global__ void plain_variance_kernel(const float* ube, const float* local, const int32_t* count_dev, int count_host,
                                      float* variance) {
    const int count = count_dev ? *count_dev : count_host;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= count) return;
    const float loc = local ? local[pos] : 0.f;
    variance[pos] = fminf(fmaxf(fmaxf(expf(ube[pos]), loc), 0.f), 4.f);
}

// SimHash straight from the packed game states, 8 boards per workgroup (HashNetwork::get_indices, net6_simhash.rs:202-236): the
// planes are built in LDS in the reference's flattening (k = plane * N*N + square, the black-to-move plane zeroed, :208-222), then
// thread (board, bit) walks k once: one broadcast LDS read and one coalesced 128-B row of the [in_size][32] matrix per step.  The
// first version (one wave per board) read the planes with a stride and re-fetched the 166 KB matrix per board: 249 us per 2048 6x6 positions
// (13 % of a simulation); this one also makes the separate plane-encoding launch unnecessary on the MFMA path.
template <int NB>
__global__ __launch_bounds__(256) void simhash_state_kernel(const tz_state* states, const int32_t* game_index, const float* matrix,
                                                            const uint32_t* bitset, const int32_t* count_dev, int count_host, float* local,
                                                            uint32_t* index_out) {
    constexpr int NN = NB * NB, CIN = 4 * NB + 12, IN = CIN * NN, BPW = 8, DW = sizeof(tz_state) / 4;
    __shared__ tz_state st[BPW];
    __shared__ float pl[BPW][IN];
    __shared__ int fd[BPW];
    const int count = count_dev ? *count_dev : count_host;
    const int pos0 = blockIdx.x * BPW, tid = threadIdx.x;
    if (pos0 >= count) return;
    const int nb = min(BPW, count - pos0);
    for (int i = tid; i < nb * DW; i += 256) {
        const int b = i / DW, d = i - b * DW, pos = pos0 + b;
        reinterpret_cast<uint32_t*>(st)[i] = reinterpret_cast<const uint32_t*>(states + (game_index ? game_index[pos] : pos))[d];
    }
    __syncthreads();
    if (tid < nb) fd[tid] = state_flat_diff<NB>(&st[tid]);
    __syncthreads();
    for (int i = tid; i < nb * IN; i += 256) {
        const int b = i / IN, k = i - b * IN, c = k / NN, px = k - c * NN;
        pl[b][k] = c == CIN - 2 ? 0.0f : plane_value<NB>(&st[b], px, c, fd[b]);
    }
    __syncthreads();
    const int bit = tid & 31, b = tid >> 5;
    float s = 0.f;
    if (b < nb) {
#pragma unroll 8
        for (int k = 0; k < IN; k++) s += pl[b][k] * matrix[(size_t)k * 32 + bit];
    }
    const unsigned long long m = __ballot(!(s < 0.0f));
    const uint32_t index = (tid & 32) ? (uint32_t)(m >> 32) : (uint32_t)m;
    if (bit == 0 && b < nb) {
        const bool seen = bitset && ((bitset[index >> 5] >> (index & 31)) & 1u);
        local[pos0 + b] = seen ? 0.0f : 4.0f;
        if (index_out) index_out[pos0 + b] = index;
    }
}

int launch_simhash_state(int n, const tz_state* states, const int32_t* gidx, const float* matrix, const uint32_t* bitset,
                         const int32_t* count_dev, int count_host, int max_positions, float* local, uint32_t* index_out, hipStream_t st) {
    const int blocks = (max_positions + 7) / 8;
    switch (n) {
        case 4: simhash_state_kernel<4><<<blocks, 256, 0, st>>>(states, gidx, matrix, bitset, count_dev, count_host, local, index_out); break;
        case 6: simhash_state_kernel<6><<<blocks, 256, 0, st>>>(states, gidx, matrix, bitset, count_dev, count_host, local, index_out); break;
        default: return tz_fail(TZ_EINVAL, "simhash: the SimHash nets are 4x4 and 6x6");
    }
    return TZ_OK;
}

// HashNetwork::update_counts (net6_simhash.rs:238-243): set the bit of every index
__global__ void bitset_set_kernel(uint32_t* bitset, const uint32_t* index, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicOr(&bitset[index[i] >> 5], 1u << (index[i] & 31));
}

// logits of the legal actions: out[b][j] = policy[b][px(a)][ch(a)]  (net5.rs:239-267)
__global__ void gather_kernel(const float* policy, int nn, int stride, const uint16_t* legal, const int32_t* cnt,
                              int amax, int batch, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * amax) return;
    const int b = idx / amax, j = idx % amax;
    float v = 0.f;
    if (j < cnt[b]) {
        const int a = legal[idx];
        v = policy[((size_t)b * nn + a % nn) * stride + a / nn];
    }
    out[idx] = v;
}
// the same for the Agent surface, plus value and variance into the same output buffer (one copy back)
__global__ void eval_pack_kernel(const float* policy, int nn, int stride, const uint16_t* legal, const int32_t* cnt, int amax, int batch,
                                 const float* value, const float* variance, float* out, float* value_out, float* variance_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < batch) {
        value_out[idx] = value[idx];
        variance_out[idx] = variance[idx];
    }
    if (idx >= batch * amax) return;
    const int b = idx / amax, j = idx % amax;
    float v = 0.f;
    if (j < cnt[b]) {
        const int a = legal[idx];
        v = policy[((size_t)b * nn + a % nn) * stride + a / nn];
    }
    out[idx] = v;
}
// policy tensor in the reference's NCHW flattening (net5.rs:238): out[b][ch*nn + px]
__global__ void policy_nchw_kernel(const float* policy, int nn, int stride, int channels, int batch, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * channels * nn) return;
    const int b = idx / (channels * nn), r = idx % (channels * nn);
    const int ch = r / nn, px = r % nn;
    out[idx] = policy[((size_t)b * nn + px) * stride + ch];
}
__global__ void planes_nchw_kernel(const float* planes, int nn, int cin, int batch, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * cin * nn) return;
    const int b = idx / (cin * nn), r = idx % (cin * nn);
    const int c = r / nn, px = r % nn;
    out[idx] = planes[((size_t)b * nn + px) * cin + c];
}

// ---------------------------------------------------------------------------------------------
uint16_t f2bf(float f) {  // round to nearest even; NaN stays NaN
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

uint16_t f2h(float f) {  // IEEE binary16, round to nearest even (host clang has _Float16)
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

// TZ_PREC_F16X2 runs the fp16 kernels everywhere (RND side networks included) except in the fused trunk + heads launch,
// where every operand is a hi / lo pair of halves (k_loop_split)
// TZ_PREC_F16C8 likewise; its fused launch takes the correction products through FP8 copies of the operands (k_loop_c8)
// TZ_PREC_F16C6 likewise, with E2M3 block-scaled copies (tz_nn_c6.hip)
inline bool prec_is_f16(int precision) { return precision == TZ_PREC_F16 || precision == TZ_PREC_F16X2 || precision == TZ_PREC_F16C8 || precision == TZ_PREC_F16C6; }
inline bool prec_is_split(int precision) { return precision == TZ_PREC_F16X2 || precision == TZ_PREC_F16C8 || precision == TZ_PREC_F16C6; }

struct Tensor {
    std::vector<uint32_t> dims;
    const float* data;
    size_t size;
};
typedef std::map<std::string, Tensor> TensorMap;

TensorMap view_of(const TensorStore& store) {
    TensorMap m;
    for (auto& kv : store) {
        Tensor t;
        t.dims = kv.second.dims;
        t.data = kv.second.data.data();
        t.size = kv.second.data.size();
        m[kv.first] = t;
    }
    return m;
}

int get_tensor(const TensorMap& m, const std::string& name, size_t expect, std::vector<float>& out) {
    auto it = m.find(name);
    if (it == m.end()) return tz_fail(TZ_EPARSE, "weights: missing tensor " + name);
    if (it->second.size != expect)
        return tz_fail(TZ_EPARSE, "weights: tensor " + name + " has " + std::to_string(it->second.size) + " elements, expected " +
                                      std::to_string(expect));
    out.resize(expect);
    memcpy(out.data(), it->second.data, 4 * expect);
    return TZ_OK;
}

// Copies of a weight build.  A build may run on a thread of its own while the search thread is capturing or replaying graphs on
// its (blocking) streams (tz_net_load_prepare): a copy on the legacy stream would make that stream depend on the capturing one -
// an error that also breaks the capture - so a build's copies go through a non-blocking stream of the building thread
// (UploadStream), waited for copy by copy (the sources are short-lived host vectors).
thread_local hipStream_t g_upload_stream = nullptr;
struct UploadStream {
    hipStream_t prev = nullptr, mine = nullptr;
    UploadStream() {
        prev = g_upload_stream;
        if (hipStreamCreateWithFlags(&mine, hipStreamNonBlocking) == hipSuccess) g_upload_stream = mine;
    }
    ~UploadStream() {
        g_upload_stream = prev;
        if (mine) (void)hipStreamDestroy(mine);
    }
};
int copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (!g_upload_stream) {
        TZ_HIP(hipMemcpy(dst, src, bytes, kind));
        return TZ_OK;
    }
    TZ_HIP(hipMemcpyAsync(dst, src, bytes, kind, g_upload_stream));
    TZ_HIP(hipStreamSynchronize(g_upload_stream));
    return TZ_OK;
}

template <typename T>
int upload(const std::vector<T>& h, T** dev) {
    T* d = nullptr;
    TZ_HIP(hipMalloc(&d, h.size() * sizeof(T)));
    const int rc = copy_sync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    if (rc) {
        (void)hipFree(d);
        return rc;
    }
    *dev = d;
    return TZ_OK;
}

// Build device weights of one conv / linear layer from w[cout][cin][taps] (+ optional BN fold).
// in_perm (optional): source input index for each of my input indices.
int build_layer(int precision, int taps, int cin, int cout, int cout_mult, const std::vector<float>& w,
                const std::vector<float>& scale, const std::vector<float>& bias, const std::vector<int>* in_perm,
                ConvW* L, bool with_lo = false, bool with_c8 = false, bool with_c6 = false) {
    L->taps = taps;
    L->cin = cin;
    L->cin_pad = (cin + 31) / 32 * 32;
    L->cout = cout;
    L->cout_pad = (cout + cout_mult - 1) / cout_mult * cout_mult;
    auto W = [&](int co, int ci, int t) -> float {
        const int src = in_perm ? (*in_perm)[ci] : ci;
        return w[((size_t)co * cin + src) * taps + t] * (scale.empty() ? 1.0f : scale[co]);
    };
    std::vector<float> b(L->cout_pad, 0.0f);
    for (int co = 0; co < cout; co++) b[co] = bias.empty() ? 0.0f : bias[co];
    int rc = upload(b, &L->bias);
    if (rc) return rc;
    if (precision != TZ_PREC_F32) {
        const int kc_total = L->cin_pad / 32, ct_total = L->cout_pad / 16;
        std::vector<uint16_t> p((size_t)taps * kc_total * ct_total * 64 * 8, 0), plo;
        if (with_lo) plo.assign(p.size(), 0);
        for (int t = 0; t < taps; t++)
            for (int kc = 0; kc < kc_total; kc++)
                for (int ct = 0; ct < ct_total; ct++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int co = ct * 16 + (lane & 15);
                        if (co >= cout) continue;
                        for (int j = 0; j < 8; j++) {
                            const int ci = kc * 32 + 8 * (lane >> 4) + j;
                            if (ci >= cin) continue;
                            const float wv = W(co, ci, t);
                            // fp16 storage: a (BatchNorm-folded) weight beyond the largest finite half would become inf
                            if (prec_is_f16(precision) && !(wv >= -65504.0f && wv <= 65504.0f))
                                return tz_fail(TZ_ENUMERIC, "weights: a folded weight is outside fp16's range (or not finite); load this model with TZ_PREC_BF16");
                            const size_t at = ((((size_t)t * kc_total + kc) * ct_total + ct) * 64 + lane) * 8 + j;
                            p[at] = prec_is_f16(precision) ? f2h(wv) : f2bf(wv);
                            if (with_lo) {   // lo half of the split-precision pair: (w - hi) * 2^11, |lo| <= |w|
                                const _Float16 hi = (_Float16)wv;
                                plo[at] = f2h((wv - (float)hi) * 2048.0f);
                            }
                        }
                    }
        if (with_lo && (rc = upload(plo, &L->w_lo))) return rc;
        if (with_c8) {
            // FP8 fragments of the two correction products (k_loop_c8).  One 16x16x128 MFMA covers 128 input channels: lane
            // (row = lane & 15, q = lane >> 4) holds 32 of them, bytes 0..15 = channels 128m + 16q .. and bytes 16..31 =
            // channels 128m + 64 + 16q .. (the two FP8 image planes of the half, piece q of each).  s = the power of two that
            // brings the layer's largest |w| to [128, 256): E4M3 then keeps 4 significant bits down to 2^-13 of it.
            if (L->cin_pad % 128) return tz_fail(TZ_EINVAL, "weights: TZ_PREC_F16C8 needs input channels in multiples of 128");
            float wmax = 0.0f;
            for (int co = 0; co < cout; co++)
                for (int ci = 0; ci < cin; ci++)
                    for (int t = 0; t < taps; t++) wmax = std::max(wmax, fabsf(W(co, ci, t)));
            const float sw = wmax > 0.0f ? exp2f(floorf(log2f(256.0f / wmax))) : 1.0f;
            L->c8_scale = 1.0f / (2048.0f * sw * C8_SX);
            const int m_total = L->cin_pad / 128;
            std::vector<unsigned char> p8((size_t)taps * m_total * ct_total * 2 * 64 * 32, 0);
            for (int t = 0; t < taps; t++)
                for (int mh = 0; mh < m_total; mh++)
                    for (int ct = 0; ct < ct_total; ct++)
                        for (int lane = 0; lane < 64; lane++) {
                            const int co = ct * 16 + (lane & 15), q = lane >> 4;
                            if (co >= cout) continue;
                            for (int i = 0; i < 32; i++) {
                                const int ci = 128 * mh + 64 * (i >> 4) + 16 * q + (i & 15);
                                if (ci >= cin) continue;
                                const float wv = W(co, ci, t);
                                const _Float16 hi = (_Float16)wv;
                                const size_t at = (((((size_t)t * m_total + mh) * ct_total + ct) * 2) * 64 + lane) * 32 + i;
                                p8[at] = tz_f32_to_e4m3((wv - (float)hi) * 2048.0f * sw);
                                p8[at + 64 * 32] = tz_f32_to_e4m3((float)hi * sw);
                            }
                        }
            if ((rc = upload(p8, &L->w8))) return rc;
        }
        if (with_c6) {
            // E2M3 records of the two correction products (k_loop_c6, tz_nn_c6.hip).  One record per (tap, K-half G of 128 input
            // channels, output tile ct): [term: wl, wh][lane][16 B] + [lane][8 B] (the 24-B operand string of a lane, element i in
            // bits 6 i ..), then [lane] one dword of scales (byte 0: wl's block, byte 1: wh's block; E8M0, 2^(byte - 127)).  Lane
            // (row = lane & 15, q = lane >> 4) holds the block of 32 input places 128 G + 32 (i / 8) + 8 q + i % 8: the 8 elements of
            // lane group q in each of the half's four fp16 fragments.  (in_perm has already moved the channels to their places.)
            if (L->cin_pad % 128) return tz_fail(TZ_EINVAL, "weights: TZ_PREC_F16C6 needs input channels in multiples of 128");
            const int g_total = L->cin_pad / 128;
            constexpr size_t REC = 3328;
            std::vector<unsigned char> p6((size_t)taps * g_total * ct_total * REC, 0);
            for (int t = 0; t < taps; t++)
                for (int g = 0; g < g_total; g++)
                    for (int ct = 0; ct < ct_total; ct++) {
                        unsigned char* rec = p6.data() + (((size_t)t * g_total + g) * ct_total + ct) * REC;
                        for (int lane = 0; lane < 64; lane++) {
                            const int co = ct * 16 + (lane & 15), q = lane >> 4;
                            float lo[32], hi[32], amax_l = 0.0f, amax_h = 0.0f;
                            for (int i = 0; i < 32; i++) {
                                const int ci = 128 * g + 32 * (i >> 3) + 8 * q + (i & 7);
                                const float wv = (co < cout && ci < cin) ? W(co, ci, t) : 0.0f;
                                const _Float16 h = (_Float16)wv;
                                hi[i] = (float)h;
                                lo[i] = wv - (float)h;
                                amax_h = std::max(amax_h, fabsf(hi[i]));
                                amax_l = std::max(amax_l, fabsf(lo[i]));
                            }
                            const uint32_t bl = tz_e2m3_block_scale_byte(amax_l), bh = tz_e2m3_block_scale_byte(amax_h);
                            uint8_t cl[32], ch[32];
                            for (int i = 0; i < 32; i++) {
                                cl[i] = tz_f32_to_e2m3(ldexpf(lo[i], 127 - (int)bl));
                                ch[i] = tz_f32_to_e2m3(ldexpf(hi[i], 127 - (int)bh));
                            }
                            uint32_t sl[6], sh[6];
                            tz_pack_fp6x32(cl, sl);
                            tz_pack_fp6x32(ch, sh);
                            memcpy(rec + 0 * 1536 + lane * 16, sl, 16);
                            memcpy(rec + 0 * 1536 + 1024 + lane * 8, sl + 4, 8);
                            memcpy(rec + 1 * 1536 + lane * 16, sh, 16);
                            memcpy(rec + 1 * 1536 + 1024 + lane * 8, sh + 4, 8);
                            const uint32_t sw = bl | (bh << 8);
                            memcpy(rec + 3072 + lane * 4, &sw, 4);
                        }
                    }
            if ((rc = upload(p6, &L->w8))) return rc;
        }
        return upload(p, &L->w_mfma);
    }
    std::vector<float> f((size_t)taps * cin * cout);
    for (int t = 0; t < taps; t++)
        for (int ci = 0; ci < cin; ci++)
            for (int co = 0; co < cout; co++) f[((size_t)t * cin + ci) * cout + co] = W(co, ci, t);
    return upload(f, &L->w_f32);
}

void free_layer(ConvW* L) {
    if (L->w_mfma) (void)hipFree(L->w_mfma);
    if (L->w_lo) (void)hipFree(L->w_lo);
    if (L->w8) (void)hipFree(L->w8);
    if (L->w_f32) (void)hipFree(L->w_f32);
    if (L->bias) (void)hipFree(L->bias);
    *L = ConvW();
}

int bn_fold(const TensorMap& m, const std::string& p, int c, std::vector<float>& scale, std::vector<float>& bias) {
    std::vector<float> g, b, mean, var;
    int rc;
    if ((rc = get_tensor(m, p + ".weight", c, g))) return rc;
    if ((rc = get_tensor(m, p + ".bias", c, b))) return rc;
    if ((rc = get_tensor(m, p + ".running_mean", c, mean))) return rc;
    if ((rc = get_tensor(m, p + ".running_var", c, var))) return rc;
    scale.resize(c);
    bias.resize(c);
    for (int i = 0; i < c; i++) {
        const float s = g[i] / sqrtf(var[i] + 1e-5f);  // tch BatchNormConfig::default eps
        scale[i] = s;
        bias[i] = b[i] - mean[i] * s;
    }
    return TZ_OK;
}

struct NetWeights {  // everything tz_net_load_weights replaces, so a failed load changes nothing
    ConvW conv_in, policy;
    std::vector<ConvW> res;
    uint16_t* tower_w = nullptr;
    uint16_t* tower_w_lo = nullptr;
    unsigned char* tower_w8 = nullptr;
    float* c8_scales = nullptr;
    float* tower_bias = nullptr;
    float* heads = nullptr;
    ConvW rnd[2][3];
    uint16_t* rndw[3] = {nullptr, nullptr, nullptr};
    float* rndb[3] = {nullptr, nullptr, nullptr};
    float rnd_min = 0.f, rnd_max = 1.f;
    float* simhash = nullptr;
};

void free_weights(NetWeights& w) {
    free_layer(&w.conv_in);
    free_layer(&w.policy);
    for (auto& l : w.res) free_layer(&l);
    w.res.clear();
    if (w.tower_w) (void)hipFree(w.tower_w);
    if (w.tower_w_lo) (void)hipFree(w.tower_w_lo);
    if (w.tower_w8) (void)hipFree(w.tower_w8);
    if (w.c8_scales) (void)hipFree(w.c8_scales);
    if (w.tower_bias) (void)hipFree(w.tower_bias);
    w.tower_w = nullptr;
    w.tower_w_lo = nullptr;
    w.tower_w8 = nullptr;
    w.c8_scales = nullptr;
    w.tower_bias = nullptr;
    if (w.heads) (void)hipFree(w.heads);
    w.heads = nullptr;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) free_layer(&w.rnd[a][b]);
    for (int l = 0; l < 3; l++) {
        if (w.rndw[l]) (void)hipFree(w.rndw[l]);
        if (w.rndb[l]) (void)hipFree(w.rndb[l]);
        w.rndw[l] = nullptr;
        w.rndb[l] = nullptr;
    }
    if (w.simhash) (void)hipFree(w.simhash);
    w.simhash = nullptr;
}

int build_weights(tz_net* net, const TensorMap& m, NetWeights& W) {
    const int nn = net->nn, cin = net->cin, prec = net->precision;
    std::vector<float> w, scale, bias;
    int rc;
    if ((rc = get_tensor(m, "core.input_conv2d.weight", (size_t)FILTERS * cin * 9, w))) return rc;
    if ((rc = bn_fold(m, "core.batch_norm", FILTERS, scale, bias))) return rc;
    const bool split = prec == TZ_PREC_F16X2, c8 = prec == TZ_PREC_F16C8, c6 = prec == TZ_PREC_F16C6;
    // TZ_PREC_F16C6: every conv that reads the tower's image takes its input channels from their places there (c6_channel_at)
    std::vector<int> c6_perm(FILTERS);
    for (int pos = 0; pos < FILTERS; pos++) c6_perm[pos] = c6_channel_at(pos);
    const std::vector<int>* tower_perm = c6 ? &c6_perm : nullptr;
    // the first conv of TZ_PREC_F16C8 / F16C6 is the split form too: its input planes are built in the kernel, its k-loop is 9 steps
    if ((rc = build_layer(prec, 9, cin, FILTERS, 256, w, scale, bias, nullptr, &W.conv_in, split || c8 || c6))) return rc;
    W.res.resize(2 * net->blocks);
    {
        // the tower's layers are independent (BatchNorm folding, fragment order, fp16 / FP8 conversion, upload: 9 ms each on one host
        // thread): a hot reload stalls the self-play loop for as long as this takes, so they are spread over a few threads
        const int nl = 2 * net->blocks;
        const int workers = std::max(1, std::min(nl, std::min(16, (int)std::thread::hardware_concurrency())));
        std::atomic<int> next{0}, failed{TZ_OK};
        std::string first_error;
        std::mutex mu;
        const bool background = g_upload_stream != nullptr;   // a build beside a running search: every thread copies on a stream of its own
        auto work = [&]() {
            if (hipSetDevice(net->device) != hipSuccess) {
                failed = TZ_EDEVICE;
                return;
            }
            std::unique_ptr<UploadStream> upload_stream(background ? new UploadStream() : nullptr);
            if (upload_stream && !upload_stream->mine) {   // no stream of its own: a legacy-stream copy would break the search thread's capture
                std::lock_guard<std::mutex> lk(mu);
                if (failed == TZ_OK) {
                    failed = TZ_EDEVICE;
                    first_error = "weights: cannot create an upload stream for a build thread";
                }
                return;
            }
            std::vector<float> lw, lscale, lbias;
            for (int l = next++; l < nl && failed == TZ_OK; l = next++) {
                const std::string p = "core.res_block_" + std::to_string(l / 2) + ((l & 1) ? ".b" : ".a");
                int r = get_tensor(m, p + ".conv2d.weight", (size_t)FILTERS * FILTERS * 9, lw);
                if (!r) r = bn_fold(m, p + ".batch_norm", FILTERS, lscale, lbias);
                if (!r) r = build_layer(prec, 9, FILTERS, FILTERS, 256, lw, lscale, lbias, tower_perm, &W.res[l], split, c8, c6);
                if (r) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (failed == TZ_OK) {
                        failed = r;
                        first_error = tz_last_error();   // the error text is per thread: hand it to the caller's
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < workers; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        if (failed != TZ_OK) return tz_fail(failed, first_error.empty() ? "weights: a tower layer could not be built" : first_error);
    }
    if (prec != TZ_PREC_F32 && net->blocks > 0) {  // the fused tower kernel reads all layers from one buffer
        const size_t layer_elems = (size_t)9 * 8 * 16 * 64 * 8, nl = W.res.size();
        TZ_HIP(hipMalloc(&W.tower_w, nl * layer_elems * 2));
        if (split) TZ_HIP(hipMalloc(&W.tower_w_lo, nl * layer_elems * 2));
        const size_t layer_bytes8 = c6 ? (size_t)9 * 2 * 16 * 3328 : (size_t)9 * 2 * 16 * 2 * 64 * 32;
        if (c8 || c6) TZ_HIP(hipMalloc(&W.tower_w8, nl * layer_bytes8));
        TZ_HIP(hipMalloc(&W.tower_bias, nl * FILTERS * sizeof(float)));
        for (size_t l = 0; l < nl; l++) {
            if ((rc = copy_sync(W.tower_w + l * layer_elems, W.res[l].w_mfma, layer_elems * 2, hipMemcpyDeviceToDevice))) return rc;
            if (split && (rc = copy_sync(W.tower_w_lo + l * layer_elems, W.res[l].w_lo, layer_elems * 2, hipMemcpyDeviceToDevice))) return rc;
            if ((c8 || c6) && (rc = copy_sync(W.tower_w8 + l * layer_bytes8, W.res[l].w8, layer_bytes8, hipMemcpyDeviceToDevice))) return rc;
            if ((rc = copy_sync(W.tower_bias + l * FILTERS, W.res[l].bias, FILTERS * sizeof(float), hipMemcpyDeviceToDevice))) return rc;
        }
    }
    if ((rc = get_tensor(m, "policy.conv2d.weight", (size_t)net->pol_ch * FILTERS * 9, w))) return rc;
    if ((rc = get_tensor(m, "policy.conv2d.bias", net->pol_ch, bias))) return rc;
    if ((rc = build_layer(prec, 9, FILTERS, net->pol_ch, net->pol_stride, w, {}, bias, tower_perm, &W.policy, split, c8, c6))) return rc;
    if (c8) {
        std::vector<float> cs;
        for (auto& l : W.res) cs.push_back(l.c8_scale);
        cs.push_back(W.policy.c8_scale);
        if ((rc = upload(cs, &W.c8_scales))) return rc;
    }
    // heads
    std::vector<float> hv, hu, lv, lu, t1;
    if ((rc = get_tensor(m, "value.conv2d.weight", FILTERS, hv))) return rc;
    if ((rc = get_tensor(m, "ube.conv2d.weight", FILTERS, hu))) return rc;
    if ((rc = get_tensor(m, "value.linear.weight", nn, lv))) return rc;
    if ((rc = get_tensor(m, "ube.linear.weight", nn, lu))) return rc;
    std::vector<float> hw;
    hw.insert(hw.end(), hv.begin(), hv.end());
    hw.insert(hw.end(), hu.begin(), hu.end());
    hw.insert(hw.end(), lv.begin(), lv.end());
    hw.insert(hw.end(), lu.begin(), lu.end());
    for (const char* nm : {"value.conv2d.bias", "ube.conv2d.bias", "value.linear.bias", "ube.linear.bias"}) {
        if ((rc = get_tensor(m, nm, 1, t1))) return rc;
        hw.push_back(t1[0]);
    }
    if ((rc = upload(hw, &W.heads))) return rc;
    if (net->has_rnd) {
        const int in_size = cin * nn;
        std::vector<int> perm(in_size);  // my index px*cin + c  <-  reference index c*nn + px
        for (int px = 0; px < nn; px++)
            for (int c = 0; c < cin; c++) perm[px * cin + c] = c * nn + px;
        const char* nets[2] = {"rnd_learning", "rnd_target"};
        const char* layers[3] = {"input_linear", "hidden_linear", "final_linear"};
        const int dims[4] = {in_size, 1024, 1024, 512};
        for (int a = 0; a < 2; a++)
            for (int l = 0; l < 3; l++) {
                const std::string p = std::string(nets[a]) + "." + layers[l];
                if ((rc = get_tensor(m, p + ".weight", (size_t)dims[l + 1] * dims[l], w))) return rc;
                if ((rc = get_tensor(m, p + ".bias", dims[l + 1], bias))) return rc;
                if ((rc = build_layer(prec, 1, dims[l], dims[l + 1], 256, w, {}, bias, l == 0 ? &perm : nullptr, &W.rnd[a][l])))
                    return rc;
            }
        if (prec != TZ_PREC_F32) {
            // layer 1: both networks read the same input -> one GEMM with 2048 outputs
            std::vector<float> wa, wb, ba, bb;
            if ((rc = get_tensor(m, "rnd_learning.input_linear.weight", (size_t)1024 * in_size, wa))) return rc;
            if ((rc = get_tensor(m, "rnd_target.input_linear.weight", (size_t)1024 * in_size, wb))) return rc;
            if ((rc = get_tensor(m, "rnd_learning.input_linear.bias", 1024, ba))) return rc;
            if ((rc = get_tensor(m, "rnd_target.input_linear.bias", 1024, bb))) return rc;
            wa.insert(wa.end(), wb.begin(), wb.end());
            ba.insert(ba.end(), bb.begin(), bb.end());
            ConvW cat;
            if ((rc = build_layer(prec, 1, in_size, 2048, 256, wa, {}, ba, &perm, &cat))) return rc;
            W.rndw[0] = cat.w_mfma;
            W.rndb[0] = cat.bias;
            // layers 2 and 3: [net][fragments] for grouped launches
            for (int l = 1; l < 3; l++) {
                const size_t welems = (size_t)(W.rnd[0][l].cin_pad / 32) * (W.rnd[0][l].cout_pad / 16) * 512;
                const size_t belems = W.rnd[0][l].cout_pad;
                TZ_HIP(hipMalloc(&W.rndw[l], 2 * welems * 2));
                TZ_HIP(hipMalloc(&W.rndb[l], 2 * belems * sizeof(float)));
                for (int a = 0; a < 2; a++) {
                    if ((rc = copy_sync(W.rndw[l] + a * welems, W.rnd[a][l].w_mfma, welems * 2, hipMemcpyDeviceToDevice))) return rc;
                    if ((rc = copy_sync(W.rndb[l] + a * belems, W.rnd[a][l].bias, belems * sizeof(float), hipMemcpyDeviceToDevice))) return rc;
                }
            }
        }
        if ((rc = get_tensor(m, "min", 1, t1))) return rc;
        W.rnd_min = t1[0];
        if ((rc = get_tensor(m, "max", 1, t1))) return rc;
        W.rnd_max = t1[0];
    }
    if (net->has_hash) {
        if ((rc = get_tensor(m, "simhash_matrix", (size_t)cin * nn * 32, w))) return rc;
        if ((rc = upload(w, &W.simhash))) return rc;
    }
    return TZ_OK;
}

// Two tilings of the same kernel:
//   cfg 0: 8 waves x 32 channels on P = ppt_for(N) boards   (1 workgroup per CU)
//   cfg 1: 4 waves x 64 channels on P = ppt_small(N) boards (2 workgroups per CU: the tile load / epilogue of one
//          overlaps the MFMA phase of the other, and every activation fragment feeds 4 MFMAs instead of 2)
__host__ __device__ constexpr int ppt_small(int nb) { return nb == 1 ? 64 : nb == 3 ? 7 : nb == 4 ? 6 : nb == 5 ? 4 : 3; }

int conv_cfg() {
    static int cfg = -1;
    if (cfg < 0) {
        const char* e = getenv("TZ_CONV_CFG");
        cfg = e ? atoi(e) : 0;  // cfg 1 measured slower on 5x5: its weight stream (2x the bytes per CU) saturates the L2->CU path
    }
    return cfg;
}

template <int NB, int P, int NW, int RN, int TAPS, bool FROM_STATE, bool SINGLE, int ABL = 0, int LAYOUT = 1, typename ET = __bf16, int OCC = 2>
int launch_conv(const ConvArgs& a, int max_positions, int n_blocks_y, hipStream_t st) {
    constexpr int NN = NB * NB, RT = (P * NN + 15) / 16, LROWS = RT * 16 + 8;
    const size_t smem = LdsImg<LAYOUT>::bytes(LROWS);
    auto kern = conv_mfma_kernel<NB, P, RN, TAPS, FROM_STATE, ABL, SINGLE, NW, LAYOUT, ET, OCC>;
    static bool attr_done[64] = {};   // per device: a function attribute belongs to the device's copy of the module
    int attr_dev = 0;
    TZ_HIP(hipGetDevice(&attr_dev));
    bool& attr_set = attr_done[attr_dev & 63];
    if (!attr_set) {
        TZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((max_positions + P - 1) / P, n_blocks_y, a.groups > 1 ? a.groups : 1);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("conv launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

// Linear layers of the RND MLP, `groups` independent networks per launch (128 positions per workgroup)
int linear_grouped_bf16(int precision, const uint16_t* w, const float* bias, int groups, int cin_pad, int cout_pad, const void* in,
                        int in_stride, int in_z, void* out, int out_stride, int out_z, bool relu, bool out_f32,
                        const int32_t* count_dev, int count_host, int max_positions, hipStream_t st) {
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in = in;
    a.count_dev = count_dev;
    a.count_host = count_host;
    a.w = w;
    a.bias = bias;
    a.out = out;
    a.cin_pad = cin_pad;
    a.kc_total = cin_pad / 32;
    a.ct_total = cout_pad / 16;
    a.out_stride = out_stride;
    a.cin_real = cin_pad;
    a.relu = relu;
    a.out_f32 = out_f32;
    a.in_stride = in_stride;
    a.groups = groups;
    a.in_z = in_z;
    a.w_z_frags = a.kc_total * a.ct_total;
    a.bias_z = cout_pad;
    a.out_z = out_z;
    if (max_positions <= 128) {   // the Agent surface's batches: 32 instead of 128 rows per workgroup, four times the workgroups (8 -> 32 per layer at batch 128)
        if (prec_is_f16(precision)) return launch_conv<1, 32, 8, 2, 1, false, false, 0, 1, _Float16>(a, max_positions, cout_pad / 256, st);
        return launch_conv<1, 32, 8, 2, 1, false, false>(a, max_positions, cout_pad / 256, st);
    }
    if (prec_is_f16(precision)) return launch_conv<1, 128, 8, 2, 1, false, false, 0, 1, _Float16>(a, max_positions, cout_pad / 256, st);
    return launch_conv<1, 128, 8, 2, 1, false, false>(a, max_positions, cout_pad / 256, st);
}

template <int NB, typename ET>
int conv_board(const ConvArgs& a, int cout_pad, bool from_state, int max_positions, hipStream_t st) {
    constexpr int PA = ppt_for(NB), PB = ppt_small(NB);
    const bool wide = cout_pad % 256 == 0;
    const int by = wide ? cout_pad / 256 : cout_pad / 128;
    if (conv_cfg() == 0) {
        if (from_state) return launch_conv<NB, PA, 8, 2, 9, true, false, 0, 1, ET>(a, max_positions, by, st);
        if (wide) return launch_conv<NB, PA, 8, 2, 9, false, true, 0, 1, ET>(a, max_positions, by, st);
        return launch_conv<NB, PA, 8, 1, 9, false, true, 0, 1, ET>(a, max_positions, by, st);
    }
    if (from_state) return launch_conv<NB, PB, 4, 4, 9, true, false, 0, 1, ET>(a, max_positions, by, st);
    if (wide) return launch_conv<NB, PB, 4, 4, 9, false, true, 0, 1, ET>(a, max_positions, by, st);
    return launch_conv<NB, PB, 4, 2, 9, false, true, 0, 1, ET>(a, max_positions, by, st);
}

template <typename ET>
int conv_dispatch(int n, bool board, const ConvArgs& a, int cout_pad, bool from_state, int max_positions, hipStream_t st) {
    if (!board) return launch_conv<1, 64, 8, 2, 1, false, false, 0, 1, ET>(a, max_positions, cout_pad / 256, st);
    switch (n) {
        case 3: return conv_board<3, ET>(a, cout_pad, from_state, max_positions, st);
        case 4: return conv_board<4, ET>(a, cout_pad, from_state, max_positions, st);
        case 5: return conv_board<5, ET>(a, cout_pad, from_state, max_positions, st);
        case 6: return conv_board<6, ET>(a, cout_pad, from_state, max_positions, st);
    }
    return tz_fail(TZ_EINVAL, "conv: unsupported board size");
}

// one bf16 layer.  in==nullptr -> first layer from packed states.
int conv_bf16(tz_net* net, const ConvW& L, const void* in, const tz_state* states, const int32_t* gidx,
              const int32_t* count_dev, int count_host, int max_positions, const void* residual, void* out,
              int out_stride, bool relu, bool out_f32, bool board, hipStream_t st) {
    ConvArgs a;
    a.in = in;
    a.states = states;
    a.game_index = gidx;
    a.count_dev = count_dev;
    a.count_host = count_host;
    a.w = L.w_mfma;
    a.bias = L.bias;
    a.residual = residual;
    a.out = out;
    a.cin_pad = L.cin_pad;
    a.kc_total = L.cin_pad / 32;
    a.ct_total = L.cout_pad / 16;
    a.out_stride = out_stride;
    a.cin_real = L.cin;
    a.relu = relu;
    a.out_f32 = out_f32;
    a.has_res = residual != nullptr;
    a.in_stride = L.cin_pad;
    a.groups = 1;
    a.in_z = a.w_z_frags = a.bias_z = a.out_z = 0;
    if (prec_is_f16(net->precision)) return conv_dispatch<_Float16>(net->n, board, a, L.cout_pad, in == nullptr, max_positions, st);
    return conv_dispatch<__bf16>(net->n, board, a, L.cout_pad, in == nullptr, max_positions, st);
}

#ifdef TZ_ABLATIONS
template <int NB>
int launch_tower4x2(const TowerArgs& a, int max_positions, hipStream_t st) {
    constexpr int P = ppt_for(NB), NN = NB * NB, RT = (P * NN + 15) / 16, LROWS = RT * 16 + 8;
    const size_t smem = (size_t)LROWS * LDS_ROWB * 8 + (size_t)9 * RT * 64 * 4;
    auto kern = tower4x2_mfma_kernel<NB, P>;
    static bool attr_done[64] = {};   // per device: a function attribute belongs to the device's copy of the module
    int attr_dev = 0;
    TZ_HIP(hipGetDevice(&attr_dev));
    bool& attr_set = attr_done[attr_dev & 63];
    if (!attr_set) {
        TZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((max_positions + P - 1) / P), dim3(512), smem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("tower4x2 launch: ") + hipGetErrorString(e));
    return TZ_OK;
}
#endif  // TZ_ABLATIONS

template <int NB, int OPT = 0, typename ET = __bf16>
int launch_tower(const TowerArgs& a, int max_positions, hipStream_t st) {
    constexpr int P = ppt_for(NB), NN = NB * NB, RT = (P * NN + 15) / 16, LROWS = RT * 16 + 8;
    const size_t smem = (size_t)LROWS * LDS_ROWB * 8 + ((OPT & 4096) ? (size_t)9 * RT * 64 * 4 : 0);
    auto kern = tower_mfma_kernel<NB, P, OPT, ET>;
    static bool attr_done[64] = {};   // per device: a function attribute belongs to the device's copy of the module
    int attr_dev = 0;
    TZ_HIP(hipGetDevice(&attr_dev));
    bool& attr_set = attr_done[attr_dev & 63];
    if (!attr_set) {
        TZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((max_positions + P - 1) / P), dim3(512), smem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("tower launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

int tower_bf16(tz_net* net, const void* in, void* out, const int32_t* count_dev, int count_host, int max_positions,
               hipStream_t st) {
    TowerArgs a;
    a.in = reinterpret_cast<const uint16_t*>(in);
    a.out = reinterpret_cast<uint16_t*>(out);
    a.w = net->tower_w;
    a.bias = net->tower_bias;
    a.count_dev = count_dev;
    a.count_host = count_host;
    a.nlayers = 2 * net->blocks;
    const bool h = prec_is_f16(net->precision);
    switch (net->n) {
        // OPT 4224 = tap table in LDS + pinned issue order: the loop the net kernel ships (see tz_debug_tower_bench)
        case 3: return h ? launch_tower<3, 4224, _Float16>(a, max_positions, st) : launch_tower<3, 4224>(a, max_positions, st);
        case 4: return h ? launch_tower<4, 4224, _Float16>(a, max_positions, st) : launch_tower<4, 4224>(a, max_positions, st);
        case 5: return h ? launch_tower<5, 4224, _Float16>(a, max_positions, st) : launch_tower<5, 4224>(a, max_positions, st);
        case 6: return h ? launch_tower<6, 4224, _Float16>(a, max_positions, st) : launch_tower<6, 4224>(a, max_positions, st);
    }
    return tz_fail(TZ_EINVAL, "tower: unsupported board size");
}

#endif  // TZ_NN_SPLIT_TU

int net_fused_mode() {  // 2: whole trunk + heads in one launch (default); 1: fused tower only; 0: one launch per conv
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("TZ_TOWER");
        mode = e ? atoi(e) : 2;
    }
    return mode;
}

constexpr int NET_SPLIT_MAX_GROUPS = 64;           // several CUs per board group: at most 64 groups x 4 members = one workgroup per CU
constexpr size_t NET_SPLIT_PLANE_BYTES = 7680;     // the largest image plane of those forms (5x5, 4 boards: 120 rows of 64 B)
constexpr int NET_SPLIT_MAX_POSITIONS = 4 * NET_SPLIT_MAX_GROUPS;   // 64 groups of 1, 2 or 4 boards (with 8 the one-CU forms are faster: 0.63 against 0.69 ms per simulation of 512 games)

template <int NB, int RNP, typename ET, bool PERM, int P = ppt_for(NB), int SP = 0, int ABL = 0, int TT = 0, int NW = 8, int SPLIT = 1>
int launch_net(const NetArgs& a, int max_positions, hipStream_t st) {
    constexpr int RT = RowMap<NB, P, PERM>::RT, LROWS = RT * 16 + 8;
    constexpr bool ROWTAB = SP != 0 && NB == 6 && P == 4;   // as in the kernel: 8-bit table, no remainder planes
    constexpr size_t tap_bytes = TT ? (size_t)9 * RT * RowMap<NB, P, PERM>::PPT * sizeof(int) : (size_t)9 * RT * 64 * (ROWTAB ? 1 : SP == 2 ? sizeof(uint16_t) : sizeof(int));
    constexpr size_t smem = (size_t)LROWS * LDS_ROWB * (SP == 2 ? (NB == 6 ? 16 : 20) : SP ? 16 : 8) + 2 * RT * 16 * sizeof(float) + tap_bytes + (SPLIT > 1 ? 16 : 0);  // image + head scratch + tap table (+ the exchange's flag)
    static_assert(smem <= 160 * 1024, "net kernel: the LDS image does not fit a CU");
    auto kern = net_mfma_kernel<NB, P, RNP, ET, PERM, SP, ABL, TT, NW, SPLIT>;
    static bool attr_done[64] = {};   // per device: a function attribute belongs to the device's copy of the module
    int attr_dev = 0;
    TZ_HIP(hipGetDevice(&attr_dev));
    bool& attr_set = attr_done[attr_dev & 63];
    if (!attr_set) {
        TZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    int blocks = (max_positions + P - 1) / P;
    if (SPLIT > 1) {   // whole octets of groups (members of a group sit 8 blocks apart), counters at zero
        const int groups = (blocks + 7) / 8 * 8;
        if (groups > NET_SPLIT_MAX_GROUPS || !a.xch || !a.xch_count) return tz_fail(TZ_EINVAL, "net launch: the several-CU form is for small batches");
        TZ_HIP(hipMemsetAsync(a.xch_count, 0, (size_t)groups * 128, st));
        blocks = groups * SPLIT;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), smem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("net launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

// TZ_NET_ROWS=board selects the board-major row order without tap skipping (A/B; the default is the square-major order
// with the all-zero (tap, row tile) pairs skipped wherever 16 % boards-per-workgroup == 0: 3x3, 5x5, 6x6)
bool net_square_major() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TZ_NET_ROWS");
        v = (e && !strcmp(e, "board")) ? 0 : 1;
    }
    return v == 1;
}

// Boards per workgroup for small batches.  A workgroup takes the same time whether its image holds 1 board or 8 (it is
// bound by its own stream of the 47 MB of weights or by its MFMAs, whichever is longer), so a batch that cannot fill the
// 256 CUs with 8-board workgroups is spread over more, smaller ones: the Agent surface at the reference's batch of 128
// (tz_net_eval) and small searches (puzzles, compete, tei-sized batches).  The result of a position does not depend on
// the choice (same k order).  TZ_NET_P=full keeps the full-size workgroups (A/B).
int net_small_p(int max_positions) {
    static const bool full = getenv("TZ_NET_P") && !strcmp(getenv("TZ_NET_P"), "full");
    if (full) return 0;
    return max_positions <= 256 ? 1 : max_positions <= 512 ? 2 : max_positions <= 1024 ? 4 : 0;
}

#if defined(TZ_NN_C6_TU)
}  // namespace   (tz_nn_c6.hip goes on from here)
#elif defined(TZ_NN_SPLIT_TU)
}  // namespace

// The split-precision launches (TZ_PREC_F16X2: sp = 1, TZ_PREC_F16C8: sp = 2) live in their own translation unit so that the two
// compile side by side (tz_nn_split.hip includes this file with TZ_NN_SPLIT_TU defined); `net_args` is the caller's NetArgs.
int tz_nn_launch_split(int sp, int n, const void* net_args, int max_positions, hipStream_t st) {
    const NetArgs& a = *static_cast<const NetArgs*>(net_args);
    if (sp == 1) {
        // split precision: hi and lo planes share the 160 KB, so half the boards per workgroup (5x5: 4 boards, square-major
        // rows, 14 of 63 (tap, tile) pairs skipped; the other sizes board-major)
        if (n == 5 && a.xch && max_positions <= NET_SPLIT_MAX_POSITIONS) {   // up to 256 positions: four CUs per group of one, two or four boards
            if (max_positions <= NET_SPLIT_MAX_GROUPS) return launch_net<5, 1, _Float16, false, 1, 1, 0, 0, 4, 4>(a, max_positions, st);
            if (max_positions <= 2 * NET_SPLIT_MAX_GROUPS) return launch_net<5, 1, _Float16, false, 2, 1, 0, 0, 4, 4>(a, max_positions, st);
            return launch_net<5, 1, _Float16, false, 4, 1, 0, 0, 4, 4>(a, max_positions, st);
        }
        switch (n) {
            case 3: return launch_net<3, 1, _Float16, false, 8, 1>(a, max_positions, st);
            case 4: return launch_net<4, 1, _Float16, false, 6, 1>(a, max_positions, st);
            case 5: return launch_net<5, 1, _Float16, true, 4, 1>(a, max_positions, st);
            case 6:   // from 1024 positions on: 4 boards, square-major rows (12 of 81 pairs skipped), table of 8-bit rows
                if (net_square_major() && max_positions >= 1024) return launch_net<6, 2, _Float16, true, 4, 1>(a, max_positions, st);
                return launch_net<6, 2, _Float16, false, 2, 1>(a, max_positions, st);
        }
        return tz_fail(TZ_EINVAL, "net: unsupported board size");
    }
    // fp16 + FP8 corrections: same workgroups as the split form (20 planes instead of 16: the tap table halves)
#ifdef TZ_ABLATIONS
    if (n == 5 && getenv("TZ_NET_ABL") && atoi(getenv("TZ_NET_ABL")) == 16) return launch_net<5, 1, _Float16, true, 4, 2, 16>(a, max_positions, st);
    if (n == 5 && getenv("TZ_NET_ABL") && atoi(getenv("TZ_NET_ABL")) == 32) return launch_net<5, 1, _Float16, true, 4, 2, 32>(a, max_positions, st);
    if (n == 5 && getenv("TZ_NET_ABL") && atoi(getenv("TZ_NET_ABL")) == 48) return launch_net<5, 1, _Float16, true, 4, 2, 48>(a, max_positions, st);
#endif
    // small batches (the Agent surface at the reference's batch of 128): one or two boards per workgroup, as in the fp16 form
    if (n == 5 && net_small_p(max_positions) == 1) return launch_net<5, 1, _Float16, false, 1, 2>(a, max_positions, st);
    if (n == 5 && net_small_p(max_positions) == 2) return launch_net<5, 1, _Float16, false, 2, 2>(a, max_positions, st);
    switch (n) {
        case 3: return launch_net<3, 1, _Float16, false, 8, 2>(a, max_positions, st);
        case 4: return launch_net<4, 1, _Float16, false, 6, 2>(a, max_positions, st);
        case 5: return launch_net<5, 1, _Float16, true, 4, 2>(a, max_positions, st);
        case 6:   // the same; on 6x6 the block input is carried as hi + one FP8 byte (no room for the remainder planes)
            if (net_square_major() && max_positions >= 1024) return launch_net<6, 2, _Float16, true, 4, 2>(a, max_positions, st);
            return launch_net<6, 2, _Float16, false, 2, 2>(a, max_positions, st);
    }
    return tz_fail(TZ_EINVAL, "net: unsupported board size");
}
#else   // the first translation unit: everything else

bool net_p6_four() {
    static const bool four = getenv("TZ_NET_P6") && !strcmp(getenv("TZ_NET_P6"), "4");
    return four;
}

template <typename ET>
int net_fused_et(tz_net* net, const NetArgs& a, int max_positions, hipStream_t st) {
    const bool sq = net_square_major();
    const int small = net_small_p(max_positions);
    // the Agent surface at the reference's batch (tz_net_eval, count known on the host): four CUs per board group
    if (net->n == 5 && a.xch && max_positions <= NET_SPLIT_MAX_POSITIONS && sizeof(ET) == 2) {   // 64 groups x 4 CUs: 1, 2 or 4 boards per group
        if (max_positions <= NET_SPLIT_MAX_GROUPS) return launch_net<5, 1, ET, false, 1, 0, 0, 0, 4, 4>(a, max_positions, st);
        if (max_positions <= 2 * NET_SPLIT_MAX_GROUPS) return launch_net<5, 1, ET, false, 2, 0, 0, 0, 4, 4>(a, max_positions, st);
        return launch_net<5, 1, ET, false, 4, 0, 0, 0, 4, 4>(a, max_positions, st);
    }
    if (net->n == 4 && a.xch && max_positions <= 2 * NET_SPLIT_MAX_GROUPS && sizeof(ET) == 2) {   // 4x4: 1 or 2 boards per group of four CUs
        if (max_positions <= NET_SPLIT_MAX_GROUPS) return launch_net<4, 1, ET, false, 1, 0, 0, 0, 4, 4>(a, max_positions, st);
        return launch_net<4, 1, ET, false, 2, 0, 0, 0, 4, 4>(a, max_positions, st);
    }
    if (net->n == 6 && a.xch && max_positions <= 2 * NET_SPLIT_MAX_GROUPS && sizeof(ET) == 2) {   // 6x6: 1 or 2 boards per group of four CUs
        if (max_positions <= NET_SPLIT_MAX_GROUPS) return launch_net<6, 2, ET, false, 1, 0, 0, 0, 4, 4>(a, max_positions, st);
        return launch_net<6, 2, ET, false, 2, 0, 0, 0, 4, 4>(a, max_positions, st);
    }
    if (net->n == 5 && small == 1) return launch_net<5, 1, ET, false, 1>(a, max_positions, st);
    if (net->n == 5 && small == 2) return launch_net<5, 1, ET, false, 2>(a, max_positions, st);
    if (net->n == 5 && small == 4) return launch_net<5, 1, ET, false, 4>(a, max_positions, st);
    if (net->n == 6 && small == 1) return launch_net<6, 2, ET, false, 1>(a, max_positions, st);
    if (net->n == 6 && small == 2) return launch_net<6, 2, ET, false, 2>(a, max_positions, st);
#ifdef TZ_ABLATIONS   // TZ_NET_ABL=<bits>: the ablated twins of the shipped 5x5 kernel (tools/net_ablation.py)
    if (const char* e = getenv("TZ_NET_ABL")) {
        if (net->n == 5 && sq && small == 0 && sizeof(ET) == 2) {
            switch (atoi(e)) {
                case 1: return launch_net<5, 1, ET, true, 8, 0, 1>(a, max_positions, st);
                case 2: return launch_net<5, 1, ET, true, 8, 0, 2>(a, max_positions, st);
                case 3: return launch_net<5, 1, ET, true, 8, 0, 3>(a, max_positions, st);
                case 4: return launch_net<5, 1, ET, true, 8, 0, 4>(a, max_positions, st);
                case 8: return launch_net<5, 1, ET, true, 8, 0, 8>(a, max_positions, st);
                case 16: return launch_net<5, 1, ET, true, 8, 0, 16>(a, max_positions, st);
                case 32: return launch_net<5, 1, ET, true, 8, 0, 32>(a, max_positions, st);
                case 48: return launch_net<5, 1, ET, true, 8, 0, 48>(a, max_positions, st);
                case 64: return launch_net<5, 1, ET, true, 8, 0, 64>(a, max_positions, st);
                case 80: return launch_net<5, 1, ET, true, 8, 0, 80>(a, max_positions, st);
                default: break;
            }
        }
    }
#endif
    switch (net->n) {
        case 3: return sq ? launch_net<3, 1, ET, true>(a, max_positions, st) : launch_net<3, 1, ET, false>(a, max_positions, st);
        case 4: return launch_net<4, 1, ET, false>(a, max_positions, st);
        case 5:
#ifdef TZ_ABLATIONS   // A/B: the 5x5 kernel on the compact tap table + fragment ring of the 6x6 form (TZ_NET_TT=1)
            if (sq && getenv("TZ_NET_TT") && atoi(getenv("TZ_NET_TT")) == 1) return launch_net<5, 1, ET, true, 8, 0, 0, 1>(a, max_positions, st);
#endif
#ifdef TZ_ABLATIONS   // A/B: four waves of 64 channels, one per SIMD (TZ_NET_W4=1)
            if (sq && getenv("TZ_NET_W4") && atoi(getenv("TZ_NET_W4")) == 1) return launch_net<5, 1, ET, true, 8, 0, 0, 0, 4>(a, max_positions, st);
#endif
            return sq ? launch_net<5, 1, ET, true>(a, max_positions, st) : launch_net<5, 1, ET, false>(a, max_positions, st);
        case 6:
            // 8 boards per workgroup (18 row tiles, compact tap table, ring loop) once that still gives every CU a workgroup:
            // the weight stream per MFMA halves against the 4-board form.  TZ_NET_P6=4 keeps the 4-board form (A/B).
            if (sq && max_positions >= 2048 && !net_p6_four()) return launch_net<6, 2, ET, true, 8, 0, 0, 1>(a, max_positions, st);
            return sq ? launch_net<6, 2, ET, true>(a, max_positions, st) : launch_net<6, 2, ET, false>(a, max_positions, st);
    }
    return tz_fail(TZ_EINVAL, "net: unsupported board size");
}

// RND input from the fused kernel: net5 only (in_size 800 = 25 squares x 32 planes, no K padding, 8 planes per 16-B store)
bool net_fused_writes_rnd_input(const tz_net* net) {
    static const bool separate = getenv("TZ_RND_PREP_KERNEL") != nullptr;   // A/B: the separate rnd_prep_state_kernel
    if (net->eval_split) return false;   // tz_net_eval at small batches: the RND MLP runs beside the net kernel, from its own input kernel
    return !separate && net->has_rnd && net->n == 5 && net->cin % 8 == 0 && (net->cin * net->nn) % 32 == 0;
}

int net_fused(tz_net* net, const tz_state* states, const int32_t* gidx, const int32_t* count_dev, int count_host,
              int max_positions, hipStream_t st) {
    NetArgs a;
    a.states = states;
    a.game_index = gidx;
    a.count_dev = count_dev;
    a.count_host = count_host;
    a.w_in = net->conv_in.w_mfma;
    a.bias_in = net->conv_in.bias;
    a.cin_real = net->conv_in.cin;
    a.kc_in = net->conv_in.cin_pad / 32;
    a.w = net->tower_w;
    a.bias = net->tower_bias;
    a.nlayers = 2 * net->blocks;
    a.w_pol = net->policy.w_mfma;
    a.bias_pol = net->policy.bias;
    a.policy_out = net->policy_out;
    a.pol_stride = net->pol_stride;
    a.heads = net->heads;
    a.value = net->value;
    a.ube = net->ube;
    a.rnd_in = nullptr;
    a.rnd_stride = 0;
    if (net_fused_writes_rnd_input(net)) {   // the fused kernel also writes the RND networks' normalised input
        a.rnd_in = net->rnd_in;
        a.rnd_stride = net->cin * net->nn;
    }
    a.w_in_lo = net->conv_in.w_lo;
    a.w_lo = net->tower_w_lo;
    a.w_pol_lo = net->policy.w_lo;
    a.w8 = net->tower_w8;
    a.w_pol8 = net->policy.w8;
    a.c8_scales = net->c8_scales;
    a.dbg = nullptr;
    a.seeds = net->seeds;
    // several CUs per board group (net_mfma_kernel SPLIT): 5x5 up to 256 positions, 4x4 and 6x6 up to 128, 16-bit storage — tz_net_eval at the reference's batch
    // and searches of such widths (the reference's selfplay runs 128 games: selfplay/src/main.rs:37).  TZ_NET_SPLIT=0: one CU per group (A/B)
    a.xch = nullptr;
    a.xch_count = nullptr;
    static const bool split_off = getenv("TZ_NET_SPLIT") && !strcmp(getenv("TZ_NET_SPLIT"), "0");
    const bool split16 = (net->precision == TZ_PREC_F16 || net->precision == TZ_PREC_BF16) &&
                         ((net->n == 5 && max_positions <= NET_SPLIT_MAX_POSITIONS) || ((net->n == 6 || net->n == 4) && max_positions <= 2 * NET_SPLIT_MAX_GROUPS));
    const bool splitx2 = net->precision == TZ_PREC_F16X2 && net->n == 5 && max_positions <= NET_SPLIT_MAX_POSITIONS;   // the tolerance-holding arithmetic at the reference's batch
    if (!split_off && (split16 || splitx2)) {
        if (!net->xch) {   // first use; a search's first two steps run outside its graph capture, so this is never inside one
            TZ_HIP(hipMalloc(&net->xch, (size_t)NET_SPLIT_MAX_GROUPS * 2 * 16 * NET_SPLIT_PLANE_BYTES));   // 16 planes: the hi / lo form
            TZ_HIP(hipMalloc((void**)&net->xch_count, (size_t)NET_SPLIT_MAX_GROUPS * 128));
        }
        a.xch = static_cast<unsigned char*>(net->xch);
        a.xch_count = net->xch_count;
    }
#ifdef TZ_ABLATIONS
    if (getenv("TZ_NET_ABL") && (atoi(getenv("TZ_NET_ABL")) == 8 || (atoi(getenv("TZ_NET_ABL")) & 16))) {
        if (!net->dbg_buf) TZ_HIP(hipMalloc(&net->dbg_buf, (size_t)65536 * 4 * sizeof(unsigned long long)));
        a.dbg = reinterpret_cast<unsigned long long*>(net->dbg_buf);
        net->dbg_groups = (max_positions + (prec_is_split(net->precision) ? 3 : 7)) / (prec_is_split(net->precision) ? 4 : 8);
    }
#endif
    if (net->precision == TZ_PREC_F16C6) return tz_nn_launch_c6(net->n, &a, max_positions, st);   // tz_nn_c6.hip
    if (prec_is_split(net->precision)) return tz_nn_launch_split(net->precision == TZ_PREC_F16C8 ? 2 : 1, net->n, &a, max_positions, st);   // tz_nn_split.hip
    if (net->precision == TZ_PREC_F16) return net_fused_et<_Float16>(net, a, max_positions, st);
    return net_fused_et<__bf16>(net, a, max_positions, st);
}

int conv_f32(tz_net* net, const ConvW& L, const float* in, int in_stride, const int32_t* count_dev, int count_host,
             int max_positions, const float* residual, float* out, int out_stride, bool relu, bool board,
             hipStream_t st) {
    const int nn = board ? net->nn : 1;
    const size_t total = (size_t)max_positions * nn * L.cout;
    const int blocks = (int)((total + 255) / 256);
#define TZ_F32_CASE(NBV)                                                                                       \
    case NBV:                                                                                                  \
        conv_f32_kernel<NBV><<<blocks, 256, 0, st>>>(in, L.w_f32, L.bias, residual, out, count_dev, count_host, \
                                                     L.taps, L.cin, in_stride, L.cout, out_stride, relu ? 1 : 0); \
        break;
    switch (board ? net->n : 1) {
        TZ_F32_CASE(1)
        TZ_F32_CASE(3)
        TZ_F32_CASE(4)
        TZ_F32_CASE(5)
        TZ_F32_CASE(6)
        default: return tz_fail(TZ_EINVAL, "conv_f32: unsupported board size");
    }
#undef TZ_F32_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("conv_f32 launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

int encode(tz_net* net, const tz_state* states, const int32_t* gidx, const int32_t* count_dev, int count_host,
           int max_positions, hipStream_t st) {
    const int total = max_positions * net->nn, blocks = (total + 127) / 128;
    switch (net->n) {
        case 3: encode_kernel<3><<<blocks, 128, 0, st>>>(states, gidx, count_dev, count_host, net->cin, net->planes); break;
        case 4: encode_kernel<4><<<blocks, 128, 0, st>>>(states, gidx, count_dev, count_host, net->cin, net->planes); break;
        case 5: encode_kernel<5><<<blocks, 128, 0, st>>>(states, gidx, count_dev, count_host, net->cin, net->planes); break;
        case 6: encode_kernel<6><<<blocks, 128, 0, st>>>(states, gidx, count_dev, count_host, net->cin, net->planes); break;
        default: return tz_fail(TZ_EINVAL, "encode: unsupported board size");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("encode launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

}  // namespace

// game_to_tensor for `count` positions into NHWC fp32 planes (repr.rs:8-135); used by the trainer (tz_learn.hip)
int tz_nn_encode_planes(int n, int cin, const tz_state* states_dev, int count, float* planes_dev, hipStream_t st) {
    const int total = count * n * n, blocks = (total + 127) / 128;
    switch (n) {
        case 3: encode_kernel<3><<<blocks, 128, 0, st>>>(states_dev, nullptr, nullptr, count, cin, planes_dev); break;
        case 4: encode_kernel<4><<<blocks, 128, 0, st>>>(states_dev, nullptr, nullptr, count, cin, planes_dev); break;
        case 5: encode_kernel<5><<<blocks, 128, 0, st>>>(states_dev, nullptr, nullptr, count, cin, planes_dev); break;
        case 6: encode_kernel<6><<<blocks, 128, 0, st>>>(states_dev, nullptr, nullptr, count, cin, planes_dev); break;
        default: return tz_fail(TZ_EINVAL, "encode: unsupported board size");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("encode launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

// ---------------------------------------------------------------------------------------------
int tz_net_ensure_batch(tz_net* net, int batch) {
    if (batch <= net->max_batch) return TZ_OK;
    TZ_HIP(hipSetDevice(net->device));
    TZ_HIP(hipStreamSynchronize(net->stream));
    void** bufs[] = {&net->act_a, &net->act_b, &net->act_c, (void**)&net->planes, (void**)&net->policy_out,
                     (void**)&net->value, (void**)&net->ube, (void**)&net->variance, (void**)&net->aux,
                     &net->rnd_in, &net->rnd_h1, &net->rnd_h2, (void**)&net->rnd_out, &net->seeds};
    for (auto b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    const size_t esz = net->precision != TZ_PREC_F32 ? 2 : 4;
    const size_t rows = (size_t)batch * net->nn;
    TZ_HIP(hipMalloc(&net->act_a, rows * FILTERS * esz));
    TZ_HIP(hipMalloc(&net->act_b, rows * FILTERS * esz));
    TZ_HIP(hipMalloc(&net->act_c, rows * FILTERS * esz));
    TZ_HIP(hipMalloc(&net->planes, rows * net->cin * sizeof(float)));
    TZ_HIP(hipMalloc(&net->policy_out, rows * net->pol_stride * sizeof(float)));
    TZ_HIP(hipMalloc(&net->value, batch * sizeof(float)));
    TZ_HIP(hipMalloc(&net->ube, batch * sizeof(float)));
    TZ_HIP(hipMalloc(&net->variance, batch * sizeof(float)));
    TZ_HIP(hipMalloc(&net->aux, batch * sizeof(float)));
    if (net->precision == TZ_PREC_F16C6) TZ_HIP(hipMalloc(&net->seeds, tz_nn_c6_seed_bytes(net->n, batch)));
    if (net->has_rnd) {
        const size_t in_pad = (size_t)(net->cin * net->nn + 31) / 32 * 32;
        TZ_HIP(hipMalloc(&net->rnd_in, (size_t)batch * in_pad * esz));
        TZ_HIP(hipMalloc(&net->rnd_h1, (size_t)batch * 2048 * esz));
        TZ_HIP(hipMalloc(&net->rnd_h2, (size_t)batch * 2048 * esz));
        TZ_HIP(hipMalloc(&net->rnd_out, (size_t)2 * batch * 512 * sizeof(float)));
    }
    net->max_batch = batch;
    return TZ_OK;
}

int tz_net_forward_device(tz_net* net, const tz_state* states, const int32_t* gidx, const int32_t* count_dev,
                          int count_host, int max_positions, hipStream_t st, NetOut* out) {
    if (!net->loaded) return tz_fail(TZ_ESTATE, "network has no weights loaded");
    if (max_positions > net->max_batch) return tz_fail(TZ_EINVAL, "forward: batch exceeds the network's buffers");
    int rc;
    const bool bf = net->precision != TZ_PREC_F32;  // 16-bit MFMA path (bf16 or fp16 storage)
    const int nn = net->nn;
    const bool need_planes = !bf;   // the MFMA path builds its planes in LDS; SimHash reads the packed states (simhash_state_kernel)
    if (need_planes && (rc = encode(net, states, gidx, count_dev, count_host, max_positions, st))) return rc;
    void *x = net->act_a, *t = net->act_b, *y = net->act_c;
    const bool split = prec_is_split(net->precision);
    const int rnd_in_pad = (net->cin * nn + 31) / 32 * 32;
    // tz_net_eval at small batches (several CUs per board group): the RND MLP does not wait for the net kernel — it reads the packed
    // states, not the trunk — so it runs on a stream of its own beside it (45 of a call's 460 us), and only rnd_finish joins the two
    const bool rnd_ahead = net->eval_split && net->has_rnd && bf && net->n == 5 && net->stream_rnd;
    auto rnd_mlp = [&](hipStream_t rs) -> int {
        // 3 launches: input planes straight from the packed states, layer 1 of both nets as one GEMM,
        // layers 2 and 3 as grouped launches (blockIdx.z = net)
        const bool fused_wrote_it = net->blocks > 0 && net->tower_w && (net_fused_mode() == 2 || split) && net_fused_writes_rnd_input(net);
        switch (net->n) {
            case 5:
                if (fused_wrote_it) break;   // net_mfma_kernel wrote x / sum(x^2) while it built the planes
                if (prec_is_f16(net->precision))
                    rnd_prep_state_kernel<5, _Float16><<<max_positions, 64, 0, rs>>>(states, gidx, count_dev, count_host, net->cin, rnd_in_pad, (_Float16*)net->rnd_in);
                else
                    rnd_prep_state_kernel<5, __bf16><<<max_positions, 64, 0, rs>>>(states, gidx, count_dev, count_host, net->cin, rnd_in_pad, (__bf16*)net->rnd_in);
                break;
            default: return tz_fail(TZ_EINVAL, "RND is a net5 (5x5) feature");
        }
        float* o = net->rnd_out;
        if ((rc = linear_grouped_bf16(net->precision, net->rndw[0], net->rndb[0], 1, rnd_in_pad, 2048, net->rnd_in, rnd_in_pad, 0, net->rnd_h1, 2048, 0,
                                      true, false, count_dev, count_host, max_positions, rs)))
            return rc;
        if ((rc = linear_grouped_bf16(net->precision, net->rndw[1], net->rndb[1], 2, 1024, 1024, net->rnd_h1, 2048, 1024, net->rnd_h2, 2048, 1024,
                                      true, false, count_dev, count_host, max_positions, rs)))
            return rc;
        if ((rc = linear_grouped_bf16(net->precision, net->rndw[2], net->rndb[2], 2, 1024, 512, net->rnd_h2, 2048, 1024, o, 1024, 512, false, true,
                                      count_dev, count_host, max_positions, rs)))
            return rc;
        return TZ_OK;
    };

    if (split && !(net->blocks > 0 && (net->tower_w_lo || net->tower_w8)))
        return tz_fail(TZ_EINVAL, "forward: TZ_PREC_F16X2 / TZ_PREC_F16C8 / TZ_PREC_F16C6 need a network with at least one residual block");
    if (bf && net->blocks > 0 && net->tower_w && (net_fused_mode() == 2 || split)) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (net->profile) {
            TZ_HIP(hipEventCreate(&e0));
            TZ_HIP(hipEventCreate(&e1));
            TZ_HIP(hipEventRecord(e0, st));
        }
        if (rnd_ahead) {
            TZ_HIP(hipEventRecord(net->ev_in, st));            // the states have arrived (tz_net_eval's copy is ahead of this on st)
            TZ_HIP(hipStreamWaitEvent(net->stream_rnd, net->ev_in, 0));
            if ((rc = rnd_mlp(net->stream_rnd))) return rc;
            TZ_HIP(hipEventRecord(net->ev_rnd, net->stream_rnd));
        }
        if ((rc = net_fused(net, states, gidx, count_dev, count_host, max_positions, st))) return rc;
        if (net->profile) {
            TZ_HIP(hipEventRecord(e1, st));
            net->conv_events.push_back({e0, e1});
            net->conv_launches += 1;
        }
    } else if (bf) {
        if ((rc = conv_bf16(net, net->conv_in, nullptr, states, gidx, count_dev, count_host, max_positions, nullptr, x, FILTERS,
                            true, false, true, st)))
            return rc;
        if (net->blocks > 0 && net->tower_w && net_fused_mode() >= 1) {
            // the residual tower as one persistent launch
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (net->profile) {
                TZ_HIP(hipEventCreate(&e0));
                TZ_HIP(hipEventCreate(&e1));
                TZ_HIP(hipEventRecord(e0, st));
            }
            if ((rc = tower_bf16(net, x, y, count_dev, count_host, max_positions, st))) return rc;
            if (net->profile) {
                TZ_HIP(hipEventRecord(e1, st));
                net->conv_events.push_back({e0, e1});
                net->conv_launches += 1;
            }
            std::swap(x, y);
        } else {
            for (int b = 0; b < net->blocks; b++) {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (net->profile) {
                    TZ_HIP(hipEventCreate(&e0));
                    TZ_HIP(hipEventCreate(&e1));
                    TZ_HIP(hipEventRecord(e0, st));
                }
                if ((rc = conv_bf16(net, net->res[2 * b], x, nullptr, nullptr, count_dev, count_host, max_positions, nullptr, t,
                                    FILTERS, true, false, true, st)))
                    return rc;
                if ((rc = conv_bf16(net, net->res[2 * b + 1], t, nullptr, nullptr, count_dev, count_host, max_positions, x, y,
                                    FILTERS, true, false, true, st)))
                    return rc;
                if (net->profile) {
                    TZ_HIP(hipEventRecord(e1, st));
                    net->conv_events.push_back({e0, e1});
                    net->conv_launches += 2;
                }
                std::swap(x, y);
            }
        }
        if ((rc = conv_bf16(net, net->policy, x, nullptr, nullptr, count_dev, count_host, max_positions, nullptr,
                            net->policy_out, net->pol_stride, false, true, true, st)))
            return rc;
        if (prec_is_f16(net->precision))
            heads_kernel<_Float16><<<max_positions, 64, 0, st>>>((const _Float16*)x, net->heads, count_dev, count_host, nn, net->value, net->ube);
        else
            heads_kernel<__bf16><<<max_positions, 64, 0, st>>>((const __bf16*)x, net->heads, count_dev, count_host, nn, net->value, net->ube);
    } else {
        float *fx = (float*)x, *ft = (float*)t, *fy = (float*)y;
        if ((rc = conv_f32(net, net->conv_in, net->planes, net->cin, count_dev, count_host, max_positions, nullptr, fx, FILTERS,
                           true, true, st)))
            return rc;
        for (int b = 0; b < net->blocks; b++) {
            if ((rc = conv_f32(net, net->res[2 * b], fx, FILTERS, count_dev, count_host, max_positions, nullptr, ft, FILTERS, true,
                               true, st)))
                return rc;
            if ((rc = conv_f32(net, net->res[2 * b + 1], ft, FILTERS, count_dev, count_host, max_positions, fx, fy, FILTERS, true,
                               true, st)))
                return rc;
            std::swap(fx, fy);
        }
        if ((rc = conv_f32(net, net->policy, fx, FILTERS, count_dev, count_host, max_positions, nullptr, net->policy_out,
                           net->pol_stride, false, true, st)))
            return rc;
        heads_kernel<float><<<max_positions, 64, 0, st>>>(fx, net->heads, count_dev, count_host, nn, net->value, net->ube);
    }
    // local uncertainty
    if (net->has_rnd) {
        const int in_size = net->cin * nn, in_pad = (in_size + 31) / 32 * 32;
        float* outs[2] = {net->rnd_out, net->rnd_out + (size_t)net->max_batch * 512};
        if (bf) {
            if (!rnd_ahead && (rc = rnd_mlp(st))) return rc;
            if (rnd_ahead) TZ_HIP(hipStreamWaitEvent(st, net->ev_rnd, 0));
            float* o = net->rnd_out;
            rnd_finish_kernel<<<max_positions, 64, 0, st>>>(o, o + 512, net->ube, count_dev, count_host, 512, 1024, net->rnd_min,
                                                            net->rnd_max, net->variance);
        } else {
            rnd_prep_kernel<float><<<max_positions, 64, 0, st>>>(net->planes, count_dev, count_host, in_size, in_pad, (float*)net->rnd_in);
            for (int a = 0; a < 2; a++) {
                if ((rc = conv_f32(net, net->rnd[a][0], (float*)net->rnd_in, in_pad, count_dev, count_host, max_positions, nullptr,
                                   (float*)net->rnd_h1, 1024, true, false, st)))
                    return rc;
                if ((rc = conv_f32(net, net->rnd[a][1], (float*)net->rnd_h1, 1024, count_dev, count_host, max_positions, nullptr,
                                   (float*)net->rnd_h2, 1024, true, false, st)))
                    return rc;
                if ((rc = conv_f32(net, net->rnd[a][2], (float*)net->rnd_h2, 1024, count_dev, count_host, max_positions, nullptr,
                                   outs[a], 512, false, false, st)))
                    return rc;
            }
        }
        if (!bf)
            rnd_finish_kernel<<<max_positions, 64, 0, st>>>(outs[0], outs[1], net->ube, count_dev, count_host, 512, 512, net->rnd_min,
                                                            net->rnd_max, net->variance);
    } else if (net->has_hash) {
        if ((rc = launch_simhash_state(net->n, states, gidx, net->simhash, net->bitset, count_dev, count_host, max_positions, net->aux, nullptr, st)))
            return rc;
        plain_variance_kernel<<<(max_positions + 255) / 256, 256, 0, st>>>(net->ube, net->aux, count_dev, count_host, net->variance);
    } else {
        plain_variance_kernel<<<(max_positions + 255) / 256, 256, 0, st>>>(net->ube, nullptr, count_dev, count_host, net->variance);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("forward launch: ") + hipGetErrorString(e));
    out->policy = net->policy_out;
    out->policy_stride = net->pol_stride;
    out->value = net->value;
    out->variance = net->variance;
    return TZ_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int tz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tz_net_create(int board_n, int arch, int device_id, int precision, int blocks, tz_net** out) {
    if (!out) return tz_fail(TZ_EINVAL, "tz_net_create: null out");
    int n = board_n;
    if (arch == TZ_ARCH_NET5) n = 5;
    else if (arch == TZ_ARCH_NET4_SIMHASH) n = 4;
    else if (arch == TZ_ARCH_NET6_SIMHASH) n = 6;
    else if (arch != TZ_ARCH_TEST) return tz_fail(TZ_EINVAL, "tz_net_create: unknown architecture");
    if (board_n && board_n != n) return tz_fail(TZ_EINVAL, "tz_net_create: board size does not match the architecture");
    if (n < 3 || n > 6) return tz_fail(TZ_EINVAL, "tz_net_create: board size must be 3..6");
    if (precision != TZ_PREC_BF16 && precision != TZ_PREC_F32 && precision != TZ_PREC_F16 && precision != TZ_PREC_F16X2 && precision != TZ_PREC_F16C8 &&
        precision != TZ_PREC_F16C6)
        return tz_fail(TZ_EINVAL, "tz_net_create: bad precision");
    if (precision == TZ_PREC_F16C6 && n != 5 && n != 6) return tz_fail(TZ_EINVAL, "tz_net_create: TZ_PREC_F16C6 is built for the 5x5 and 6x6 networks");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return tz_fail(TZ_EDEVICE, "tz_net_create: no HIP device available (the HIP path has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return tz_fail(TZ_EINVAL, "tz_net_create: bad device id");
    TZ_HIP(hipSetDevice(device_id));
    tz_net* net = new tz_net();
    net->n = n;
    net->nn = n * n;
    net->arch = arch;
    net->device = device_id;
    net->precision = precision;
    net->blocks = arch == TZ_ARCH_NET5 ? 20 : arch == TZ_ARCH_TEST ? blocks : 16;
    net->cin = 2 * ((3 + (n - 1) + (n + 1)) + 2) + 2;
    net->cin_pad = (net->cin + 31) / 32 * 32;
    net->pol_ch = 3 + 4 * ((1 << n) - 2);
    net->pol_stride = (net->pol_ch + 127) / 128 * 128;
    net->ppt = ppt_for(n);
    net->has_rnd = arch == TZ_ARCH_NET5;
    net->has_hash = arch == TZ_ARCH_NET4_SIMHASH || arch == TZ_ARCH_NET6_SIMHASH;
    if (hipStreamCreate(&net->stream) != hipSuccess) {
        delete net;
        return tz_fail(TZ_EDEVICE, "tz_net_create: hipStreamCreate failed");
    }
    if (net->has_hash) {  // BitBox of 2^32 bits, net6_simhash.rs:138
        if (hipMalloc(&net->bitset, (size_t)1 << 29) != hipSuccess || hipMemset(net->bitset, 0, (size_t)1 << 29) != hipSuccess) {
            delete net;
            return tz_fail(TZ_ENOMEM, "tz_net_create: cannot allocate the 512 MiB SimHash set");
        }
    }
    *out = net;
    return TZ_OK;
}

// Replaces the network's weights by `store` (all or nothing: a failure leaves the previous weights active,
// selfplay/src/main.rs:112-115) and keeps the store as the host copy of the VarStore (save / clone / load_partial).
// A load in two halves.  Preparing (BatchNorm folding, fragment order, conversions, uploads into fresh buffers) reads only what
// tz_net_create fixed, so it may run on any thread while the network is evaluating; committing waits for the device and swaps the
// pointers, and must not run beside a forward of this network.
struct tz_pending_weights {
    TensorStore store;
    NetWeights W;
    std::string path;
    tz_net* net = nullptr;
};

static int net_prepare_store(tz_net* net, TensorStore&& store, tz_pending_weights** out, bool background) {
    TZ_HIP(hipSetDevice(net->device));
    tz_pending_weights* p = new tz_pending_weights();
    p->store = std::move(store);
    p->net = net;
    const TensorMap m = view_of(p->store);
    std::unique_ptr<UploadStream> upload_stream(background ? new UploadStream() : nullptr);
    if (upload_stream && !upload_stream->mine) {   // never fall back to the legacy stream beside a running search (see UploadStream)
        delete p;
        return tz_fail(TZ_EDEVICE, "tz_net_load_prepare: cannot create the upload stream");
    }
    const int rc = build_weights(net, m, p->W);
    if (rc) {
        free_weights(p->W);
        delete p;
        return rc;
    }
    *out = p;
    return TZ_OK;
}

static int net_commit_pending(tz_net* net, tz_pending_weights* p) {
    NetWeights& W = p->W;
    TZ_HIP(hipSetDevice(net->device));
    TZ_HIP(hipDeviceSynchronize());  // searches run this net on their own streams; the uploads above have landed too
    NetWeights old;
    old.conv_in = net->conv_in;
    old.policy = net->policy;
    old.res = net->res;
    old.tower_w = net->tower_w;
    old.tower_w_lo = net->tower_w_lo;
    old.tower_w8 = net->tower_w8;
    old.c8_scales = net->c8_scales;
    old.tower_bias = net->tower_bias;
    old.heads = net->heads;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) old.rnd[a][b] = net->rnd[a][b];
    for (int l = 0; l < 3; l++) {
        old.rndw[l] = net->rndw[l];
        old.rndb[l] = net->rndb[l];
        net->rndw[l] = W.rndw[l];
        net->rndb[l] = W.rndb[l];
    }
    old.simhash = net->simhash;
    net->conv_in = W.conv_in;
    net->policy = W.policy;
    net->res = W.res;
    net->tower_w = W.tower_w;
    net->tower_w_lo = W.tower_w_lo;
    net->tower_w8 = W.tower_w8;
    net->c8_scales = W.c8_scales;
    net->tower_bias = W.tower_bias;
    net->heads = W.heads;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) net->rnd[a][b] = W.rnd[a][b];
    net->rnd_min = W.rnd_min;
    net->rnd_max = W.rnd_max;
    net->simhash = W.simhash;
    free_weights(old);
    net->store = std::move(p->store);
    net->loaded = true;
    net->weights_gen++;
    delete p;
    return TZ_OK;
}

static int net_apply_store(tz_net* net, TensorStore&& store) {
    tz_pending_weights* p = nullptr;
    const int rc = net_prepare_store(net, std::move(store), &p, false);
    return rc ? rc : net_commit_pending(net, p);
}

static bool file_exists(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb");
    if (f) fclose(f);
    return f != nullptr;
}
static std::string sibling(const char* path, const char* name) {
    const std::string p(path);
    const size_t slash = p.find_last_of('/');
    return (slash == std::string::npos ? std::string() : p.substr(0, slash + 1)) + name;
}

int tz_net_load_weights_mem(tz_net* net, const void* data, size_t bytes) {
    if (!net || !data) return tz_fail(TZ_EINVAL, "tz_net_load_weights: null argument");
    TensorStore store;
    const unsigned char* p = (const unsigned char*)data;
    int rc;
    if (bytes >= 4 && !memcmp(p, "PK\x03\x04", 4)) {   // a LibTorch archive held in memory
        NamedTensors named;
        if ((rc = ot_read_archive(p, bytes, named))) return rc;
        if ((rc = ot_canonical_names(named, store))) return rc;
    } else if ((rc = tzw_parse(p, bytes, store))) {
        return rc;
    }
    return net_apply_store(net, std::move(store));
}

int tz_net_load_weights(tz_net* net, const char* path) {
    if (!net || !path) return tz_fail(TZ_EINVAL, "tz_net_load_weights: null argument");
    TensorStore store;
    int rc = weights_read_file(path, store);
    if (rc) return rc;
    if ((rc = net_apply_store(net, std::move(store)))) return rc;
    // SimHash nets keep their set beside the model (net6_simhash.rs:173-190)
    if (net->has_hash) {
        const std::string bits = sibling(path, "bitvec.bin");
        if (file_exists(bits)) return tz_net_load_bitset(net, bits.c_str());
    }
    return TZ_OK;
}

int tz_net_load_prepare(tz_net* net, const char* path, tz_pending_weights** out) {
    if (!net || !path || !out) return tz_fail(TZ_EINVAL, "tz_net_load_prepare: null argument");
    TensorStore store;
    int rc = weights_read_file(path, store);
    if (rc) return rc;
    tz_pending_weights* p = nullptr;
    if ((rc = net_prepare_store(net, std::move(store), &p, true))) return rc;
    p->path = path;
    *out = p;
    return TZ_OK;
}

int tz_net_load_commit(tz_net* net, tz_pending_weights* p) {
    if (!net || !p || p->net != net) return tz_fail(TZ_EINVAL, "tz_net_load_commit: these weights were not prepared for this network");
    const std::string path = p->path;
    int rc = net_commit_pending(net, p);
    if (rc) return rc;
    if (net->has_hash) {   // SimHash nets keep their set beside the model (net6_simhash.rs:173-190)
        const std::string bits = sibling(path.c_str(), "bitvec.bin");
        if (file_exists(bits)) return tz_net_load_bitset(net, bits.c_str());
    }
    return TZ_OK;
}

int tz_net_load_discard(tz_pending_weights* p) {
    if (!p) return TZ_OK;
    if (p->net) (void)hipSetDevice(p->net->device);
    free_weights(p->W);
    delete p;
    return TZ_OK;
}

int tz_net_load_partial(tz_net* net, const char* path, char* missing_out, int missing_cap, int* n_missing_out) {
    if (!net || !path) return tz_fail(TZ_EINVAL, "tz_net_load_partial: null argument");
    if (!net->loaded) return tz_fail(TZ_ESTATE, "tz_net_load_partial: initialise the network first (tz_net_init_random or a full load)");
    TensorStore found, merged = net->store;
    int rc = weights_read_file(path, found);
    if (rc) return rc;
    std::string missing;
    int n_missing = 0;
    for (auto& kv : merged) {
        auto it = found.find(kv.first);
        if (it != found.end() && it->second.data.size() == kv.second.data.size()) {
            kv.second = it->second;
        } else {
            missing += kv.first + "\n";
            n_missing++;
        }
    }
    if ((rc = net_apply_store(net, std::move(merged)))) return rc;
    if (n_missing_out) *n_missing_out = n_missing;
    if (missing_out && missing_cap > 0) {
        const size_t k = std::min(missing.size(), (size_t)missing_cap - 1);
        memcpy(missing_out, missing.data(), k);
        missing_out[k] = 0;
    }
    return TZ_OK;
}

int tz_net_save(tz_net* net, const char* path) {
    if (!net || !path) return tz_fail(TZ_EINVAL, "tz_net_save: null argument");
    if (!net->loaded) return tz_fail(TZ_ESTATE, "tz_net_save: the network has no weights");
    int rc = weights_write_file(path, net->store);
    if (rc) return rc;
    if (net->has_hash) return tz_net_save_bitset(net, sibling(path, "bitvec.bin").c_str());   // net6_simhash.rs:152-170
    return TZ_OK;
}

int tz_net_clone(tz_net* net, int device_id, tz_net** out) {
    if (!net || !out) return tz_fail(TZ_EINVAL, "tz_net_clone: null argument");
    if (!net->loaded) return tz_fail(TZ_ESTATE, "tz_net_clone: the network has no weights");
    tz_net* other = nullptr;
    int rc = tz_net_create(net->arch == TZ_ARCH_TEST ? net->n : 0, net->arch, device_id, net->precision, net->blocks, &other);
    if (rc) return rc;
    TensorStore copy = net->store;
    if ((rc = net_apply_store(other, std::move(copy)))) {
        tz_net_destroy(other);
        return rc;
    }
    if (net->has_hash) {
        if (hipMemcpy(other->bitset, net->bitset, (size_t)1 << 29, hipMemcpyDeviceToDevice) != hipSuccess) {
            tz_net_destroy(other);
            return tz_fail(TZ_EDEVICE, "tz_net_clone: copying the SimHash set failed");
        }
    }
    *out = other;
    return TZ_OK;
}

int tz_net_get_tensor(tz_net* net, const char* name, float* out, uint64_t count, uint64_t* count_out) {
    if (!net || !name) return tz_fail(TZ_EINVAL, "tz_net_get_tensor: null argument");
    auto it = net->store.find(name);
    if (it == net->store.end()) return tz_fail(TZ_EINVAL, std::string("tz_net_get_tensor: no tensor named ") + name);
    if (count_out) *count_out = it->second.data.size();
    if (out) {
        if (count < it->second.data.size()) return tz_fail(TZ_EINVAL, "tz_net_get_tensor: buffer too small");
        memcpy(out, it->second.data.data(), 4 * it->second.data.size());
    }
    return TZ_OK;
}

int tz_net_tensor_count(tz_net* net) { return net ? (int)net->store.size() : 0; }

int tz_net_tensor_info(tz_net* net, int i, char* name_out, int name_cap, uint64_t* count_out) {
    if (!net || i < 0 || i >= (int)net->store.size()) return tz_fail(TZ_EINVAL, "tz_net_tensor_info: index out of range");
    auto it = net->store.begin();
    std::advance(it, i);
    if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", it->first.c_str());
    if (count_out) *count_out = it->second.data.size();
    return TZ_OK;
}

// Network::new(device, seed) (network/mod.rs:11; net5.rs:152-170): tch's default initialisers - conv / linear weights
// Kaiming-uniform(a = sqrt 5) = U(+-1/sqrt(fan_in)), biases U(+-1/sqrt(fan_in)), BatchNorm weight U(0,1), bias 0, running
// statistics 0 / 1, RND min 0 / max 1 (net5.rs:166-168), simhash_matrix N(0,1).  The draws come from a counter-based
// generator keyed by (seed, tensor name): tch's own stream (torch's Philox/mt19937 through manual_seed) is not reproduced.
namespace {
uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct InitRng {
    uint64_t key, ctr = 0;
    InitRng(uint64_t seed, const std::string& name) : key(mix64(seed)) {
        for (unsigned char c : name) key = mix64(key ^ c);
    }
    double uniform() { return (double)(mix64(key + (ctr++)) >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {
        const double u = std::max(uniform(), 1e-300), v = uniform();
        return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
    }
};
void init_uniform(TensorStore& st, uint64_t seed, const std::string& name, std::vector<uint32_t> dims, double lo, double hi) {
    HostTensor t;
    t.dims = dims;
    size_t n = 1;
    for (auto d : dims) n *= d;
    t.data.resize(n);
    InitRng r(seed, name);
    for (auto& x : t.data) x = (float)(lo + (hi - lo) * r.uniform());
    st[name] = std::move(t);
}
void init_const(TensorStore& st, const std::string& name, uint32_t n, float v) {
    HostTensor t;
    t.dims = {n};
    t.data.assign(n, v);
    st[name] = std::move(t);
}
void init_conv(TensorStore& st, uint64_t seed, const std::string& p, uint32_t cout, uint32_t cin, uint32_t k, bool bias) {
    const double b = 1.0 / sqrt((double)cin * k * k);
    init_uniform(st, seed, p + ".weight", {cout, cin, k, k}, -b, b);
    if (bias) init_uniform(st, seed, p + ".bias", {cout}, -b, b);
}
void init_linear(TensorStore& st, uint64_t seed, const std::string& p, uint32_t cout, uint32_t cin) {
    const double b = 1.0 / sqrt((double)cin);
    init_uniform(st, seed, p + ".weight", {cout, cin}, -b, b);
    init_uniform(st, seed, p + ".bias", {cout}, -b, b);
}
void init_bn(TensorStore& st, uint64_t seed, const std::string& p, uint32_t c) {
    init_uniform(st, seed, p + ".weight", {c}, 0.0, 1.0);
    init_const(st, p + ".bias", c, 0.f);
    init_const(st, p + ".running_mean", c, 0.f);
    init_const(st, p + ".running_var", c, 1.f);
}
}  // namespace

int tz_net_init_random(tz_net* net, uint64_t seed) {
    if (!net) return tz_fail(TZ_EINVAL, "tz_net_init_random: null argument");
    TensorStore st;
    const uint32_t cin = net->cin, nn = net->nn;
    init_conv(st, seed, "core.input_conv2d", FILTERS, cin, 3, false);
    init_bn(st, seed, "core.batch_norm", FILTERS);
    for (int b = 0; b < net->blocks; b++)
        for (const char* half : {".a", ".b"}) {
            const std::string p = "core.res_block_" + std::to_string(b) + half;
            init_conv(st, seed, p + ".conv2d", FILTERS, FILTERS, 3, false);
            init_bn(st, seed, p + ".batch_norm", FILTERS);
        }
    init_conv(st, seed, "policy.conv2d", net->pol_ch, FILTERS, 3, true);
    for (const char* head : {"value", "ube"}) {
        init_conv(st, seed, std::string(head) + ".conv2d", 1, FILTERS, 1, true);
        init_linear(st, seed, std::string(head) + ".linear", 1, nn);
    }
    if (net->has_rnd) {
        for (const char* r : {"rnd_learning", "rnd_target"}) {
            init_linear(st, seed, std::string(r) + ".input_linear", 1024, cin * nn);
            init_linear(st, seed, std::string(r) + ".hidden_linear", 1024, 1024);
            init_linear(st, seed, std::string(r) + ".final_linear", 512, 1024);
        }
        init_const(st, "min", 1, 0.f);
        init_const(st, "max", 1, 1.f);
    }
    if (net->has_hash) {
        HostTensor t;
        t.dims = {cin * nn, 32};
        t.data.resize((size_t)cin * nn * 32);
        InitRng r(seed, "simhash_matrix");
        for (auto& x : t.data) x = (float)r.normal();
        st["simhash_matrix"] = std::move(t);
    }
    return net_apply_store(net, std::move(st));
}

// Net::load on the root, then the same variables on every rank (selfplay/src/main.rs:107-121 for N shards): `status` is the
// root's load result (0 = new model loaded, anything else = nothing new / unreadable: every rank keeps its weights), so all
// ranks take the same branch.
int tz_net_broadcast(tz_net* net, tz_comm* c, int root, int status) {
    if (!net || !c) return tz_fail(TZ_EINVAL, "tz_net_broadcast: bad argument");
    int rank = 0, world = 1;
    int rc = tz_comm_info(c, &rank, &world, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (root < 0 || root >= world) return tz_fail(TZ_EINVAL, "tz_net_broadcast: bad root");
    int64_t head[2] = {status, 0};
    std::vector<unsigned char> blob;
    if (rank == root && status == 0) {
        // a root without weights says so in the header: every rank leaves with the same error, nobody is left inside the collective
        if (!net->loaded) head[0] = -1;
        else {
            tzw_dump(net->store, blob);
            head[1] = (int64_t)blob.size();
        }
    }
    rc = tz_comm_broadcast(c, head, sizeof head, root);
    if (rc) return rc;
    if (head[0] == -1) return tz_fail(TZ_ESTATE, "tz_net_broadcast: the root network has no weights");
    if (head[0] != 0) return TZ_OK;   // nothing to hand over
    blob.resize((size_t)head[1]);
    if ((rc = tz_comm_broadcast(c, blob.data(), blob.size(), root))) return rc;
    // SimHash nets: the root's load also replaced its set of seen hashes (bitvec.bin beside the model, net6_simhash.rs:173-190), so
    // the set travels with the variables — 2^32 bits in 64 MiB pieces through a host buffer.  Every rank runs every round, whatever
    // happens to its own copies: an error is reported after the last collective.
    int set_rc = TZ_OK;
    uint32_t* staging = nullptr;
    if (net->has_hash && world > 1) {
        const size_t total = (size_t)1 << 29, chunk = (size_t)1 << 26;
        std::vector<unsigned char> buf(chunk);
        if (hipSetDevice(net->device) != hipSuccess) set_rc = TZ_EDEVICE;
        if (!set_rc && rank == root && hipStreamSynchronize(net->stream) != hipSuccess) set_rc = TZ_EDEVICE;
        if (!set_rc && rank != root && hipMalloc(&staging, total) != hipSuccess) set_rc = TZ_ENOMEM;
        for (size_t done = 0; done < total; done += chunk) {
            if (rank == root && !set_rc && hipMemcpy(buf.data(), (unsigned char*)net->bitset + done, chunk, hipMemcpyDeviceToHost) != hipSuccess)
                set_rc = TZ_EDEVICE;
            if ((rc = tz_comm_broadcast(c, buf.data(), chunk, root))) break;
            if (rank != root && !set_rc && hipMemcpy((unsigned char*)staging + done, buf.data(), chunk, hipMemcpyHostToDevice) != hipSuccess)
                set_rc = TZ_EDEVICE;
        }
        if (rc || set_rc) {
            if (staging) (void)hipFree(staging);
            return rc ? rc : tz_fail(set_rc, "tz_net_broadcast: the set of seen hashes could not be handed over");
        }
    }
    if (rank == root) return TZ_OK;
    rc = tz_net_load_weights_mem(net, blob.data(), blob.size());
    if (staging) {
        if (rc) {   // the old model stays, and so does its set
            (void)hipFree(staging);
            return rc;
        }
        TZ_HIP(hipStreamSynchronize(net->stream));
        (void)hipFree(net->bitset);
        net->bitset = staging;
    }
    return rc;
}

int tz_weights_convert(const char* src, const char* dst) {
    if (!src || !dst) return tz_fail(TZ_EINVAL, "tz_weights_convert: null argument");
    TensorStore st;
    int rc = weights_read_file(src, st);
    if (rc) return rc;
    return weights_write_file(dst, st);
}

// Diagnostic: time the residual-tower conv kernel (5x5, 256->256) on `positions` boards with an
// ablation variant; returns average ms per launch over `iters` launches.
int tz_debug_conv_bench(tz_net* net, int variant, int positions, int iters, float* ms_out) {
    if (!net || !net->loaded || net->n != 5 || net->precision != TZ_PREC_BF16 || net->res.empty())
        return tz_fail(TZ_EINVAL, "tz_debug_conv_bench: needs a loaded 5x5 bf16 network");
    TZ_HIP(hipSetDevice(net->device));
    int rc = tz_net_ensure_batch(net, positions);
    if (rc) return rc;
    const ConvW& L = net->res[0];
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in = net->act_a;
    a.count_host = positions;
    a.w = L.w_mfma;
    a.bias = L.bias;
    a.out = net->act_b;
    a.cin_pad = L.cin_pad;
    a.kc_total = L.cin_pad / 32;
    a.ct_total = L.cout_pad / 16;
    a.out_stride = FILTERS;
    a.cin_real = L.cin;
    a.relu = 1;
    a.in_stride = L.cin_pad;
    a.groups = 1;
    {   // pseudo-random bf16 activations in [-1,1): zero or constant operands flatter the clock (DVFS)
        std::vector<uint16_t> h((size_t)positions * 25 * FILTERS);
        uint32_t x = 12345u;
        for (auto& v : h) {
            x = x * 1664525u + 1013904223u;
            v = f2bf(((int)(x >> 8) % 20001 - 10000) * 1e-4f);
        }
        TZ_HIP(hipMemcpy(net->act_a, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        TZ_HIP(hipMemcpy(net->act_c, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    a.residual = net->act_c;
    a.has_res = 1;
    hipEvent_t e0, e1;
    TZ_HIP(hipEventCreate(&e0));
    TZ_HIP(hipEventCreate(&e1));
    // variant = cfg * 10 + ablation (0 shipped, 1 no LDS reads, 2 no weight loads, 3 neither, 4 one tap only)
    auto run = [&](int n) -> int {
        int rc = TZ_OK;
        for (int i = 0; i < n && !rc; i++) {
            switch (variant) {
                case 0: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 0>(a, positions, 1, net->stream); break;
#ifdef TZ_ABLATIONS   // diagnostic builds only (python -m takzero_amd.build --ablations): the A/B twins stay out of the shipped object
                case 1: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 1>(a, positions, 1, net->stream); break;
                case 2: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 2>(a, positions, 1, net->stream); break;
                case 3: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 3>(a, positions, 1, net->stream); break;
                case 4: rc = launch_conv<5, ppt_for(5), 8, 2, 1, false, true, 0>(a, positions, 1, net->stream); break;
                case 20: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 0, 0>(a, positions, 1, net->stream); break;
                case 21: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 1, 0>(a, positions, 1, net->stream); break;
                case 23: rc = launch_conv<5, ppt_for(5), 8, 2, 9, false, true, 3, 0>(a, positions, 1, net->stream); break;
                case 30: rc = launch_conv<5, ppt_for(5), 4, 4, 9, false, true, 0, 1, __bf16, 1>(a, positions, 1, net->stream); break;
                case 31: rc = launch_conv<5, ppt_for(5), 4, 4, 9, false, true, 1, 1, __bf16, 1>(a, positions, 1, net->stream); break;
                case 33: rc = launch_conv<5, ppt_for(5), 4, 4, 9, false, true, 3, 1, __bf16, 1>(a, positions, 1, net->stream); break;
                case 10: rc = launch_conv<5, ppt_small(5), 4, 4, 9, false, true, 0>(a, positions, 1, net->stream); break;
                case 11: rc = launch_conv<5, ppt_small(5), 4, 4, 9, false, true, 1>(a, positions, 1, net->stream); break;
                case 12: rc = launch_conv<5, ppt_small(5), 4, 4, 9, false, true, 2>(a, positions, 1, net->stream); break;
                case 13: rc = launch_conv<5, ppt_small(5), 4, 4, 9, false, true, 3>(a, positions, 1, net->stream); break;
                case 14: rc = launch_conv<5, ppt_small(5), 4, 4, 1, false, true, 0>(a, positions, 1, net->stream); break;
#endif
                default: rc = tz_fail(TZ_EINVAL, "tz_debug_conv_bench: unknown variant (the ablation variants need a build with --ablations)");
            }
        }
        return rc;
    };
    if ((rc = run(3))) return rc;
    TZ_HIP(hipEventRecord(e0, net->stream));
    if ((rc = run(iters))) return rc;
    TZ_HIP(hipEventRecord(e1, net->stream));
    TZ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / iters;
    return TZ_OK;
}

// Diagnostic: average ms per launch of the fused 5x5 tower (the net's own weights) on `positions` boards of
// pseudo-random activations; variant = OPT bitmask of tower_mfma_kernel.  For in-process A/B.
int tz_debug_tower_bench(tz_net* net, int variant, int positions, int iters, float* ms_out) {
    if (!net || !net->loaded || net->n != 5 || net->precision != TZ_PREC_BF16 || !net->tower_w)
        return tz_fail(TZ_EINVAL, "tz_debug_tower_bench: needs a loaded 5x5 bf16 network");
    TZ_HIP(hipSetDevice(net->device));
    int rc = tz_net_ensure_batch(net, positions);
    if (rc) return rc;
    {
        std::vector<uint16_t> h((size_t)positions * 25 * FILTERS);
        uint32_t x = 999u;
        for (auto& v : h) {
            x = x * 1664525u + 1013904223u;
            v = (x >> 31) ? 0 : f2bf((float)((x >> 8) % 10001) * 1e-4f);  // post-ReLU-like: half zeros
        }
        TZ_HIP(hipMemcpy(net->act_a, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    TowerArgs a;
    a.in = reinterpret_cast<const uint16_t*>(net->act_a);
    a.out = reinterpret_cast<uint16_t*>(net->act_b);
    a.w = net->tower_w;
    a.bias = net->tower_bias;
    a.count_dev = nullptr;
    a.count_host = positions;
    a.nlayers = 2 * net->blocks;
    hipEvent_t e0, e1;
    TZ_HIP(hipEventCreate(&e0));
    TZ_HIP(hipEventCreate(&e1));
    auto run = [&](int n) -> int {
        int r = TZ_OK;
        for (int i = 0; i < n && !r; i++) {
            switch (variant) {
                case 4224: r = launch_tower<5, 4224>(a, positions, net->stream); break;  // 4096 + 128 = the shipped loop (TZ_TOWER=1)
#ifdef TZ_ABLATIONS   // diagnostic builds only (python -m takzero_amd.build --ablations)
                case 0: r = launch_tower<5, 0>(a, positions, net->stream); break;
                case 1: r = launch_tower<5, 1>(a, positions, net->stream); break;
                case 2: r = launch_tower<5, 2>(a, positions, net->stream); break;
                case 4: r = launch_tower<5, 4>(a, positions, net->stream); break;
                case 8: r = launch_tower<5, 8>(a, positions, net->stream); break;
                case 16: r = launch_tower<5, 16>(a, positions, net->stream); break;
                case 32: r = launch_tower<5, 32>(a, positions, net->stream); break;
                case 48: r = launch_tower<5, 48>(a, positions, net->stream); break;
                case 64: r = launch_tower<5, 64>(a, positions, net->stream); break;
                case 112: r = launch_tower<5, 112>(a, positions, net->stream); break;
                case 128: r = launch_tower<5, 128>(a, positions, net->stream); break;
                case 256: r = launch_tower<5, 256>(a, positions, net->stream); break;
                case 512: r = launch_tower<5, 512>(a, positions, net->stream); break;
                case 1024: r = launch_tower<5, 1024>(a, positions, net->stream); break;
                case 2048: r = launch_tower<5, 2048>(a, positions, net->stream); break;
                case 4240: r = launch_tower<5, 4240>(a, positions, net->stream); break;  // ... without activation reads
                case 4256: r = launch_tower<5, 4256>(a, positions, net->stream); break;  // ... without the weight stream
                case 4272: r = launch_tower<5, 4272>(a, positions, net->stream); break;  // ... without both
                case 4336: r = launch_tower<5, 4336>(a, positions, net->stream); break;  // ... MFMAs, barriers only
                case 4288: r = launch_tower<5, 4288>(a, positions, net->stream); break;  // shipped loop without the layer epilogue
                case 12416: r = launch_tower<5, 12416>(a, positions, net->stream); break;  // shipped loop without the mid-tap rebase adds
                case 100000: r = launch_tower4x2<5>(a, positions, net->stream); break;     // 4 channel groups x 2 row halves
                case 4225: r = launch_tower<5, 4225>(a, positions, net->stream); break;     // shipped loop + cross-layer weight prefetch
                case 4226: r = launch_tower<5, 4226>(a, positions, net->stream); break;     // ... + wave stagger
                case 4232: r = launch_tower<5, 4232>(a, positions, net->stream); break;     // ... + prefetch distance 3
                case 20608: r = launch_tower<5, 20608>(a, positions, net->stream); break;   // shipped loop, weight loads sc0
                case 36992: r = launch_tower<5, 36992>(a, positions, net->stream); break;   // ... sc1
                case 69760: r = launch_tower<5, 69760>(a, positions, net->stream); break;   // ... nt
#endif
                default: r = tz_fail(TZ_EINVAL, "tz_debug_tower_bench: unknown variant (the ablation variants need a build with --ablations)");
            }
        }
        return r;
    };
    if ((rc = run(2))) return rc;
    TZ_HIP(hipEventRecord(e0, net->stream));
    if ((rc = run(iters))) return rc;
    TZ_HIP(hipEventRecord(e1, net->stream));
    TZ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / iters;
    if (getenv("TZ_TOWER_BENCH_CHECKSUM")) {  // A/B variants must agree: print a checksum of the output of the last launch
        std::vector<uint16_t> h((size_t)positions * 25 * FILTERS);
        TZ_HIP(hipMemcpy(h.data(), net->act_b, h.size() * 2, hipMemcpyDeviceToHost));
        unsigned long long sum = 0, x = 1469598103934665603ull;
        for (uint16_t v : h) {
            sum += v;
            x = (x ^ v) * 1099511628211ull;
        }
        fprintf(stderr, "tower variant %d: output sum %llu hash %016llx\n", variant, sum, x);
    }
    return TZ_OK;
}

// Diagnostic builds: the in-kernel shader clock of the last stamped launch (TZ_NET_ABL=8): median over workgroups of
// delta s_memtime / delta s_memrealtime x 100 MHz, and the median tower duration in microseconds.
int tz_debug_net_clock(tz_net* net, double* mhz_out, double* tower_us_out) {
    if (!net || !net->dbg_buf || net->dbg_groups <= 0) return tz_fail(TZ_ESTATE, "tz_debug_net_clock: no stamped launch (build --ablations, TZ_NET_ABL=8)");
    TZ_HIP(hipSetDevice(net->device));
    TZ_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)net->dbg_groups * 4);
    TZ_HIP(hipMemcpy(h.data(), net->dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz, us;
    for (int g = 0; g < net->dbg_groups; g++) {
        const double dc = (double)(h[g * 4 + 2] - h[g * 4 + 0]), dr = (double)(h[g * 4 + 3] - h[g * 4 + 1]);
        if (dr > 0) {
            mhz.push_back(dc / dr * 100.0);
            us.push_back(dr / 100.0);
        }
    }
    if (mhz.empty()) return tz_fail(TZ_ESTATE, "tz_debug_net_clock: empty stamps");
    std::sort(mhz.begin(), mhz.end());
    std::sort(us.begin(), us.end());
    if (mhz_out) *mhz_out = mhz[mhz.size() / 2];
    if (tower_us_out) *tower_us_out = us[us.size() / 2];
    return TZ_OK;
}

// diagnostic builds: the raw stamps of the last stamped launch, [workgroup][4]
int tz_debug_net_stamps(tz_net* net, unsigned long long* out, int max_groups, int* groups_out) {
    if (!net || !net->dbg_buf || net->dbg_groups <= 0) return tz_fail(TZ_ESTATE, "tz_debug_net_stamps: no stamped launch (build --ablations, TZ_NET_ABL=8 or 16)");
    TZ_HIP(hipSetDevice(net->device));
    TZ_HIP(hipDeviceSynchronize());
    const int g = std::min(max_groups, net->dbg_groups);
    TZ_HIP(hipMemcpy(out, net->dbg_buf, (size_t)g * 4 * 8, hipMemcpyDeviceToHost));
    if (groups_out) *groups_out = g;
    return TZ_OK;
}

int tz_net_destroy(tz_net* net) {
    if (!net) return TZ_OK;
    (void)hipSetDevice(net->device);
    if (net->stream) (void)hipStreamSynchronize(net->stream);
    NetWeights old;
    old.conv_in = net->conv_in;
    old.policy = net->policy;
    old.res = net->res;
    old.tower_w = net->tower_w;
    old.tower_w_lo = net->tower_w_lo;
    old.tower_w8 = net->tower_w8;
    old.c8_scales = net->c8_scales;
    old.tower_bias = net->tower_bias;
    old.heads = net->heads;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) old.rnd[a][b] = net->rnd[a][b];
    for (int l = 0; l < 3; l++) {
        old.rndw[l] = net->rndw[l];
        old.rndb[l] = net->rndb[l];
    }
    old.simhash = net->simhash;
    free_weights(old);
    void* bufs[] = {net->act_a, net->act_b, net->act_c, net->planes, net->policy_out, net->value, net->ube,
                    net->variance, net->aux, net->rnd_in, net->rnd_h1, net->rnd_h2, net->rnd_out, net->bitset, net->seeds, net->xch, net->xch_count};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (net->stream_rnd) (void)hipStreamDestroy(net->stream_rnd);
    if (net->ev_in) (void)hipEventDestroy(net->ev_in);
    if (net->ev_rnd) (void)hipEventDestroy(net->ev_rnd);
    for (auto& ev : net->conv_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (net->eval_dev) (void)hipFree(net->eval_dev);
    if (net->eval_host) (void)hipHostFree(net->eval_host);
    if (net->stream) (void)hipStreamDestroy(net->stream);
    delete net;
    return TZ_OK;
}

static int net_upload_states(tz_net* net, int batch, const tz_state* states, tz_state** dev) {
    TZ_HIP(hipSetDevice(net->device));
    int rc = tz_net_ensure_batch(net, batch);
    if (rc) return rc;
    TZ_HIP(hipMalloc(dev, (size_t)batch * sizeof(tz_state)));
    TZ_HIP(hipMemcpyAsync(*dev, states, (size_t)batch * sizeof(tz_state), hipMemcpyHostToDevice, net->stream));
    return TZ_OK;
}

int tz_net_eval(tz_net* net, int batch, const tz_state* states, const uint16_t* legal_idx, const int32_t* legal_count,
                int amax, float* logits_out, float* value_out, float* variance_out) {
    if (!net || !states || !legal_idx || !legal_count || !logits_out || !value_out || !variance_out)
        return tz_fail(TZ_EINVAL, "tz_net_eval: null argument");
    if (batch <= 0 || amax <= 0) return tz_fail(TZ_EINVAL, "tz_net_eval: empty batch (net5.rs:226-227 asserts)");
    for (int b = 0; b < batch; b++) {
        if (states[b].n != net->n) return tz_fail(TZ_EINVAL, "tz_net_eval: state board size does not match the network");
        if (legal_count[b] < 0 || legal_count[b] > amax) return tz_fail(TZ_EINVAL, "tz_net_eval: legal_count out of range");
        for (int j = 0; j < legal_count[b]; j++)
            if (legal_idx[(size_t)b * amax + j] >= net->pol_ch * net->nn)
                return tz_fail(TZ_EINVAL, "tz_net_eval: move index out of range");
    }
    // one staging buffer on each side: [states | legal_idx | legal_count] in, [logits | value | variance] out
    TZ_HIP(hipSetDevice(net->device));
    int rc = tz_net_ensure_batch(net, batch);
    if (rc) return rc;
    const size_t cells = (size_t)batch * amax;
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_states = 0, o_legal = up16(o_states + (size_t)batch * sizeof(tz_state)), o_cnt = up16(o_legal + cells * 2),
                 in_bytes = up16(o_cnt + (size_t)batch * 4);
    const size_t o_log = in_bytes, o_val = up16(o_log + cells * 4), o_var = up16(o_val + (size_t)batch * 4),
                 total = up16(o_var + (size_t)batch * 4);
    if (total > net->eval_bytes) {
        TZ_HIP(hipStreamSynchronize(net->stream));
        if (net->eval_dev) (void)hipFree(net->eval_dev);
        if (net->eval_host) (void)hipHostFree(net->eval_host);
        net->eval_dev = net->eval_host = nullptr;
        net->eval_bytes = 0;
        const size_t want = total + total / 2;
        if (hipMalloc(&net->eval_dev, want) != hipSuccess || hipHostMalloc(&net->eval_host, want, hipHostMallocDefault) != hipSuccess) {
            if (net->eval_dev) (void)hipFree(net->eval_dev);
            net->eval_dev = nullptr;
            return tz_fail(TZ_ENOMEM, "tz_net_eval: staging allocation failed");
        }
        net->eval_bytes = want;
    }
    char* host = static_cast<char*>(net->eval_host);
    char* dev = static_cast<char*>(net->eval_dev);
    memcpy(host + o_states, states, (size_t)batch * sizeof(tz_state));
    memcpy(host + o_legal, legal_idx, cells * 2);
    memcpy(host + o_cnt, legal_count, (size_t)batch * 4);
    hipStream_t st = net->stream;
    TZ_HIP(hipMemcpyAsync(dev, host, in_bytes, hipMemcpyHostToDevice, st));
    NetOut o;
    // several CUs per board group for batches up to 128 on 5x5 in the 16-bit storage types (TZ_NET_SPLIT=0: one CU per group, A/B)
    static const bool split_off = getenv("TZ_NET_SPLIT") && !strcmp(getenv("TZ_NET_SPLIT"), "0");
    if (!split_off && net->n == 5 && ((batch <= NET_SPLIT_MAX_POSITIONS && (net->precision == TZ_PREC_F16 || net->precision == TZ_PREC_BF16)) ||
                                      (batch <= NET_SPLIT_MAX_POSITIONS && net->precision == TZ_PREC_F16X2)) &&
        net->blocks > 0 && net_fused_mode() == 2) {
        if (!net->stream_rnd && !net->ev_in) {
            if (hipStreamCreateWithFlags(&net->stream_rnd, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&net->ev_in, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&net->ev_rnd, hipEventDisableTiming) != hipSuccess)
                net->stream_rnd = nullptr;   // without it the RND MLP follows the net kernel on its stream, as in every other path
        }
        net->eval_split = true;
    }
    rc = tz_net_forward_device(net, reinterpret_cast<const tz_state*>(dev + o_states), nullptr, nullptr, batch, batch, st, &o);
    net->eval_split = false;
    if (rc) return rc;
    eval_pack_kernel<<<(int)((cells + 255) / 256), 256, 0, st>>>(o.policy, net->nn, o.policy_stride, reinterpret_cast<const uint16_t*>(dev + o_legal),
                                                               reinterpret_cast<const int32_t*>(dev + o_cnt), amax, batch, o.value, o.variance,
                                                               reinterpret_cast<float*>(dev + o_log), reinterpret_cast<float*>(dev + o_val),
                                                               reinterpret_cast<float*>(dev + o_var));
    TZ_HIP(hipGetLastError());
    TZ_HIP(hipMemcpyAsync(host + o_log, dev + o_log, total - o_log, hipMemcpyDeviceToHost, st));
    TZ_HIP(hipStreamSynchronize(st));
    memcpy(logits_out, host + o_log, cells * 4);
    memcpy(value_out, host + o_val, (size_t)batch * 4);
    memcpy(variance_out, host + o_var, (size_t)batch * 4);
    for (size_t i = 0; i < cells; i++)
        if (logits_out[i] != logits_out[i]) return tz_fail(TZ_ENUMERIC, "tz_net_eval: NaN logit (net5.rs:263 panics)");
    return TZ_OK;
}

int tz_net_encode(tz_net* net, int batch, const tz_state* states, float* planes_out) {
    if (!net || !states || !planes_out || batch <= 0) return tz_fail(TZ_EINVAL, "tz_net_encode: bad argument");
    tz_state* dstates = nullptr;
    int rc = net_upload_states(net, batch, states, &dstates);
    if (rc) return rc;
    hipStream_t st = net->stream;
    rc = encode(net, dstates, nullptr, nullptr, batch, batch, st);
    float* dout = nullptr;
    const size_t total = (size_t)batch * net->cin * net->nn;
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMalloc(&dout, total * 4);
        if (e == hipSuccess) {
            planes_nchw_kernel<<<(int)((total + 255) / 256), 256, 0, st>>>(net->planes, net->nn, net->cin, batch, dout);
            e = hipMemcpyAsync(planes_out, dout, total * 4, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(dstates);
    if (dout) (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("tz_net_encode: ") + hipGetErrorString(e));
    return TZ_OK;
}

// HashNetwork::{get_indices, update_counts} and the bitvec.bin half of Network::load for the SimHash nets
// (net6_simhash.rs:152-190,202-243).  indices_out may be NULL; update != 0 also marks the positions as seen.
int tz_net_hash_indices(tz_net* net, int batch, const tz_state* states, uint32_t* indices_out, int update) {
    if (!net || !states || batch <= 0) return tz_fail(TZ_EINVAL, "tz_net_hash_indices: bad argument");
    if (!net->has_hash || !net->loaded) return tz_fail(TZ_ESTATE, "tz_net_hash_indices: not a loaded SimHash network");
    tz_state* dstates = nullptr;
    int rc = net_upload_states(net, batch, states, &dstates);
    if (rc) return rc;
    hipStream_t st = net->stream;
    uint32_t* didx = nullptr;
    hipError_t e = hipMalloc(&didx, batch * 4);
    if (e == hipSuccess) {
        rc = tz_net_ensure_batch(net, batch);
        if (!rc) rc = launch_simhash_state(net->n, dstates, nullptr, net->simhash, net->bitset, nullptr, batch, batch, net->aux, didx, st);
        if (!rc) {
            if (update) bitset_set_kernel<<<(batch + 255) / 256, 256, 0, st>>>(net->bitset, didx, batch);
            if (indices_out) e = hipMemcpyAsync(indices_out, didx, batch * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
    }
    (void)hipFree(dstates);
    if (didx) (void)hipFree(didx);
    if (rc) return rc;
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("tz_net_hash_indices: ") + hipGetErrorString(e));
    return TZ_OK;
}

// bitvec.bin: the raw words of the reference's BitBox<usize, Lsb0> (2^32 bits = 512 MiB), net6_simhash.rs:152-190
int tz_net_load_bitset(tz_net* net, const char* path) {
    if (!net || !path) return tz_fail(TZ_EINVAL, "tz_net_load_bitset: null argument");
    if (!net->has_hash) return tz_fail(TZ_ESTATE, "tz_net_load_bitset: not a SimHash network");
    TZ_HIP(hipSetDevice(net->device));
    FILE* f = fopen(path, "rb");
    if (!f) return tz_fail(TZ_EPARSE, std::string("tz_net_load_bitset: cannot open ") + path);
    const size_t total = (size_t)1 << 29, chunk = (size_t)1 << 26;
    std::vector<unsigned char> buf(chunk);
    uint32_t* staging = nullptr;   // a failed load leaves the old set active
    if (hipMalloc(&staging, total) != hipSuccess) {
        fclose(f);
        return tz_fail(TZ_ENOMEM, "tz_net_load_bitset: cannot allocate the staging set");
    }
    size_t done = 0;
    while (done < total) {
        const size_t rd = fread(buf.data(), 1, chunk, f);
        if (rd != chunk || hipMemcpy((unsigned char*)staging + done, buf.data(), chunk, hipMemcpyHostToDevice) != hipSuccess) break;
        done += chunk;
    }
    fclose(f);
    if (done != total) {
        (void)hipFree(staging);
        return tz_fail(TZ_EPARSE, "tz_net_load_bitset: file is not 2^32 bits");
    }
    TZ_HIP(hipStreamSynchronize(net->stream));
    (void)hipFree(net->bitset);
    net->bitset = staging;
    return TZ_OK;
}

int tz_net_save_bitset(tz_net* net, const char* path) {
    if (!net || !path) return tz_fail(TZ_EINVAL, "tz_net_save_bitset: null argument");
    if (!net->has_hash) return tz_fail(TZ_ESTATE, "tz_net_save_bitset: not a SimHash network");
    TZ_HIP(hipSetDevice(net->device));
    TZ_HIP(hipStreamSynchronize(net->stream));
    FILE* f = fopen(path, "wb");
    if (!f) return tz_fail(TZ_EPARSE, std::string("tz_net_save_bitset: cannot open ") + path);
    const size_t total = (size_t)1 << 29, chunk = (size_t)1 << 26;
    std::vector<unsigned char> buf(chunk);
    for (size_t done = 0; done < total; done += chunk) {
        if (hipMemcpy(buf.data(), (unsigned char*)net->bitset + done, chunk, hipMemcpyDeviceToHost) != hipSuccess ||
            fwrite(buf.data(), 1, chunk, f) != chunk) {
            fclose(f);
            return tz_fail(TZ_EDEVICE, "tz_net_save_bitset: write failed");
        }
    }
    fclose(f);
    return TZ_OK;
}

int tz_net_forward_raw(tz_net* net, int batch, const tz_state* states, float* policy_out, float* value_out, float* ube_out) {
    if (!net || !states || batch <= 0) return tz_fail(TZ_EINVAL, "tz_net_forward_raw: bad argument");
    tz_state* dstates = nullptr;
    int rc = net_upload_states(net, batch, states, &dstates);
    if (rc) return rc;
    hipStream_t st = net->stream;
    NetOut o;
    rc = tz_net_forward_device(net, dstates, nullptr, nullptr, batch, batch, st, &o);
    float* dout = nullptr;
    hipError_t e = hipSuccess;
    const size_t total = (size_t)batch * net->pol_ch * net->nn;
    if (!rc) {
        if (policy_out) {
            e = hipMalloc(&dout, total * 4);
            if (e == hipSuccess) {
                policy_nchw_kernel<<<(int)((total + 255) / 256), 256, 0, st>>>(o.policy, net->nn, o.policy_stride, net->pol_ch, batch, dout);
                e = hipMemcpyAsync(policy_out, dout, total * 4, hipMemcpyDeviceToHost, st);
            }
        }
        if (e == hipSuccess && value_out) e = hipMemcpyAsync(value_out, net->value, batch * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && ube_out) e = hipMemcpyAsync(ube_out, net->ube, batch * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(dstates);
    if (dout) (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("tz_net_forward_raw: ") + hipGetErrorString(e));
    return TZ_OK;
}

}  // extern "C"
#endif  // TZ_NN_SPLIT_TU
