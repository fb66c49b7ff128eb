// tz_learn.hip — the `learn` training step on MI355X: forward in training mode, the
// three losses, backward and Adam (learn/src/main.rs:376-423; graph net5.rs:44-191, residual.rs:13-63).
//
// The reference trains in fp32 through LibTorch; so does this: every contraction is an fp32 MFMA GEMM
// (v_mfma_f32_32x32x2_f32) whose A operand is the im2col view of an NHWC tensor, gathered by the tile loader
// (no col matrix exists) — at the reference's batch of 128 positions a step is 0.46 TFLOP:
//
//   activations   NHWC fp32  [M = batch*n*n][256]            (everything of one step stays resident)
//   conv weights  GEMM layout [K = 9*cin (padded to 64)][cout (padded to 64)], k = tap*cin + ci
//   forward       c = im2col(x) x W;  BatchNorm with batch statistics;  ReLU (+ skip)
//   backward      dW = im2col(x)^T x dc  (A-transposed GEMM);  dx = im2col(dc) x W'  with W' = taps mirrored, ci/co swapped
//   parameters, gradients and the two Adam moments live in four parallel arenas; one Adam launch per step.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <map>
#include <string>
#include <vector>

#include "tz_nn.h"

namespace {

constexpr int FILTERS = 256;
constexpr float BN_EPS = 1e-5f;       // tch BatchNormConfig::default
constexpr float BN_MOMENTUM = 0.1f;   // tch BatchNormConfig::default
constexpr int SPLITS = 64;            // row splits of the per-channel reductions (8 x 64 workgroups)
constexpr float MINIMUM_UBE_TARGET = -10.0f;  // learn/src/main.rs:47
constexpr float MAXIMUM_VARIANCE = 4.0f;      // net5.rs:23

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// C[M][N] = (accumulate ? C : 0) + A x B (+ bias[n]).  M, N multiples of 64, K multiple of 32.
// AT = false: A(m,k) = A[m*lda + k];  AT = true: A(m,k) = A[k*lda + m].   B(k,n) = B[k*ldb + n].
// One 64x64 tile per workgroup of 8 waves = two groups of 4: group g multiplies the k-slabs 2i+g (16 deep), each
// of its waves owning a 32x32 quadrant, so every SIMD has two waves to interleave; the two partial tiles are added
// through LDS at the end (fixed order: deterministic).  Slabs are double buffered in LDS (one barrier per step) and
// the next pair is fetched into registers while the MFMAs of the current one run.
//
// GATHER: A is not a stored matrix but the im2col view of an NHWC tensor x [pixels][gc]: A(pixel, tap*gc + c) =
// x[(board, y + tap/3 - 1, x + tap%3 - 1)][c], zero off the board and for k >= 9*gc (the K padding).  The loader
// computes that address itself (4 consecutive channels per thread), so no col matrix is ever written or read.
template <bool AT, bool GATHER>
__global__ __launch_bounds__(512) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, const float* __restrict__ bias, int K,
                                                       int lda, int ldb, int ldc, int accumulate, int gn, int gc) {
    __shared__ float As[2][2][16][68];  // [buffer][group][k][m]
    __shared__ float Bs[2][2][16][68];
    const int g = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const size_t m0 = (size_t)blockIdx.x * 64, n0 = (size_t)blockIdx.y * 64;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int ak = AT ? (t >> 4) : (t & 3) * 4, am = AT ? (t & 15) * 4 : (t >> 2);
    const int bk = t >> 4, bn = (t & 15) * 4;
    const float* ap = AT ? A + (size_t)(16 * g + ak) * lda + m0 + am : A + (m0 + am) * lda + 16 * g + ak;
    const float* bp = B + (size_t)(16 * g + bk) * ldb + n0 + bn;
    const size_t astep = AT ? (size_t)32 * lda : 32, bstep = (size_t)32 * ldb;
    // gather state.  The address arithmetic of the im2col view is taken out of the loop (integer VALU work inside an MFMA
    // loop is expensive: it competes with the other wave of the SIMD for issue slots): a small LDS table holds, for every
    // (tap, pixel of a board), the offset of the source pixel or -1 off the board; per step a thread advances one counter,
    // reads one table entry and adds.
    //   !AT: the thread's pixel is fixed (row m0+am), its k = tap*gc + c advances by 32 per step.
    //   AT : its 4 k-indices are fixed (m0+am..+3 = one tap, 4 channels), its pixel advances by 32 per step.
    __shared__ int srcoff[9][40];   // [tap][pixel in board] -> (source pixel in board) * gc, or -1
    const int gnn = gn * gn;
    int g_b = 0, g_lp = 0, g_tap = 0, g_c = 0;          // board, pixel inside the board, tap, channel
    if (GATHER) {
        for (int i = threadIdx.x; i < 9 * gnn; i += 512) {
            const int tap = i / gnn, lp = i - tap * gnn, y = lp / gn + tap / 3 - 1, x = lp % gn + tap % 3 - 1;
            srcoff[tap][lp] = (y >= 0 && y < gn && x >= 0 && x < gn) ? (y * gn + x) * gc : -1;
        }
        const int pix = AT ? 16 * g + ak : (int)m0 + am, k = AT ? (int)m0 + am : 16 * g + ak;
        g_b = pix / gnn;
        g_lp = pix - g_b * gnn;
        g_tap = k / gc;            // >= 9 marks the K padding
        g_c = k - g_tap * gc;
        __syncthreads();
    }
    const float* g_row = A + (size_t)g_b * gnn * gc;    // start of the thread's current board
    auto gather = [&]() -> float4 {
        const int off = g_tap < 9 ? srcoff[g_tap][g_lp] : -1;
        if (off < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *(const float4*)(g_row + off + g_c);
    };
    auto gather_advance = [&]() {
        if (AT) {
            g_lp += 32;
            while (g_lp >= gnn) {
                g_lp -= gnn;
                g_row += gnn * gc;
            }
        } else {
            g_c += 32;
            while (g_c >= gc) {
                g_c -= gc;
                g_tap++;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    // operands of k-steps i and i + 1 are in flight while step i - 1 multiplies: one step (a thousand cycles) does not cover a
    // miss of the XCD's L2 on the gathered view, two do
    float4 ra0, ra1, rb0, rb1;   // named, not an array: a slot picked at run time would send them to scratch
    auto fetch = [&](float4& ra, float4& rb) {
        ra = GATHER ? gather() : *(const float4*)ap;
        rb = *(const float4*)bp;
    };
    auto advance = [&]() {
        bp += bstep;
        if (GATHER) gather_advance();
        else ap += astep;
    };
    fetch(ra0, rb0);
    if (32 < K) {
        advance();
        fetch(ra1, rb1);
    }
    auto step = [&](int k0, int buf, float4& ra, float4& rb) {
        if (AT) {
            *(float4*)&As[buf][g][ak][am] = ra;
        } else {
            As[buf][g][ak + 0][am] = ra.x;
            As[buf][g][ak + 1][am] = ra.y;
            As[buf][g][ak + 2][am] = ra.z;
            As[buf][g][ak + 3][am] = ra.w;
        }
        *(float4*)&Bs[buf][g][bk][bn] = rb;
        __syncthreads();
        if (k0 + 64 < K) {
            advance();
            fetch(ra, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float a = As[buf][g][kk + (lane >> 5)][wm + (lane & 31)];
            const float b = Bs[buf][g][kk + (lane >> 5)][wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    };
    for (int k0 = 0; k0 < K; k0 += 64) {
        step(k0, 0, ra0, rb0);
        if (k0 + 32 < K) step(k0 + 32, 1, ra1, rb1);
    }
    __syncthreads();
    float* red = &As[0][0][0][0];  // 64 x 64 partial tile of group 1
    if (g == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[(wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 64 + wn + (lane & 31)] = acc[r];
    }
    __syncthreads();
    if (g == 1) return;
    const size_t n = n0 + wn + (lane & 31);
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ml = wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        const size_t m = m0 + ml;
        float v = acc[r] + red[ml * 64 + wn + (lane & 31)] + bv;
        if (accumulate) v += C[m * ldc + n];
        C[m * ldc + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same product, its work cut so that every CU gets an equal share ("stream-K").  At the reference's batch of 128 a forward or
// data-gradient GEMM is 200 tiles of 64x64 and a weight-gradient GEMM 144 — for 256 CUs.  Here the (tile, k-slab) pairs are laid
// out in one line, tile by tile, and workgroup w of G takes the contiguous range [w T / G, (w + 1) T / G): it walks through at most
// one partial tile, whole tiles, one partial tile.  A whole tile gets the usual epilogue; a partial one goes to the workgroup's
// slot in a workspace, and gemm_fixup_kernel adds a tile's parts in ascending k (ascending workgroup) — a fixed order, so the
// result does not depend on timing.  A workgroup is 4 waves, one 32x32 quadrant each over the full slab depth of 32: two of them
// share a CU (G = 2 x CUs), each with its own barriers, so one's barrier wait is the other's MFMA time.
struct SkRange {
    long long begin, end;
};
__device__ __host__ inline SkRange sk_range(int w, int G, long long total) {
    return SkRange{(long long)w * total / G, (long long)(w + 1) * total / G};
}

template <bool AT, bool GATHER>
__global__ __launch_bounds__(256) void gemm_sk_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                      const float* __restrict__ bias, int K, int lda, int ldb, int ldc, int accumulate,
                                                      int gn, int gc, int tiles_m, long long total, float* __restrict__ ws) {
    __shared__ float As[2][32][68];  // [buffer][k][m]
    __shared__ float Bs[2][32][68];
    __shared__ int srcoff[9][40];    // [tap][pixel in board] -> (source pixel in board) * gc, or -1 (see gemm_f32_kernel)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int S = K / 32, gnn = gn * gn;
    const SkRange mine = sk_range(blockIdx.x, gridDim.x, total);
    if (GATHER) {
        for (int i = t; i < 9 * gnn; i += 256) {
            const int tap = i / gnn, lp = i - tap * gnn, y = lp / gn + tap / 3 - 1, x = lp % gn + tap % 3 - 1;
            srcoff[tap][lp] = (y >= 0 && y < gn && x >= 0 && x < gn) ? (y * gn + x) * gc : -1;
        }
        __syncthreads();
    }
    // loader: two float4 per thread and operand per slab.  !AT: 4 consecutive k of rows am, am + 32;  AT: 4 consecutive m of k-rows ak, ak + 16
    const int ak = AT ? (t >> 4) : (t & 7) * 4, am = AT ? (t & 15) * 4 : (t >> 3);
    const int bk = t >> 4, bn = (t & 15) * 4;
    for (long long cur = mine.begin; cur < mine.end;) {
        const int tile = (int)(cur / S), ks0 = (int)(cur - (long long)tile * S);
        const int ks1 = (int)((long long)S < ks0 + (mine.end - cur) ? (long long)S : ks0 + (mine.end - cur));
        const size_t m0 = (size_t)(tile % tiles_m) * 64, n0 = (size_t)(tile / tiles_m) * 64;
        // state of the thread's two fetches per slab (h = 0, 1): plain pointer, or board / pixel inside the board / tap / channel of the
        // gathered view.  Named objects throughout: an array indexed by h would live in scratch.
        struct Src {
            const float* ap;
            const float* row;
            int lp, tap, c;
        };
        auto src_init = [&](int h) -> Src {
            Src q{A, A, 0, 0, 0};
            if (GATHER) {
                const int pix = AT ? 32 * ks0 + ak + 16 * h : (int)m0 + am + 32 * h, k = AT ? (int)m0 + am : 32 * ks0 + ak;
                const int b = pix / gnn;
                q.lp = pix - b * gnn;
                q.row = A + (size_t)b * gnn * gc;
                q.tap = k / gc;            // >= 9 marks the K padding
                q.c = k - q.tap * gc;
            } else {
                q.ap = AT ? A + (size_t)(32 * ks0 + ak + 16 * h) * lda + m0 + am : A + (m0 + am + 32 * h) * lda + 32 * ks0 + ak;
            }
            return q;
        };
        Src s0 = src_init(0), s1 = src_init(1);
        const float* bp = B + (size_t)(32 * ks0 + bk) * ldb + n0 + bn;
        const size_t astep = AT ? (size_t)32 * lda : 32, bstep = (size_t)32 * ldb, bhalf = (size_t)16 * ldb;
        auto load_a = [&](const Src& q) -> float4 {
            if (GATHER) {
                const int off = q.tap < 9 ? srcoff[q.tap][q.lp] : -1;
                return off < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(q.row + off + q.c);
            }
            return *(const float4*)q.ap;
        };
        auto advance_a = [&](Src& q) {
            if (!GATHER) {
                q.ap += astep;
            } else if (AT) {
                q.lp += 32;
                while (q.lp >= gnn) {
                    q.lp -= gnn;
                    q.row += gnn * gc;
                }
            } else {
                q.c += 32;
                while (q.c >= gc) {
                    q.c -= gc;
                    q.tap++;
                }
            }
        };
        struct Slab {
            float4 a0, a1, b0, b1;
        };
        auto fetch = [&](Slab& f) {
            f.a0 = load_a(s0);
            f.a1 = load_a(s1);
            f.b0 = *(const float4*)bp;
            f.b1 = *(const float4*)(bp + bhalf);
        };
        auto advance = [&]() {
            bp += bstep;
            advance_a(s0);
            advance_a(s1);
        };
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
        Slab f0, f1;   // slabs i and i + 1 in flight while slab i - 1 multiplies
        fetch(f0);
        if (ks0 + 1 < ks1) {
            advance();
            fetch(f1);
        }
        auto store_a = [&](int buf, int h, const float4& v) {
            if (AT) {
                *(float4*)&As[buf][ak + 16 * h][am] = v;
            } else {
                As[buf][ak + 0][am + 32 * h] = v.x;
                As[buf][ak + 1][am + 32 * h] = v.y;
                As[buf][ak + 2][am + 32 * h] = v.z;
                As[buf][ak + 3][am + 32 * h] = v.w;
            }
        };
        auto step = [&](int ks, int buf, Slab& f) {
            store_a(buf, 0, f.a0);
            store_a(buf, 1, f.a1);
            *(float4*)&Bs[buf][bk][bn] = f.b0;
            *(float4*)&Bs[buf][bk + 16][bn] = f.b1;
            __syncthreads();
            if (ks + 2 < ks1) {
                advance();
                fetch(f);
            }
#pragma unroll
            for (int kk = 0; kk < 32; kk += 2) {
                const float a = As[buf][kk + (lane >> 5)][wm + (lane & 31)];
                const float b = Bs[buf][kk + (lane >> 5)][wn + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        };
        for (int ks = ks0; ks < ks1; ks += 2) {
            step(ks, 0, f0);
            if (ks + 1 < ks1) step(ks + 1, 1, f1);
        }
        __syncthreads();   // the next segment's first store goes to buffer 0
        const bool whole = ks0 == 0 && ks1 == S;
        if (whole) {
            const size_t n = n0 + wn + (lane & 31);
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const size_t m = m0 + wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                float v = acc[r] + bv;
                if (accumulate) v += C[m * ldc + n];
                C[m * ldc + n] = v;
            }
        } else {   // slot 0: the range's first segment, slot 1: its last
            float* part = ws + ((size_t)blockIdx.x * 2 + (cur == mine.begin ? 0 : 1)) * 4096;
#pragma unroll
            for (int r = 0; r < 16; r++) part[(wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 64 + wn + (lane & 31)] = acc[r];
        }
        cur += ks1 - ks0;
    }
}

// One workgroup per tile: nothing to do for a tile that one workgroup multiplied whole; otherwise the parts, in ascending k.
// (32-bit range arithmetic: w * total < 2^32 for every launch — 64-bit divisions would be most of this kernel's time.)
__device__ inline unsigned sk_begin32(unsigned w, unsigned G, unsigned total) { return w * total / G; }

__global__ __launch_bounds__(256) void gemm_fixup_kernel(float* __restrict__ C, const float* __restrict__ bias, int S, int ldc, int accumulate,
                                                         int tiles_m, unsigned total, unsigned G, const float* __restrict__ ws) {
    const unsigned tile = blockIdx.x;
    const unsigned t0 = tile * (unsigned)S, t1 = t0 + (unsigned)S;
    unsigned w = t0 * G / total;
    while (w > 0 && sk_begin32(w, G, total) > t0) w--;
    while (sk_begin32(w + 1, G, total) <= t0) w++;
    if (sk_begin32(w, G, total) <= t0 && sk_begin32(w + 1, G, total) >= t1) return;
    const size_t m0 = (size_t)(tile % tiles_m) * 64, n0 = (size_t)(tile / tiles_m) * 64;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; w < G; w++) {
        const unsigned begin = sk_begin32(w, G, total);
        if (begin >= t1) break;
        // its first segment unless the range began in an earlier tile
        const float4* part = reinterpret_cast<const float4*>(ws + ((size_t)w * 2 + (begin >= t0 ? 0 : 1)) * 4096);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 p = part[i * 256 + threadIdx.x];
            v[i].x += p.x;
            v[i].y += p.y;
            v[i].z += p.z;
            v[i].w += p.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = (i * 256 + threadIdx.x) * 4;
        const size_t m = m0 + (e >> 6), n = n0 + (e & 63);
        float4 x = v[i];
        if (bias) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + n);
            x.x += bv.x;
            x.y += bv.y;
            x.z += bv.z;
            x.w += bv.w;
        }
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
        if (accumulate) {
            const float4 c = *dst;
            x.x += c.x;
            x.y += c.y;
            x.z += c.z;
            x.w += c.w;
        }
        *dst = x;
    }
}

// W'[(tap'*Co + co)][ci] = W[((8-tap')*Ci + ci)][co]: the weight matrix of the data gradient.  ldw / ldo = row strides.
__global__ void mirror_weights_kernel(const float* __restrict__ W, float* __restrict__ out, int Ci, int Co, int ldw,
                                      int ldo) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)9 * Co * Ci) return;
    const int ci = (int)(idx % Ci), co = (int)((idx / Ci) % Co), tap = (int)(idx / ((size_t)Ci * Co));
    out[((size_t)tap * Co + co) * ldo + ci] = W[((size_t)(8 - tap) * Ci + ci) * ldw + co];
}

// per-channel partial sums over a slice of the rows, in double.
// mode 0 (statistics): (x, x*x) of c.   mode 1 (BatchNorm backward): (dy, dy*xhat), dy = da * (a > 0).
__global__ __launch_bounds__(256) void column_partials_kernel(int mode, const float* __restrict__ c,
                                                              const float* __restrict__ da, const float* __restrict__ a,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int M,
                                                              double* __restrict__ partial) {
    __shared__ double s1[8][32], s2[8][32];
    const int ch = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
    const int rows = (M + SPLITS - 1) / SPLITS, lo = blockIdx.y * rows, hi = min(M, lo + rows);
    double x1 = 0.0, x2 = 0.0;
    if (mode == 0) {
        for (int m = lo + r; m < hi; m += 8) {
            const double v = c[(size_t)m * FILTERS + ch];
            x1 += v;
            x2 += v * v;
        }
    } else {
        const float mu = mean[ch], is = invstd[ch];
        for (int m = lo + r; m < hi; m += 8) {
            const size_t i = (size_t)m * FILTERS + ch;
            const float dy = a[i] > 0.f ? da[i] : 0.f;
            x1 += dy;
            x2 += (double)dy * (double)((c[i] - mu) * is);
        }
    }
    s1[r][threadIdx.x & 31] = x1;
    s2[r][threadIdx.x & 31] = x2;
    __syncthreads();
    if (r == 0) {
        for (int i = 1; i < 8; i++) {
            x1 += s1[i][threadIdx.x];
            x2 += s2[i][threadIdx.x];
        }
        partial[((size_t)blockIdx.y * FILTERS + ch) * 2 + 0] = x1;
        partial[((size_t)blockIdx.y * FILTERS + ch) * 2 + 1] = x2;
    }
}

// batch statistics + running statistics (torch: biased variance normalises, unbiased one is tracked)
__global__ void bn_stats_finish_kernel(const double* __restrict__ partial, int M, float* mean, float* invstd,
                                       float* running_mean, float* running_var) {
    const int ch = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < SPLITS; s++) {
        s1 += partial[((size_t)s * FILTERS + ch) * 2];
        s2 += partial[((size_t)s * FILTERS + ch) * 2 + 1];
    }
    const double mu = s1 / M;
    double var = s2 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[ch] = (float)mu;
    invstd[ch] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    running_mean[ch] = (1.f - BN_MOMENTUM) * running_mean[ch] + BN_MOMENTUM * (float)mu;
    running_var[ch] = (1.f - BN_MOMENTUM) * running_var[ch] + BN_MOMENTUM * (float)(var * M / (M - 1));
}

// a = relu(gamma * (c - mean) * invstd + beta (+ skip))
__global__ void bn_apply_kernel(const float* __restrict__ c, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ skip, float* __restrict__ a,
                                size_t total) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    const int ch = (int)(i % FILTERS);
    const float4 x = *(const float4*)(c + i), mu = *(const float4*)(mean + ch), is = *(const float4*)(invstd + ch);
    const float4 g = *(const float4*)(gamma + ch), b = *(const float4*)(beta + ch);
    float4 y;
    y.x = (x.x - mu.x) * is.x * g.x + b.x;
    y.y = (x.y - mu.y) * is.y * g.y + b.y;
    y.z = (x.z - mu.z) * is.z * g.z + b.z;
    y.w = (x.w - mu.w) * is.w * g.w + b.w;
    if (skip) {
        const float4 s = *(const float4*)(skip + i);
        y.x += s.x;
        y.y += s.y;
        y.z += s.z;
        y.w += s.w;
    }
    y.x = fmaxf(y.x, 0.f);
    y.y = fmaxf(y.y, 0.f);
    y.z = fmaxf(y.z, 0.f);
    y.w = fmaxf(y.w, 0.f);
    *(float4*)(a + i) = y;
}

// sums of the backward partials -> d gamma, d beta (into the gradient arena) and the two means the data term needs
__global__ void bn_bwd_finish_kernel(const double* __restrict__ partial, int M, float* dgamma, float* dbeta,
                                     float* mean_dy, float* mean_dyx) {
    const int ch = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < SPLITS; s++) {
        s1 += partial[((size_t)s * FILTERS + ch) * 2];
        s2 += partial[((size_t)s * FILTERS + ch) * 2 + 1];
    }
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
    mean_dy[ch] = (float)(s1 / M);
    mean_dyx[ch] = (float)(s2 / M);
}

// dc = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)),  dy = da * (a > 0);  dskip (optional) = dy
__global__ void bn_bwd_apply_kernel(const float* __restrict__ c, const float* __restrict__ da,
                                    const float* __restrict__ a, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ mean_dy, const float* __restrict__ mean_dyx,
                                    float* __restrict__ dc, float* __restrict__ dskip, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % FILTERS);
    const float dy = a[i] > 0.f ? da[i] : 0.f;
    const float xhat = (c[i] - mean[ch]) * invstd[ch];
    dc[i] = gamma[ch] * invstd[ch] * (dy - mean_dy[ch] - xhat * mean_dyx[ch]);
    if (dskip) dskip[i] = dy;
}

// value / UBE heads forward: conv1x1 (256 -> 1) + bias, ReLU, Linear (nn -> 1), tanh for value.  One wave per board.
// hp = [wv 256 | bv | wl nn | bl] (value) and the same for ube; pre[2][M] keeps the conv outputs for backward.
__global__ __launch_bounds__(64) void heads_fwd_kernel(const float* __restrict__ T, const float* __restrict__ vconv_w,
                                                       const float* __restrict__ vconv_b, const float* __restrict__ vlin_w,
                                                       const float* __restrict__ vlin_b, const float* __restrict__ uconv_w,
                                                       const float* __restrict__ uconv_b, const float* __restrict__ ulin_w,
                                                       const float* __restrict__ ulin_b, int nn, float* pre, int M,
                                                       float* value, float* ube) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float pv = 0.f, pu = 0.f;
    if (lane < nn) {
        const float* row = T + ((size_t)b * nn + lane) * FILTERS;
        float sv = 0.f, su = 0.f;
        for (int c = 0; c < FILTERS; c++) {
            sv += row[c] * vconv_w[c];
            su += row[c] * uconv_w[c];
        }
        sv += vconv_b[0];
        su += uconv_b[0];
        pre[(size_t)b * nn + lane] = sv;
        pre[(size_t)M + (size_t)b * nn + lane] = su;
        pv = fmaxf(sv, 0.f) * vlin_w[lane];
        pu = fmaxf(su, 0.f) * ulin_w[lane];
    }
    for (int o = 32; o > 0; o >>= 1) {
        pv += __shfl_down(pv, o);
        pu += __shfl_down(pu, o);
    }
    if (lane == 0) {
        value[b] = tanhf(pv + vlin_b[0]);
        ube[b] = pu + ulin_b[0];
    }
}

// heads backward.  Per board partial gradients (summed over boards in board order by heads_reduce_kernel):
// part[b] = [d vconv_w 256 | d vconv_b | d vlin_w nn | d vlin_b | the same for ube], and the value head's
// contribution to dT (the UBE head reads a detached trunk: net5.rs:188-189).
__global__ __launch_bounds__(64) void heads_bwd_kernel(const float* __restrict__ T, const float* __restrict__ vconv_w,
                                                       const float* __restrict__ vlin_w, const float* __restrict__ ulin_w,
                                                       const float* __restrict__ pre, int M, int nn, int batch,
                                                       const float* __restrict__ value, const float* __restrict__ ube,
                                                       const float* __restrict__ tv, const float* __restrict__ tu,
                                                       int train_ube, float* __restrict__ part, int pstride,
                                                       float* __restrict__ dT, float* loss_v, float* loss_u) {
    __shared__ float dv1[64], du1[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float v = value[b], u = ube[b];
    const float ev = v - tv[b];
    const float tgt_u = fminf(fmaxf(logf(tu[b]), MINIMUM_UBE_TARGET), logf(MAXIMUM_VARIANCE));  // learn/src/main.rs:360-363
    const float eu = u - tgt_u;
    const float dpre_v = 2.f * ev / batch * (1.f - v * v);
    const float dpre_u = train_ube ? 2.f * eu / batch : 0.f;
    float* P = part + (size_t)b * pstride;
    float* Pu = P + FILTERS + 1 + nn + 1;
    float sv = 0.f, su = 0.f;
    if (lane < nn) {
        const float cv = pre[(size_t)b * nn + lane], cu = pre[(size_t)M + (size_t)b * nn + lane];
        P[FILTERS + 1 + lane] = dpre_v * fmaxf(cv, 0.f);
        Pu[FILTERS + 1 + lane] = dpre_u * fmaxf(cu, 0.f);
        sv = cv > 0.f ? dpre_v * vlin_w[lane] : 0.f;
        su = cu > 0.f ? dpre_u * ulin_w[lane] : 0.f;
    }
    dv1[lane] = sv;
    du1[lane] = su;
    float tsv = sv, tsu = su;
    for (int o = 32; o > 0; o >>= 1) {
        tsv += __shfl_down(tsv, o);
        tsu += __shfl_down(tsu, o);
    }
    if (lane == 0) {
        P[FILTERS] = tsv;
        Pu[FILTERS] = tsu;
        P[FILTERS + 1 + nn] = dpre_v;
        Pu[FILTERS + 1 + nn] = dpre_u;
        loss_v[b] = ev * ev;
        loss_u[b] = train_ube ? eu * eu : 0.f;
    }
    __syncthreads();
    for (int c = lane; c < FILTERS; c += 64) {
        float gv = 0.f, gu = 0.f;
        const float w = vconv_w[c];
        for (int px = 0; px < nn; px++) {
            const size_t i = ((size_t)b * nn + px) * FILTERS + c;
            const float x = T[i];
            gv += dv1[px] * x;
            gu += du1[px] * x;
            dT[i] = dv1[px] * w;
        }
        P[c] = gv;
        Pu[c] = gu;
    }
}

// out[j] = sum over boards (in order) of part[b][j]
__global__ void sum_rows_kernel(const float* __restrict__ part, int rows, int stride, int count, float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    float s = 0.f;
    for (int b = 0; b < rows; b++) s += part[(size_t)b * stride + j];
    out[j] = s;
}

// policy loss and its gradient (learn/src/main.rs:384-391): masked_fill(mask, f32::MIN), log_softmax over the
// board's c*nn + px outputs, loss = -sum(logp * target) / batch.  One workgroup per board.
// pol / dpol are [M][np] NHWC (channel c of pixel px); target / mask are [batch][pol_ch * nn] in the reference's order.
__global__ __launch_bounds__(256) void policy_loss_kernel(const float* __restrict__ pol, const float* __restrict__ target,
                                                          const uint8_t* __restrict__ mask, int nn, int pol_ch, int np,
                                                          int batch, float* __restrict__ dpol, float* __restrict__ loss_p) {
    __shared__ float red[256];
    __shared__ float sh_max, sh_sum, sh_tsum;
    const int b = blockIdx.x, t = threadIdx.x, out = pol_ch * nn;
    const float* tg = target + (size_t)b * out;
    const uint8_t* mk = mask + (size_t)b * out;
    float mx = -3.402823466e38f;
    for (int e = t; e < out; e += 256) {
        const int c = e / nn, px = e % nn;
        if (!mk[e]) mx = fmaxf(mx, pol[((size_t)b * nn + px) * np + c]);
    }
    red[t] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = fmaxf(red[t], red[t + o]);
        __syncthreads();
    }
    if (t == 0) sh_max = red[0];
    __syncthreads();
    mx = sh_max;
    float se = 0.f, ts = 0.f;
    for (int e = t; e < out; e += 256) {
        const int c = e / nn, px = e % nn;
        if (!mk[e]) se += expf(pol[((size_t)b * nn + px) * np + c] - mx);
        ts += tg[e];
    }
    red[t] = se;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) sh_sum = red[0];
    __syncthreads();
    red[t] = ts;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) sh_tsum = red[0];
    __syncthreads();
    const float lse = mx + logf(sh_sum), tsum = sh_tsum;
    float l = 0.f;
    for (int e = t; e < np * nn; e += 256) {  // every element of the padded tensor gets a gradient (zero in the padding)
        const int c = e / nn, px = e % nn;
        float g = 0.f;
        if (c < pol_ch && !mk[c * nn + px]) {
            const float logp = pol[((size_t)b * nn + px) * np + c] - lse;
            const float tgt = tg[c * nn + px];
            l -= logp * tgt;
            g = (expf(logp) * tsum - tgt) / batch;
        }
        dpol[((size_t)b * nn + px) * np + c] = g;
    }
    red[t] = l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) loss_p[b] = red[0];
}

// column sums of a [M][ld] matrix (bias gradient of the policy conv): one workgroup per 32 columns, 8 row lanes
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ x, int M, int ld, int count,
                                                         float* __restrict__ out) {
    __shared__ double part[8][32];
    const int j = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
    double s = 0.0;
    if (j < count)
        for (int m = r; m < M; m += 8) s += x[(size_t)m * ld + j];
    part[r][threadIdx.x & 31] = s;
    __syncthreads();
    if (r == 0 && j < count) {
        for (int i = 1; i < 8; i++) s += part[i][threadIdx.x];
        out[j] = (float)s;
    }
}

__global__ void losses_kernel(const float* lp, const float* lv, const float* lu, int batch, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double p = 0.0, v = 0.0, u = 0.0;
    for (int b = 0; b < batch; b++) {
        p += lp[b];
        v += lv[b];
        u += lu[b];
    }
    out[0] = (float)(p / batch);
    out[1] = (float)(v / batch);
    out[2] = (float)(u / batch);
}

// torch.optim.Adam (tch nn::Adam::default(): beta1 0.9, beta2 0.999, eps 1e-8, no weight decay, no amsgrad).
// group[block of 1024 elements]: 0 = not trained, 1 = stepped every call, 2 = the UBE head (stepped when trained).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M1,
                                                   float* __restrict__ M2, const uint8_t* __restrict__ group, float lr,
                                                   int t_main, int t_ube) {
    const int g = group[blockIdx.x];
    const int step = g == 1 ? t_main : g == 2 ? t_ube : 0;
    if (step <= 0) return;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    for (int j = 0; j < 4; j++) {
        const size_t i = i0 + j;
        const float gr = G[i];
        const float m = b1 * M1[i] + (1.f - b1) * gr;
        const float v = b2 * M2[i] + (1.f - b2) * gr * gr;
        M1[i] = m;
        M2[i] = v;
        P[i] -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}

// ------------------------------------------------------------------------------------------------
struct Param {
    std::string name;
    size_t off = 0, slots = 0;     // offset / extent in the arenas (multiple of 1024)
    int kind = 0;                  // 0 vector as is, 1 conv weight [co][ci][3][3] <-> GEMM layout, 2 buffer (not trained)
    int co = 0, ci = 0, ld = 0, kp = 0;
    size_t count = 0;              // elements in the reference's tensor
    int group = 0;
};

}  // namespace

struct tz_trainer {
    int arch = 0, n = 0, nn = 0, blocks = 0, batch = 0, M = 0, device = 0;
    int cin = 0, kp_in = 0, pol_ch = 0, np = 0, layers = 0;
    float lr = 1e-4f;
    int t_main = 0, t_ube = 0;
    std::vector<Param> params;
    std::map<std::string, int> index;
    size_t total = 0;
    float *P = nullptr, *G = nullptr, *M1 = nullptr, *M2 = nullptr;
    uint8_t* group_dev = nullptr;
    // activations of one step
    tz_state* states = nullptr;
    float* x0 = nullptr;                 // [M][cin]
    std::vector<float*> c, a;            // per conv layer: pre-BN output, post-activation output  [M][256]
    float* stats = nullptr;              // [layers][2][256] mean, invstd
    double* partial = nullptr;           // [SPLITS][256][2]
    float *mean_dy = nullptr, *mean_dyx = nullptr;
    float* wmirror = nullptr;            // [layers][max K'][256]: slot l - 1 for trunk layer l, the last for the policy conv (mirror_all)
    int kmax = 0;
    hipEvent_t ev_mirror = nullptr;
    float *dA = nullptr, *dB = nullptr, *dC = nullptr, *dskip = nullptr;  // [M][256] gradient ping-pong
    float *pol = nullptr, *dpol = nullptr;                                // [M][np]
    float *pre = nullptr, *value = nullptr, *ube = nullptr;
    float *tpol = nullptr, *tv = nullptr, *tu = nullptr;
    uint8_t* mask = nullptr;
    float *part = nullptr, *loss_p = nullptr, *loss_v = nullptr, *loss_u = nullptr, *losses = nullptr;
    int pstride = 0;
    hipStream_t stream = nullptr;
    // weight gradients run on a second stream beside the data-gradient chain (two dC slots)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_dc[2] = {nullptr, nullptr}, ev_w[2] = {nullptr, nullptr};
    float* dC2 = nullptr;
    // stream-K GEMMs: workgroups per launch (two per CU) and, per stream, two 64x64 slots per workgroup for partial tiles
    int sk_groups = 512;
    float* sk_ws[2] = {nullptr, nullptr};
    std::vector<void*> allocs;
    TensorStore extra;   // variables of the VarStore the step never touches (RND nets, SimHash matrix): carried through save / load
};

namespace {

int add_param(tz_trainer* t, const std::string& name, int kind, size_t count, int group, int co = 0, int ci = 0) {
    Param p;
    p.name = name;
    p.kind = kind;
    p.count = count;
    p.group = group;
    size_t elems = count;
    if (kind == 1) {
        p.co = co;
        p.ci = ci;
        p.ld = (co + 63) / 64 * 64;
        p.kp = (9 * ci + 63) / 64 * 64;  // K of the forward GEMM and M of the weight-gradient GEMM
        elems = (size_t)p.kp * p.ld;
    } else if (name.find("policy.conv2d.bias") != std::string::npos) {
        elems = (size_t)(count + 63) / 64 * 64;
    }
    p.slots = (elems + 1023) / 1024 * 1024;
    p.off = t->total;
    t->total += p.slots;
    t->index[name] = (int)t->params.size();
    t->params.push_back(p);
    return TZ_OK;
}

template <typename T>
int dalloc(tz_trainer* t, T** p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return tz_fail(TZ_ENOMEM, "trainer: device allocation failed");
    if (hipMemset(q, 0, count * sizeof(T)) != hipSuccess) return tz_fail(TZ_EDEVICE, "trainer: memset failed");
    t->allocs.push_back(q);
    *p = (T*)q;
    return TZ_OK;
}

const Param& par(const tz_trainer* t, const std::string& name) { return t->params[t->index.at(name)]; }
float* pp(const tz_trainer* t, const std::string& name) { return t->P + par(t, name).off; }
float* gp(const tz_trainer* t, const std::string& name) { return t->G + par(t, name).off; }

std::string conv_name(const tz_trainer* t, int layer) {
    if (layer == 0) return "core.input_conv2d";
    const int b = (layer - 1) / 2;
    return "core.res_block_" + std::to_string(b) + ((layer - 1) % 2 ? ".b" : ".a") + ".conv2d";
}
std::string bn_name(const tz_trainer* t, int layer) {
    if (layer == 0) return "core.batch_norm";
    const int b = (layer - 1) / 2;
    return "core.res_block_" + std::to_string(b) + ((layer - 1) % 2 ? ".b" : ".a") + ".batch_norm";
}

int launch_check(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return TZ_OK;
}

int gemm(tz_trainer* t, bool at, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
         int ldb, int ldc, bool accumulate, hipStream_t st = nullptr, int gather_c = 0) {
    if (!st) st = t->stream;
    if (M % 64 || N % 64 || K % 32 || lda % 4 || ldb % 4 || ldc % 4 || gather_c % 4 || (long long)(M / 64) * (N / 64) * (K / 32) > (1ll << 22))
        return tz_fail(TZ_EINVAL, "trainer gemm: unaligned shape");
    const int acc = accumulate ? 1 : 0;
    static const bool tiled = getenv("TZ_LEARN_GEMM") && !strcmp(getenv("TZ_LEARN_GEMM"), "tile");   // A/B: one workgroup per tile
    if (!tiled) {
        const int tiles_m = M / 64, tiles = tiles_m * (N / 64), S = K / 32;
        const long long total = (long long)tiles * S;
        const int G = (int)std::min<long long>(total, t->sk_groups);
        float* ws = st == t->stream2 ? t->sk_ws[1] : t->sk_ws[0];
        const int gn = gather_c ? t->n : 0;
        if (gather_c && at)
            gemm_sk_kernel<true, true><<<G, 256, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, gn, gather_c, tiles_m, total, ws);
        else if (gather_c)
            gemm_sk_kernel<false, true><<<G, 256, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, gn, gather_c, tiles_m, total, ws);
        else if (at)
            gemm_sk_kernel<true, false><<<G, 256, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, 0, 0, tiles_m, total, ws);
        else
            gemm_sk_kernel<false, false><<<G, 256, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, 0, 0, tiles_m, total, ws);
        if (total % G || (total / G) % S)   // some tile is shared between workgroups
            gemm_fixup_kernel<<<tiles, 256, 0, st>>>(C, bias, S, ldc, acc, tiles_m, (unsigned)total, (unsigned)G, ws);
        return launch_check("gemm_sk");
    }
    const dim3 grid(M / 64, N / 64);
    if (gather_c) {   // A = im2col view of the NHWC tensor at `A` with gather_c channels
        if (at)
            gemm_f32_kernel<true, true><<<grid, 512, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, t->n, gather_c);
        else
            gemm_f32_kernel<false, true><<<grid, 512, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, t->n, gather_c);
    } else if (at) {
        gemm_f32_kernel<true, false><<<grid, 512, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, 0, 0);
    } else {
        gemm_f32_kernel<false, false><<<grid, 512, 0, st>>>(A, B, C, bias, K, lda, ldb, ldc, acc, 0, 0);
    }
    return launch_check("gemm_f32");
}

int forward(tz_trainer* t) {
    int rc;
    const int M = t->M;
    const size_t act = (size_t)M * FILTERS;
    if ((rc = tz_nn_encode_planes(t->n, t->cin, t->states, t->batch, t->x0, t->stream))) return rc;
    for (int l = 0; l < t->layers; l++) {
        const float* in = l == 0 ? t->x0 : t->a[l - 1];
        const int C = l == 0 ? t->cin : FILTERS, Kp = l == 0 ? t->kp_in : 9 * FILTERS;
        if ((rc = gemm(t, false, in, pp(t, conv_name(t, l) + ".weight"), t->c[l], nullptr, M, FILTERS, Kp, Kp, FILTERS,
                       FILTERS, false, nullptr, C)))
            return rc;
        float* mean = t->stats + (size_t)l * 2 * FILTERS;
        float* invstd = mean + FILTERS;
        const std::string bn = bn_name(t, l);
        column_partials_kernel<<<dim3(FILTERS / 32, SPLITS), 256, 0, t->stream>>>(0, t->c[l], nullptr, nullptr, nullptr,
                                                                                  nullptr, M, t->partial);
        bn_stats_finish_kernel<<<1, FILTERS, 0, t->stream>>>(t->partial, M, mean, invstd, pp(t, bn + ".running_mean"),
                                                             pp(t, bn + ".running_var"));
        const float* skip = (l > 0 && (l - 1) % 2 == 1) ? (l == 2 ? t->a[0] : t->a[l - 2]) : nullptr;
        bn_apply_kernel<<<(unsigned)((act / 4 + 255) / 256), 256, 0, t->stream>>>(
            t->c[l], mean, invstd, pp(t, bn + ".weight"), pp(t, bn + ".bias"), skip, t->a[l], act);
        if ((rc = launch_check("batch norm forward"))) return rc;
    }
    const float* T = t->a[t->layers - 1];
    if ((rc = gemm(t, false, T, pp(t, "policy.conv2d.weight"), t->pol, pp(t, "policy.conv2d.bias"), M, t->np,
                   9 * FILTERS, 9 * FILTERS, t->np, t->np, false, nullptr, FILTERS)))
        return rc;
    heads_fwd_kernel<<<t->batch, 64, 0, t->stream>>>(T, pp(t, "value.conv2d.weight"), pp(t, "value.conv2d.bias"),
                                                    pp(t, "value.linear.weight"), pp(t, "value.linear.bias"),
                                                    pp(t, "ube.conv2d.weight"), pp(t, "ube.conv2d.bias"),
                                                    pp(t, "ube.linear.weight"), pp(t, "ube.linear.bias"), t->nn, t->pre, M,
                                                    t->value, t->ube);
    return launch_check("heads forward");
}

// data gradient of a 3x3 conv: out[M][256] (+)= im2col(dc [M][Cd]) x mirror(W); `wm` = the mirrored matrix (mirror_all)
int conv_dgrad(tz_trainer* t, const float* dc, int Cd, const float* wm, float* out, bool accumulate) {
    return gemm(t, false, dc, wm, out, nullptr, t->M, FILTERS, 9 * Cd, 9 * Cd, FILTERS, FILTERS, accumulate, nullptr, Cd);
}

// The mirrored weight matrices of every data gradient of the step (trunk layers 1 .., slot l - 1; the policy conv in the last slot),
// built on the second stream while the forward pass runs on the first: they depend on the weights only, and one small launch ahead of
// each of the 41 data-gradient GEMMs was 0.26 ms of the main stream's critical path.
int mirror_all(tz_trainer* t) {
    const size_t slot = (size_t)t->kmax * FILTERS;
    for (int l = 1; l <= t->layers; l++) {
        const bool policy = l == t->layers;
        const int Cd = policy ? t->np : FILTERS;
        const float* W = pp(t, policy ? std::string("policy.conv2d.weight") : conv_name(t, l) + ".weight");
        const size_t total = (size_t)9 * Cd * FILTERS;
        mirror_weights_kernel<<<(unsigned)((total + 255) / 256), 256, 0, t->stream2>>>(W, t->wmirror + (size_t)(l - 1) * slot, FILTERS, Cd, Cd, FILTERS);
    }
    int rc;
    if ((rc = launch_check("mirror weights"))) return rc;
    TZ_HIP(hipEventRecord(t->ev_mirror, t->stream2));
    return TZ_OK;
}

int backward(tz_trainer* t, int train_ube) {
    int rc;
    const int M = t->M, L = t->layers;
    const size_t act = (size_t)M * FILTERS;
    const float* T = t->a[L - 1];
    // losses and the gradients at the three outputs
    policy_loss_kernel<<<t->batch, 256, 0, t->stream>>>(t->pol, t->tpol, t->mask, t->nn, t->pol_ch, t->np, t->batch, t->dpol,
                                                       t->loss_p);
    heads_bwd_kernel<<<t->batch, 64, 0, t->stream>>>(T, pp(t, "value.conv2d.weight"), pp(t, "value.linear.weight"),
                                                    pp(t, "ube.linear.weight"), t->pre, M, t->nn, t->batch, t->value, t->ube,
                                                    t->tv, t->tu, train_ube, t->part, t->pstride, t->dA, t->loss_v,
                                                    t->loss_u);
    losses_kernel<<<1, 1, 0, t->stream>>>(t->loss_p, t->loss_v, t->loss_u, t->batch, t->losses);
    if ((rc = launch_check("losses"))) return rc;
    {   // head parameter gradients: part[b] = [conv w 256 | conv b | lin w nn | lin b] x {value, ube}
        const int hs = FILTERS + 1 + t->nn + 1;
        const char* heads[2] = {"value", "ube"};
        for (int h = 0; h < 2; h++) {
            const float* base = t->part + (size_t)h * hs;
            const std::string p = heads[h];
            sum_rows_kernel<<<1, 256, 0, t->stream>>>(base, t->batch, t->pstride, FILTERS, gp(t, p + ".conv2d.weight"));
            sum_rows_kernel<<<1, 64, 0, t->stream>>>(base + FILTERS, t->batch, t->pstride, 1, gp(t, p + ".conv2d.bias"));
            sum_rows_kernel<<<1, 64, 0, t->stream>>>(base + FILTERS + 1, t->batch, t->pstride, t->nn, gp(t, p + ".linear.weight"));
            sum_rows_kernel<<<1, 64, 0, t->stream>>>(base + FILTERS + 1 + t->nn, t->batch, t->pstride, 1, gp(t, p + ".linear.bias"));
        }
        if ((rc = launch_check("head gradients"))) return rc;
    }
    // policy conv: weight / bias gradient, and its data gradient added to the value head's (already in dA)
    if ((rc = gemm(t, true, T, t->dpol, gp(t, "policy.conv2d.weight"), nullptr, 9 * FILTERS, t->np, M, 9 * FILTERS, t->np,
                   t->np, false, nullptr, FILTERS)))
        return rc;
    column_sum_kernel<<<(t->np + 31) / 32, 256, 0, t->stream>>>(t->dpol, M, t->np, t->np, gp(t, "policy.conv2d.bias"));
    if ((rc = launch_check("policy bias gradient"))) return rc;
    TZ_HIP(hipStreamWaitEvent(t->stream, t->ev_mirror, 0));
    if ((rc = conv_dgrad(t, t->dpol, t->np, t->wmirror + (size_t)(t->layers - 1) * t->kmax * FILTERS, t->dA, true))) return rc;
    // trunk, last layer first.  `da` = gradient w.r.t. a[l]; three [M][256] buffers rotate between the roles
    // "gradient coming in", "gradient of the skip connection" and "gradient going out".
    float* bufs[3] = {t->dA, t->dB, t->dskip};
    float* da = t->dA;
    float* skip = nullptr;
    auto free_buf = [&](const float* x, const float* y) {
        for (float* b : bufs)
            if (b != x && b != y) return b;
        return (float*)nullptr;
    };
    // events of this step only: the previous step ended with the main stream waiting for both (below), and a wait for an event that
    // was recorded outside a stream capture has no place in the captured graph
    bool recorded[2] = {false, false};
    for (int l = L - 1; l >= 0; l--) {
        const bool second = l > 0 && (l - 1) % 2 == 1;  // second SmallBlock of a residual block: output joins the skip
        const float* mean = t->stats + (size_t)l * 2 * FILTERS;
        const float* invstd = mean + FILTERS;
        const std::string bn = bn_name(t, l), cv = conv_name(t, l);
        if (second) skip = free_buf(da, nullptr);
        const int slot = l & 1;
        float* dc = slot ? t->dC2 : t->dC;
        if (recorded[slot]) TZ_HIP(hipStreamWaitEvent(t->stream, t->ev_w[slot], 0));  // the weight gradient of layer l+2 has read this slot
        column_partials_kernel<<<dim3(FILTERS / 32, SPLITS), 256, 0, t->stream>>>(1, t->c[l], da, t->a[l], mean, invstd, M,
                                                                                  t->partial);
        bn_bwd_finish_kernel<<<1, FILTERS, 0, t->stream>>>(t->partial, M, gp(t, bn + ".weight"), gp(t, bn + ".bias"),
                                                           t->mean_dy, t->mean_dyx);
        bn_bwd_apply_kernel<<<(unsigned)((act + 255) / 256), 256, 0, t->stream>>>(
            t->c[l], da, t->a[l], mean, invstd, pp(t, bn + ".weight"), t->mean_dy, t->mean_dyx, dc,
            second ? skip : nullptr, act);
        if ((rc = launch_check("batch norm backward"))) return rc;
        TZ_HIP(hipEventRecord(t->ev_dc[slot], t->stream));
        // weight gradient on the second stream: dW = im2col(input)^T x dc
        const float* in = l == 0 ? t->x0 : t->a[l - 1];
        const int C = l == 0 ? t->cin : FILTERS, Kp = l == 0 ? t->kp_in : 9 * FILTERS;
        TZ_HIP(hipStreamWaitEvent(t->stream2, t->ev_dc[slot], 0));
        if ((rc = gemm(t, true, in, dc, gp(t, cv + ".weight"), nullptr, Kp, FILTERS, M, Kp, FILTERS, FILTERS, false,
                       t->stream2, C)))
            return rc;
        TZ_HIP(hipEventRecord(t->ev_w[slot], t->stream2));
        recorded[slot] = true;
        if (l == 0) break;
        if (second) {  // gradient w.r.t. a[l-1] (the ReLU between the two SmallBlocks)
            float* out = free_buf(da, skip);
            if ((rc = conv_dgrad(t, dc, FILTERS, t->wmirror + (size_t)(l - 1) * t->kmax * FILTERS, out, false))) return rc;
            da = out;
        } else {       // gradient w.r.t. the block input: through the first SmallBlock plus the skip connection
            if ((rc = conv_dgrad(t, dc, FILTERS, t->wmirror + (size_t)(l - 1) * t->kmax * FILTERS, skip, true))) return rc;
            da = skip;
            skip = nullptr;
        }
    }
    // every weight gradient is in place before anything else (Adam, a read-back) runs on the main stream
    for (int slot = 0; slot < 2; slot++)
        if (recorded[slot]) TZ_HIP(hipStreamWaitEvent(t->stream, t->ev_w[slot], 0));
    return TZ_OK;
}

}  // namespace
// ------------------------------------------------------------------------------------------------
// C ABI (include/takzero_hip.h, "Trainer")
extern "C" {

int tz_trainer_create(int board_n, int arch, int device_id, int blocks, int batch, float learning_rate, tz_trainer** out) {
    if (!out) return tz_fail(TZ_EINVAL, "tz_trainer_create: null out");
    *out = nullptr;
    if (board_n < 3 || board_n > 6) return tz_fail(TZ_EINVAL, "tz_trainer_create: board size must be 3..6");
    if (arch == TZ_ARCH_NET5 && board_n != 5) return tz_fail(TZ_EINVAL, "tz_trainer_create: net5 is a 5x5 network");
    if (arch == TZ_ARCH_NET4_SIMHASH && board_n != 4) return tz_fail(TZ_EINVAL, "tz_trainer_create: net4_simhash is 4x4");
    if (arch == TZ_ARCH_NET6_SIMHASH && board_n != 6) return tz_fail(TZ_EINVAL, "tz_trainer_create: net6_simhash is 6x6");
    if (batch <= 0 || batch % 64) return tz_fail(TZ_EINVAL, "tz_trainer_create: batch must be a positive multiple of 64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
        return tz_fail(TZ_EDEVICE, "tz_trainer_create: no such HIP device (there is no CPU path)");
    TZ_HIP(hipSetDevice(device_id));
    tz_trainer* t = new tz_trainer();
    t->arch = arch;
    t->n = board_n;
    t->nn = board_n * board_n;
    t->device = device_id;
    t->blocks = arch == TZ_ARCH_NET5 ? 20 : arch == TZ_ARCH_TEST ? blocks : 16;
    t->batch = batch;
    t->M = batch * t->nn;
    t->cin = tz_input_channels(board_n);
    t->kp_in = (9 * t->cin + 63) / 64 * 64;
    t->pol_ch = 3 + 4 * ((1 << board_n) - 2);
    t->np = (t->pol_ch + 63) / 64 * 64;
    t->layers = 1 + 2 * t->blocks;
    t->lr = learning_rate;
    int rc = TZ_OK;
    auto bn = [&](const std::string& p) {
        add_param(t, p + ".weight", 0, FILTERS, 1);
        add_param(t, p + ".bias", 0, FILTERS, 1);
        add_param(t, p + ".running_mean", 2, FILTERS, 0);
        add_param(t, p + ".running_var", 2, FILTERS, 0);
    };
    add_param(t, "core.input_conv2d.weight", 1, (size_t)FILTERS * t->cin * 9, 1, FILTERS, t->cin);
    bn("core.batch_norm");
    for (int b = 0; b < t->blocks; b++)
        for (const char* half : {".a", ".b"}) {
            const std::string p = "core.res_block_" + std::to_string(b) + half;
            add_param(t, p + ".conv2d.weight", 1, (size_t)FILTERS * FILTERS * 9, 1, FILTERS, FILTERS);
            bn(p + ".batch_norm");
        }
    add_param(t, "policy.conv2d.weight", 1, (size_t)t->pol_ch * FILTERS * 9, 1, t->pol_ch, FILTERS);
    add_param(t, "policy.conv2d.bias", 0, t->pol_ch, 1);
    for (const char* h : {"value", "ube"}) {
        const int g = std::string(h) == "ube" ? 2 : 1;
        add_param(t, std::string(h) + ".conv2d.weight", 0, FILTERS, g);
        add_param(t, std::string(h) + ".conv2d.bias", 0, 1, g);
        add_param(t, std::string(h) + ".linear.weight", 0, t->nn, g);
        add_param(t, std::string(h) + ".linear.bias", 0, 1, g);
    }
    const size_t act = (size_t)t->M * FILTERS;
    const int kmax = 9 * (t->np > FILTERS ? t->np : FILTERS);
    t->pstride = 2 * (FILTERS + 1 + t->nn + 1);
    do {
        if ((rc = dalloc(t, &t->P, t->total))) break;
        if ((rc = dalloc(t, &t->G, t->total))) break;
        if ((rc = dalloc(t, &t->M1, t->total))) break;
        if ((rc = dalloc(t, &t->M2, t->total))) break;
        if ((rc = dalloc(t, &t->group_dev, t->total / 1024))) break;
        if ((rc = dalloc(t, &t->states, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->x0, (size_t)t->M * t->cin))) break;
        t->c.resize(t->layers);
        t->a.resize(t->layers);
        for (int l = 0; l < t->layers && !rc; l++) {
            if ((rc = dalloc(t, &t->c[l], act))) break;
            rc = dalloc(t, &t->a[l], act);
        }
        if (rc) break;
        if ((rc = dalloc(t, &t->stats, (size_t)t->layers * 2 * FILTERS))) break;
        if ((rc = dalloc(t, &t->partial, (size_t)SPLITS * FILTERS * 2))) break;
        if ((rc = dalloc(t, &t->mean_dy, (size_t)FILTERS))) break;
        if ((rc = dalloc(t, &t->mean_dyx, (size_t)FILTERS))) break;
        t->kmax = kmax;
        if ((rc = dalloc(t, &t->wmirror, (size_t)t->layers * kmax * FILTERS))) break;
        if ((rc = dalloc(t, &t->dA, act))) break;
        if ((rc = dalloc(t, &t->dB, act))) break;
        if ((rc = dalloc(t, &t->dC, act))) break;
        if ((rc = dalloc(t, &t->dC2, act))) break;
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) == hipSuccess && cus > 0) t->sk_groups = 2 * cus;
            if ((rc = dalloc(t, &t->sk_ws[0], (size_t)t->sk_groups * 2 * 4096))) break;
            if ((rc = dalloc(t, &t->sk_ws[1], (size_t)t->sk_groups * 2 * 4096))) break;
        }
        if ((rc = dalloc(t, &t->dskip, act))) break;
        if ((rc = dalloc(t, &t->pol, (size_t)t->M * t->np))) break;
        if ((rc = dalloc(t, &t->dpol, (size_t)t->M * t->np))) break;
        if ((rc = dalloc(t, &t->pre, (size_t)2 * t->M))) break;
        if ((rc = dalloc(t, &t->value, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->ube, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->tpol, (size_t)batch * t->pol_ch * t->nn))) break;
        if ((rc = dalloc(t, &t->mask, (size_t)batch * t->pol_ch * t->nn))) break;
        if ((rc = dalloc(t, &t->tv, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->tu, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->part, (size_t)batch * t->pstride))) break;
        if ((rc = dalloc(t, &t->loss_p, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->loss_v, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->loss_u, (size_t)batch))) break;
        if ((rc = dalloc(t, &t->losses, (size_t)4))) break;
    } while (0);
    if (!rc && (hipStreamCreate(&t->stream) != hipSuccess || hipStreamCreate(&t->stream2) != hipSuccess))
        rc = tz_fail(TZ_EDEVICE, "tz_trainer_create: stream");
    if (!rc && hipEventCreateWithFlags(&t->ev_mirror, hipEventDisableTiming) != hipSuccess) rc = tz_fail(TZ_EDEVICE, "tz_trainer_create: event");
    for (int i = 0; i < 2 && !rc; i++)
        if (hipEventCreateWithFlags(&t->ev_dc[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&t->ev_w[i], hipEventDisableTiming) != hipSuccess)
            rc = tz_fail(TZ_EDEVICE, "tz_trainer_create: event");
    if (!rc) {
        std::vector<uint8_t> groups(t->total / 1024, 0);
        for (const Param& p : t->params)
            for (size_t b = 0; b < p.slots / 1024; b++) groups[p.off / 1024 + b] = (uint8_t)p.group;
        if (hipMemcpy(t->group_dev, groups.data(), groups.size(), hipMemcpyHostToDevice) != hipSuccess)
            rc = tz_fail(TZ_EDEVICE, "tz_trainer_create: upload");
    }
    if (rc) {
        for (void* q : t->allocs) (void)hipFree(q);
        if (t->stream) (void)hipStreamDestroy(t->stream);
        if (t->stream2) (void)hipStreamDestroy(t->stream2);
        for (int i = 0; i < 2; i++) {
            if (t->ev_dc[i]) (void)hipEventDestroy(t->ev_dc[i]);
            if (t->ev_w[i]) (void)hipEventDestroy(t->ev_w[i]);
        }
        delete t;
        return rc;
    }
    *out = t;
    return TZ_OK;
}

int tz_trainer_destroy(tz_trainer* t) {
    if (!t) return TZ_OK;
    (void)hipSetDevice(t->device);
    (void)hipStreamSynchronize(t->stream);
    (void)hipStreamSynchronize(t->stream2);
    for (void* q : t->allocs) (void)hipFree(q);
    (void)hipStreamDestroy(t->stream);
    (void)hipStreamDestroy(t->stream2);
    for (int i = 0; i < 2; i++) {
        (void)hipEventDestroy(t->ev_dc[i]);
        (void)hipEventDestroy(t->ev_w[i]);
    }
    if (t->ev_mirror) (void)hipEventDestroy(t->ev_mirror);
    delete t;
    return TZ_OK;
}

// number of tensors the trainer owns, and the name / element count of tensor i (reference names, `.a.` / `.b.`)
int tz_trainer_shape(tz_trainer* t, int* board_n_out, int* batch_out, int* arch_out) {
    if (!t) return tz_fail(TZ_EINVAL, "tz_trainer_shape: null handle");
    if (board_n_out) *board_n_out = t->n;
    if (batch_out) *batch_out = t->batch;
    if (arch_out) *arch_out = t->arch;
    return TZ_OK;
}

int tz_trainer_tensor_count(tz_trainer* t) { return t ? (int)t->params.size() : 0; }

int tz_trainer_tensor_info(tz_trainer* t, int i, char* name_out, int name_cap, uint64_t* count_out) {
    if (!t || i < 0 || i >= (int)t->params.size() || !name_out || name_cap <= 0)
        return tz_fail(TZ_EINVAL, "tz_trainer_tensor_info: bad argument");
    const Param& p = t->params[i];
    if ((int)p.name.size() + 1 > name_cap) return tz_fail(TZ_EINVAL, "tz_trainer_tensor_info: name buffer too small");
    memcpy(name_out, p.name.c_str(), p.name.size() + 1);
    if (count_out) *count_out = p.count;
    return TZ_OK;
}

static float* arena_of(tz_trainer* t, int what) {
    return what == 0 ? t->P : what == 1 ? t->G : what == 2 ? t->M1 : what == 3 ? t->M2 : nullptr;
}

// what: 0 parameter, 1 gradient of the last step, 2 / 3 Adam first / second moment.  Data in the reference's layout.
int tz_trainer_set_tensor(tz_trainer* t, const char* name, int what, const float* data, uint64_t count) {
    if (!t || !name || !data) return tz_fail(TZ_EINVAL, "tz_trainer_set_tensor: null argument");
    auto it = t->index.find(name);
    float* arena = arena_of(t, what);
    if (it == t->index.end() || !arena) return tz_fail(TZ_EINVAL, std::string("tz_trainer_set_tensor: unknown tensor ") + name);
    const Param& p = t->params[it->second];
    if (count != p.count) return tz_fail(TZ_EPARSE, std::string("tz_trainer_set_tensor: wrong size for ") + name);
    TZ_HIP(hipSetDevice(t->device));
    std::vector<float> host(p.slots, 0.f);
    if (p.kind == 1) {
        for (int co = 0; co < p.co; co++)
            for (int ci = 0; ci < p.ci; ci++)
                for (int tap = 0; tap < 9; tap++)
                    host[((size_t)tap * p.ci + ci) * p.ld + co] = data[((size_t)co * p.ci + ci) * 9 + tap];
    } else {
        memcpy(host.data(), data, count * sizeof(float));
    }
    TZ_HIP(hipStreamSynchronize(t->stream));
    TZ_HIP(hipMemcpy(arena + p.off, host.data(), p.slots * sizeof(float), hipMemcpyHostToDevice));
    return TZ_OK;
}

int tz_trainer_get_tensor(tz_trainer* t, const char* name, int what, float* out, uint64_t count) {
    if (!t || !name || !out) return tz_fail(TZ_EINVAL, "tz_trainer_get_tensor: null argument");
    auto it = t->index.find(name);
    float* arena = arena_of(t, what);
    if (it == t->index.end() || !arena) return tz_fail(TZ_EINVAL, std::string("tz_trainer_get_tensor: unknown tensor ") + name);
    const Param& p = t->params[it->second];
    if (count != p.count) return tz_fail(TZ_EPARSE, std::string("tz_trainer_get_tensor: wrong size for ") + name);
    TZ_HIP(hipSetDevice(t->device));
    std::vector<float> host(p.slots);
    TZ_HIP(hipStreamSynchronize(t->stream));
    TZ_HIP(hipMemcpy(host.data(), arena + p.off, p.slots * sizeof(float), hipMemcpyDeviceToHost));
    if (p.kind == 1) {
        for (int co = 0; co < p.co; co++)
            for (int ci = 0; ci < p.ci; ci++)
                for (int tap = 0; tap < 9; tap++)
                    out[((size_t)co * p.ci + ci) * 9 + tap] = host[((size_t)tap * p.ci + ci) * p.ld + co];
    } else {
        memcpy(out, host.data(), count * sizeof(float));
    }
    return TZ_OK;
}

// shape of a variable in the reference's VarStore
static std::vector<uint32_t> param_dims(const tz_trainer* t, const Param& p) {
    auto ends = [&](const char* s) { const size_t n = strlen(s); return p.name.size() >= n && !p.name.compare(p.name.size() - n, n, s); };
    if (p.kind == 1) return {(uint32_t)p.co, (uint32_t)p.ci, 3u, 3u};
    if (ends("value.conv2d.weight") || ends("ube.conv2d.weight")) return {1u, (uint32_t)p.count, 1u, 1u};
    if (ends(".linear.weight")) return {1u, (uint32_t)t->nn};
    return {(uint32_t)p.count};
}

// The trainer's VarStore as a host store (current parameters and buffers + the carried extras)
int tz_trainer_snapshot(tz_trainer* t, TensorStore& out) {
    out = t->extra;
    for (const Param& p : t->params) {
        HostTensor h;
        h.dims = param_dims(t, p);
        h.data.resize(p.count);
        int rc = tz_trainer_get_tensor(t, p.name.c_str(), 0, h.data.data(), p.count);
        if (rc) return rc;
        out[p.name] = std::move(h);
    }
    return TZ_OK;
}

static int trainer_apply_store(tz_trainer* t, const TensorStore& st) {
    for (const Param& p : t->params) {
        auto it = st.find(p.name);
        if (it == st.end()) return tz_fail(TZ_EPARSE, "trainer: the model has no variable " + p.name);
        if (it->second.data.size() != p.count) return tz_fail(TZ_EPARSE, "trainer: wrong size of variable " + p.name);
    }
    for (const Param& p : t->params) {
        int rc = tz_trainer_set_tensor(t, p.name.c_str(), 0, st.at(p.name).data.data(), p.count);
        if (rc) return rc;
    }
    t->extra.clear();
    for (auto& kv : st)
        if (!t->index.count(kv.first)) t->extra[kv.first] = kv.second;
    return TZ_OK;
}

// Network::load / Network::save on the trainer's VarStore (learn/src/main.rs:107-120, 247-266): LibTorch archive or .tzw
int tz_trainer_load(tz_trainer* t, const char* path) {
    if (!t || !path) return tz_fail(TZ_EINVAL, "tz_trainer_load: null argument");
    TensorStore st;
    int rc = weights_read_file(path, st);
    if (rc) return rc;
    return trainer_apply_store(t, st);
}

int tz_trainer_save(tz_trainer* t, const char* path) {
    if (!t || !path) return tz_fail(TZ_EINVAL, "tz_trainer_save: null argument");
    TensorStore st;
    int rc = tz_trainer_snapshot(t, st);
    if (rc) return rc;
    return weights_write_file(path, st);
}

// the variables of a network (Net::new / a loaded model) become the trainer's, and back without a file
int tz_trainer_from_net(tz_trainer* t, tz_net* net) {
    if (!t || !net) return tz_fail(TZ_EINVAL, "tz_trainer_from_net: null argument");
    if (!net->loaded) return tz_fail(TZ_ESTATE, "tz_trainer_from_net: the network has no weights");
    return trainer_apply_store(t, net->store);
}

int tz_trainer_to_net(tz_trainer* t, tz_net* net) {
    if (!t || !net) return tz_fail(TZ_EINVAL, "tz_trainer_to_net: null argument");
    TensorStore st;
    int rc = tz_trainer_snapshot(t, st);
    if (rc) return rc;
    std::vector<unsigned char> blob;
    tzw_dump(st, blob);
    return tz_net_load_weights_mem(net, blob.data(), blob.size());
}

// compute_loss_and_take_step (learn/src/main.rs:376-423) on one batch:
//   states[batch]; target_policy[batch][policy_size] (policy_tensor); mask[batch][policy_size], 1 = not a legal move
//   (move_mask); target_value[batch]; target_ube[batch] = variance targets (the log / clamp of :360-363 is applied here).
//   losses_out[3] = policy, value, ube.  apply_step = 0 computes losses and gradients only.
int tz_trainer_step(tz_trainer* t, const tz_state* states, const float* target_policy, const uint8_t* mask,
                    const float* target_value, const float* target_ube, int train_ube, int apply_step, float* losses_out) {
    if (!t || !states || !target_policy || !mask || !target_value || !target_ube)
        return tz_fail(TZ_EINVAL, "tz_trainer_step: null argument");
    TZ_HIP(hipSetDevice(t->device));
    const size_t out = (size_t)t->pol_ch * t->nn;
    TZ_HIP(hipMemcpyAsync(t->states, states, sizeof(tz_state) * t->batch, hipMemcpyHostToDevice, t->stream));
    TZ_HIP(hipMemcpyAsync(t->tpol, target_policy, sizeof(float) * t->batch * out, hipMemcpyHostToDevice, t->stream));
    TZ_HIP(hipMemcpyAsync(t->mask, mask, (size_t)t->batch * out, hipMemcpyHostToDevice, t->stream));
    TZ_HIP(hipMemcpyAsync(t->tv, target_value, sizeof(float) * t->batch, hipMemcpyHostToDevice, t->stream));
    TZ_HIP(hipMemcpyAsync(t->tu, target_ube, sizeof(float) * t->batch, hipMemcpyHostToDevice, t->stream));
    int rc;
    // (forward + backward captured as one HIP graph per train_ube was measured and not kept: 9.27 against 7.90 ms per step — the
    // graph's two branches do not overlap the way the two streams do)
    if ((rc = mirror_all(t))) return rc;
    if ((rc = forward(t))) return rc;
    if ((rc = backward(t, train_ube))) return rc;
    if (apply_step) {
        t->t_main++;
        if (train_ube) t->t_ube++;
        adam_kernel<<<(unsigned)(t->total / 1024), 256, 0, t->stream>>>(t->P, t->G, t->M1, t->M2, t->group_dev, t->lr,
                                                                        t->t_main, train_ube ? t->t_ube : 0);
        if ((rc = launch_check("adam"))) return rc;
    }
    float losses[4] = {0, 0, 0, 0};
    TZ_HIP(hipMemcpyAsync(losses, t->losses, sizeof(float) * 3, hipMemcpyDeviceToHost, t->stream));
    TZ_HIP(hipStreamSynchronize(t->stream));
    if (losses_out) memcpy(losses_out, losses, sizeof(float) * 3);
    if (!(losses[0] == losses[0]) || !(losses[1] == losses[1]) || !(losses[2] == losses[2]))
        return tz_fail(TZ_ENUMERIC, "tz_trainer_step: a loss is NaN");
    return TZ_OK;
}

// network outputs of the last step's forward pass (training mode): policy [batch][policy_size] in the reference's
// c*nn + px order, value [batch] (after tanh), ube [batch] (log variance)
int tz_trainer_outputs(tz_trainer* t, float* policy_out, float* value_out, float* ube_out) {
    if (!t) return tz_fail(TZ_EINVAL, "tz_trainer_outputs: null trainer");
    TZ_HIP(hipSetDevice(t->device));
    TZ_HIP(hipStreamSynchronize(t->stream));
    if (policy_out) {
        std::vector<float> host((size_t)t->M * t->np);
        TZ_HIP(hipMemcpy(host.data(), t->pol, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int b = 0; b < t->batch; b++)
            for (int c = 0; c < t->pol_ch; c++)
                for (int px = 0; px < t->nn; px++)
                    policy_out[((size_t)b * t->pol_ch + c) * t->nn + px] = host[((size_t)b * t->nn + px) * t->np + c];
    }
    if (value_out) TZ_HIP(hipMemcpy(value_out, t->value, sizeof(float) * t->batch, hipMemcpyDeviceToHost));
    if (ube_out) TZ_HIP(hipMemcpy(ube_out, t->ube, sizeof(float) * t->batch, hipMemcpyDeviceToHost));
    return TZ_OK;
}

int tz_trainer_activation(tz_trainer* t, int layer, float* out, uint64_t cap) {
    if (!t || !out) return tz_fail(TZ_EINVAL, "tz_trainer_activation: null argument");
    if (layer < 0 || layer >= t->layers) return tz_fail(TZ_EINVAL, "tz_trainer_activation: no such trunk layer");
    const size_t real = (size_t)t->batch * t->nn * FILTERS;
    if (cap < real) return tz_fail(TZ_EINVAL, "tz_trainer_activation: the buffer holds fewer than batch * n * n * 256 values");
    TZ_HIP(hipSetDevice(t->device));
    TZ_HIP(hipStreamSynchronize(t->stream));
    TZ_HIP(hipMemcpy(out, t->a[layer], real * sizeof(float), hipMemcpyDeviceToHost));
    return TZ_OK;
}

}  // extern "C"
