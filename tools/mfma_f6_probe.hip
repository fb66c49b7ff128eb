// Probe behind TZ_PREC_F16C6 (round 3): what v_mfma_scale_f32_16x16x128_f8f6f4 does with FP6 (E2M3) operands and block scales,
// what v_cvt_scalef32_pk32_fp6_f16 emits, what the two permlane swaps move, and what they all cost.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f6_probe.hip -o /tmp/f6probe && /tmp/f6probe
// Facts it establishes (profiles/r03_mfma_f6_probe.txt):
//   * operand layout: lane l holds A[row l&15][k = 32 (l>>4) + i] / B[k][col l&15], element i in bits [6i, 6i+6) of its 192-bit string
//   * scales: byte `opsel` of the lane's scale register is an E8M0 exponent for THAT lane's 32 elements (row l&15, k-block l>>4)
//   * the conversion: element e of the source -> field e; round to nearest even; saturates at +-7.5; the scale divides
//   * cycles per instruction against the fp16 16x16x32 form, one and two waves per SIMD
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static float e2m3_to_float(unsigned c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float f = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -f : f;
}

// ---- layout + scale probe: codes[row][k] (6-bit), per-(row, kblock) scale bytes
__global__ void layout_probe(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* C, int opsel) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    unsigned wa[6] = {0, 0, 0, 0, 0, 0}, wb[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 32; i++) {
        const unsigned long long ca = A[r * 128 + 32 * q + i] & 63, cb = B[(32 * q + i) * 16 + r] & 63;
        const int bit = 6 * i, w = bit >> 5, sh = bit & 31;
        wa[w] |= (unsigned)(ca << sh);
        wb[w] |= (unsigned)(cb << sh);
        if (sh > 26) {
            wa[w + 1] |= (unsigned)(ca >> (32 - sh));
            wb[w + 1] |= (unsigned)(cb >> (32 - sh));
        }
    }
    v8i a = {(int)wa[0], (int)wa[1], (int)wa[2], (int)wa[3], (int)wa[4], (int)wa[5], 0, 0};
    v8i b = {(int)wb[0], (int)wb[1], (int)wb[2], (int)wb[3], (int)wb[4], (int)wb[5], 0, 0};
    // the lane's own scale in byte `opsel`, junk in the other bytes
    const int sa = (int)(0x11223344u & ~(0xffu << (8 * opsel))) | ((int)SA[r * 4 + q] << (8 * opsel));
    const int sb = (int)(0x55667788u & ~(0xffu << (8 * opsel))) | ((int)SB[r * 4 + q] << (8 * opsel));
    f32x4 c = {0, 0, 0, 0};
    if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa, 0, sb);
    else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 1, sa, 1, sb);
    else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 2, sa, 2, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 3, sa, 3, sb);
    for (int k = 0; k < 4; k++) C[(q * 4 + k) * 16 + r] = c[k];   // C[row = 4q + k][col = r]
}

__global__ void cvt_probe(const _Float16* in, float scale, unsigned* out) {
    h32 x;
    for (int i = 0; i < 32; i++) x[i] = in[threadIdx.x * 32 + i];
    const v6u r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, scale);
    for (int i = 0; i < 6; i++) out[threadIdx.x * 6 + i] = r[i];
}

__global__ void swap_probe(unsigned* out) {
    const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    const v2u s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const v2u s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 4 + 0] = s32[0];
    out[threadIdx.x * 4 + 1] = s32[1];
    out[threadIdx.x * 4 + 2] = s16[0];
    out[threadIdx.x * 4 + 3] = s16[1];
}

// ---- rates.  MODE 0: fp16 16x16x32; 1: FP8 x FP8 scaled 16x16x128; 2: FP6 x FP6 scaled; 3: the c6 mix per (row tile, 128 channels, 2 cout tiles):
// 8 fp16 + 4 FP6 MFMAs + one cvt_pk32; 4: the same without the conversion; 5: conversions only
template <int MODE>
__global__ __launch_bounds__(512, 2) void rate(float* out, int iters, unsigned long long* stamps) {
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
    v8i a8, b8;
    f16x8 ah[4], bh;
    h32 src;
    for (int d = 0; d < 8; d++) {
        a8[d] = 0x2a4b1c2d + threadIdx.x * 77 + d;
        b8[d] = 0x1b3a2c4d + d * 1234567 + threadIdx.x;
        bh[d] = (_Float16)(d * 0.01f + 0.1f);
        for (int c = 0; c < 4; c++) ah[c][d] = (_Float16)(threadIdx.x * 0.001f + c + d * 0.125f);
    }
    for (int i = 0; i < 32; i++) src[i] = ah[i >> 3][i & 7];
    const int sc = 0x7f7f7f7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    v6u x6 = {1, 2, 3, 4, 5, 6};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 3], bh, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, sc, 0, sc);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 2, 2, 0, sc, 0, sc);
        } else {
            if (MODE != 5) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i & 3], bh, acc[i & 1], 0, 0, 0);
            }
            if (MODE == 3 || MODE == 5) {
                asm volatile("" : "+v"(src));
                x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, 2.0f);
                asm volatile("" : "+v"(x6));
            }
            if (MODE != 5) {
                const v8i xb = {(int)x6[0], (int)x6[1], (int)x6[2], (int)x6[3], (int)x6[4], (int)x6[5], 0, 0};
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i & 1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, i < 2 ? xb : b8, acc[i & 1], 2, 2, 0, sc, 0, sc);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)x6[0];
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

template <int MODE>
static void run_rate(const char* name, float* dout, unsigned long long* dst, double mfma_cycles_16, int per_iter_items) {
    const int iters = 20000;
    for (int cfg = 0; cfg < 3; cfg++) {
        const int blocks = cfg == 0 ? 1 : 1024, threads = cfg == 0 ? 64 : cfg == 1 ? 256 : 512;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            rate<MODE><<<blocks, threads>>>(dout, iters, dst);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long st[2];
        hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
        const double seconds = st[1] / 100e6;
        printf("%-34s blocks %4d threads %3d: %7.1f shader cycles per iteration of one wave (%d items), clock %.0f MHz, %.3f ms\n", name, blocks, threads,
               (double)st[0] / iters, per_iter_items, st[0] / seconds * 1e-6, ms);
    }
    (void)mfma_cycles_16;
}

int main() {
    srand(7);
    // exact-integer friendly codes: values in {-2,-1,-0.5,0,0.5,1,2,3}
    std::vector<unsigned> good;
    for (unsigned c = 0; c < 64; c++) {
        const float v = e2m3_to_float(c);
        if (v == floorf(v * 2) / 2 && fabsf(v) <= 3 && !(c == 32)) good.push_back(c);
    }
    std::vector<unsigned char> A(16 * 128), B(128 * 16), SA(64), SB(64);
    for (auto& x : A) x = (unsigned char)good[rand() % good.size()];
    for (auto& x : B) x = (unsigned char)good[rand() % good.size()];
    unsigned char *dA, *dB, *dSA, *dSB;
    float* dC;
    hipMalloc(&dA, A.size());
    hipMalloc(&dB, B.size());
    hipMalloc(&dSA, 64);
    hipMalloc(&dSB, 64);
    hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int opsel = 0; opsel < 4; opsel++) {
        for (int i = 0; i < 64; i++) {
            SA[i] = (unsigned char)(127 + (rand() % 7) - 3);
            SB[i] = (unsigned char)(127 + (rand() % 7) - 3);
        }
        hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice);
        hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
        layout_probe<<<1, 64>>>(dA, dB, dSA, dSB, dC, opsel);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double want = 0;
                for (int k = 0; k < 128; k++)
                    want += (double)e2m3_to_float(A[i * 128 + k]) * ldexp(1.0, SA[i * 4 + k / 32] - 127) * e2m3_to_float(B[k * 16 + j]) * ldexp(1.0, SB[j * 4 + k / 32] - 127);
                if ((float)want != C[i * 16 + j]) bad++;
            }
        printf("FP6 layout + per-lane block scale probe (opsel %d): %d of 256 wrong\n", opsel, bad);
    }
    // ---- conversion
    {
        const float vals[32] = {0.f,   0.0624f, 0.0626f, 0.125f, 0.19f, 0.3f,  0.9f,  0.95f, 1.0f,  1.06f, 1.0625f, 1.07f, 1.1875f, 1.9f,  1.96f, 2.1f,
                                2.125f, 2.2f,   3.9f,    4.2f,   4.25f, 4.3f,  7.4f,  7.6f,  7.8f,  9.0f,  100.f,   -0.3f, -1.3f,   -7.9f, -100.f, 65504.f};
        std::vector<_Float16> h(64 * 32);
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 32; i++) h[l * 32 + i] = (_Float16)vals[(i + l) % 32];
        _Float16* din;
        unsigned* dout;
        hipMalloc(&din, h.size() * 2);
        hipMalloc(&dout, 64 * 24);
        hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (float scale : {1.0f, 4.0f, 0.25f}) {
            cvt_probe<<<1, 64>>>(din, scale, dout);
            std::vector<unsigned> o(64 * 6);
            hipMemcpy(o.data(), dout, 64 * 24, hipMemcpyDeviceToHost);
            printf("cvt_scalef32_pk32_fp6_f16, scale %g (lane 0: element e -> field e):\n", scale);
            for (int i = 0; i < 32; i++) {
                const int bit = 6 * i, w = bit >> 5, sh = bit & 31;
                unsigned long long two = o[w] | ((unsigned long long)(w + 1 < 6 ? o[w + 1] : 0) << 32);
                const unsigned code = (unsigned)(two >> sh) & 63;
                printf("  %10g -> %7g%s", (double)vals[i], e2m3_to_float(code), i % 4 == 3 ? "\n" : "");
            }
            // element order: lane l's element i is vals[(i + l) % 32]: check lane 5 the same way
            int bad = 0;
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 32; i++) {
                    const int bit = 6 * i, w = bit >> 5, sh = bit & 31;
                    unsigned long long two = o[l * 6 + w] | ((unsigned long long)(w + 1 < 6 ? o[l * 6 + w + 1] : 0) << 32);
                    const unsigned code = (unsigned)(two >> sh) & 63;
                    const int i0 = (i + l) % 32, b0 = 6 * i0, w0 = b0 >> 5, s0 = b0 & 31;
                    unsigned long long ref2 = o[w0] | ((unsigned long long)(w0 + 1 < 6 ? o[w0 + 1] : 0) << 32);
                    if (code != ((unsigned)(ref2 >> s0) & 63)) bad++;
                }
            printf("  element order consistent across lanes: %d mismatches\n", bad);
        }
    }
    // ---- swaps
    {
        unsigned* d;
        hipMalloc(&d, 64 * 16);
        swap_probe<<<1, 64>>>(d);
        std::vector<unsigned> o(256);
        hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        printf("permlane32_swap(a = 1000 + lane, b = 2000 + lane): lane 0 -> (%u, %u), lane 16 -> (%u, %u), lane 32 -> (%u, %u), lane 48 -> (%u, %u)\n", o[0], o[1],
               o[64], o[65], o[128], o[129], o[192], o[193]);
        printf("permlane16_swap(a, b):                            lane 0 -> (%u, %u), lane 16 -> (%u, %u), lane 32 -> (%u, %u), lane 48 -> (%u, %u)\n", o[2], o[3],
               o[66], o[67], o[130], o[131], o[194], o[195]);
    }
    // ---- rates
    float* dout2;
    hipMalloc(&dout2, 1024 * 512 * 4);
    unsigned long long* dst;
    hipMalloc(&dst, 16);
    run_rate<0>("fp16 16x16x32 x8", dout2, dst, 16, 8);
    run_rate<1>("FP8xFP8 scaled 16x16x128 x8", dout2, dst, 32, 8);
    run_rate<2>("FP6xFP6 scaled 16x16x128 x8", dout2, dst, 16, 8);
    run_rate<3>("c6 mix: 8 fp16 + cvt_pk32 + 4 FP6", dout2, dst, 0, 13);
    run_rate<4>("c6 mix without the conversion", dout2, dst, 0, 12);
    run_rate<5>("cvt_pk32_fp6_f16 alone", dout2, dst, 0, 1);
    return 0;
}
