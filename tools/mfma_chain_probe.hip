// Does it cost MFMA issue time that a row tile's twelve MFMAs go into only two accumulators (TZ_PREC_F16C6's k-loop), and that the
// FP6-scaled and the fp16 form alternate in such a chain?  One workgroup of 512 threads per CU (two waves per SIMD), operands in
// registers.   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip -o /tmp/chain && /tmp/chain
//   mode 0: 12 independent accumulators, order F H H H H F per pair
//   mode 1: 2 accumulators (j = 0, 1 alternating), per accumulator F H H H H F      (the k-loop's order)
//   mode 2: 2 accumulators, per accumulator F F H H H H
//   mode 3: 2 accumulators, fp16 only (6 per accumulator)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void chain(float* out, int iters, unsigned long long* stamps) {
    f32x4 acc[12];
    for (int i = 0; i < 12; i++) acc[i] = f32x4{0, 0, 0, 0};
    v8i a6, b6;
    f16x8 ah, bh;
    for (int d = 0; d < 8; d++) {
        a6[d] = 0x2a4b1c2d + threadIdx.x * 77 + d;
        b6[d] = 0x1b3a2c4d + d * 1234567 + threadIdx.x;
        ah[d] = (_Float16)(threadIdx.x * 0.001f + d * 0.125f);
        bh[d] = (_Float16)(d * 0.01f + 0.1f);
    }
    const int sc = 0x7f7f7f7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        // the twelve MFMAs of a step: positions 0..5 of the two chains, j alternating
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ai = MODE == 0 ? p * 2 + j : j;
                const bool f6 = MODE == 3 ? false : MODE == 2 ? p < 2 : (p == 0 || p == 5);
                if (f6) acc[ai] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, acc[ai], 2, 2, 0, sc, 0, sc);
                else acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[ai], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 12; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, float* dout, unsigned long long* dst) {
    const int iters = 20000;
    for (int threads : {64, 512}) {
        for (int rep = 0; rep < 2; rep++) {
            chain<MODE><<<256, threads>>>(dout, iters, dst);
            (void)hipDeviceSynchronize();
        }
        unsigned long long st[2];
        (void)hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
        printf("%-44s %3d threads per CU: %6.1f cycles per 12-MFMA step of one wave, clock %.0f MHz\n", name, threads, (double)st[0] / iters,
               st[0] / (st[1] / 100e6) * 1e-6);
    }
}

int main() {
    float* dout;
    unsigned long long* dst;
    (void)hipMalloc(&dout, 256 * 512 * 4);
    (void)hipMalloc(&dst, 16);
    run<0>("12 accumulators, F H H H H F", dout, dst);
    run<1>("2 accumulators, F H H H H F (k-loop order)", dout, dst);
    run<2>("2 accumulators, F F H H H H", dout, dst);
    run<3>("2 accumulators, fp16 only", dout, dst);
    return 0;
}
