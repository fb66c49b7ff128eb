#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned char* A, const unsigned char* B, float* C, int noscale) {
    // hypothesis: lane l holds A[row l&15][k = 32*(l>>4) + i], B[k = 32*(l>>4)+i][col l&15], i = byte 0..31
    const int lane = threadIdx.x;
    v8i a, b;
    const int r = lane & 15, q = lane >> 4;
    for (int d = 0; d < 8; d++) {
        uint32_t wa = 0, wb = 0;
        for (int e = 0; e < 4; e++) {
            const int k = 32 * q + d * 4 + e;
            wa |= (uint32_t)A[r * 128 + k] << (8 * e);
            wb |= (uint32_t)B[k * 16 + r] << (8 * e);
        }
        a[d] = wa; b[d] = wb;
    }
    f32x4 c = {0, 0, 0, 0};
    if (noscale) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int k = 0; k < 4; k++) C[(q * 4 + k) * 16 + r] = c[k];
}

// fp8 conversion probe: pack 4 floats to a dword and back
__global__ void cvt_probe(const float* in, uint32_t* out, float* back) {
    const int i = threadIdx.x;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i * 4 + 0], in[i * 4 + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i * 4 + 2], in[i * 4 + 3], w, true);
    out[i] = (uint32_t)w;
    back[i * 4 + 0] = __builtin_amdgcn_cvt_f32_fp8(w, 0);
    back[i * 4 + 1] = __builtin_amdgcn_cvt_f32_fp8(w, 1);
    back[i * 4 + 2] = __builtin_amdgcn_cvt_f32_fp8(w, 2);
    back[i * 4 + 3] = __builtin_amdgcn_cvt_f32_fp8(w, 3);
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, unsigned long long* stamps = nullptr) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
    v8i a8, b8;
    f16x8 ah, bh;
    for (int d = 0; d < 8; d++) { a8[d] = 0x38383838 + threadIdx.x; b8[d] = 0x38383838 + d; ah[d] = (_Float16)(threadIdx.x * 0.001f); bh[d] = (_Float16)(d * 0.01f); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(ah), "v"(bh));
            else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a8), "v"(b8));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static float e4m3_to_float(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    std::vector<unsigned char> A(16 * 128), B(128 * 16);
    srand(5);
    // e4m3 codes of small integers: build table of exact ints -4..4
    unsigned char codes[9];
    for (int v = -4; v <= 4; v++) {
        for (int c = 0; c < 256; c++) if (c != 0x7f && c != 0xff && e4m3_to_float((unsigned char)c) == (float)v && !(v == 0 && c == 0x80)) { codes[v + 4] = (unsigned char)c; break; }
    }
    for (auto& x : A) x = codes[rand() % 9];
    for (auto& x : B) x = codes[rand() % 9];
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int noscale = 0; noscale < 2; noscale++) {
        probe<<<1, 64>>>(dA, dB, dC, noscale);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            float want = 0;
            for (int k = 0; k < 128; k++) want += e4m3_to_float(A[i * 128 + k]) * e4m3_to_float(B[k * 16 + j]);
            if (want != C[i * 16 + j]) bad++;
        }
        printf("layout probe (noscale=%d): %d of 256 wrong\n", noscale, bad);
    }
    // conversions
    float hin[16] = {0.1f, 1.3f, 300.f, 500.f, 1e-3f, -0.3f, 448.f, 1000.f, 0.0019f, 0.001f, -7.7f, 17.f, 0.f, 1e6f, -1e6f, 0.06f};
    float* din; uint32_t* dout; float* dback;
    hipMalloc(&din, 64); hipMalloc(&dout, 16); hipMalloc(&dback, 64);
    hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
    cvt_probe<<<1, 4>>>(din, dout, dback);
    float hb[16];
    hipMemcpy(hb, dback, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; i++) printf("cvt %g -> %g\n", hin[i], hb[i]);
    // rates
    float* dout2; hipMalloc(&dout2, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 1024;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<blocks, 256>>>(dout2, iters);
            else if (mode == 1) rate<1><<<blocks, 256>>>(dout2, iters);
            else rate<2><<<blocks, 256>>>(dout2, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)blocks * 4 * iters * 8;
            const double flops = mf * 2.0 * 16 * 16 * (mode == 0 ? 32 : 128);
            if (rep) printf("mode %d: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per wave-slot\n", mode, ms, flops / ms * 1e-9, ms * 1e6 / (iters * 8.0 * (blocks * 4 / 1024.0)));
        }
    }
    unsigned long long* dst; hipMalloc(&dst, 16);
    for (int blocks2 : {1, 256, 1024}) for (int threads : {64, 256}) for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) rate<0><<<blocks2, threads>>>(dout2, iters, dst);
            else rate<1><<<blocks2, threads>>>(dout2, iters, dst);
            hipDeviceSynchronize();
        }
        unsigned long long st[2];
        hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
        const double seconds = st[1] / 100e6;
        printf("blocks %4d threads %3d mode %d: %.1f shader cycles per MFMA of one wave (memtime), clock %.0f MHz\n", blocks2, threads, mode,
               (double)st[0] / (iters * 8.0), st[0] / seconds * 1e-6);
    }
    return 0;
}
