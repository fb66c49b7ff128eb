// semantics of gfx950's scaled FP8 conversions (does the scale multiply or divide, does overflow saturate?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, float scale, float* out32, float* out16) {
    const int i = threadIdx.x;
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, in[2 * i], in[2 * i + 1], scale, false);
    const int wi = __builtin_bit_cast(int, w);
    out32[2 * i] = __builtin_amdgcn_cvt_f32_fp8(wi, 0);
    out32[2 * i + 1] = __builtin_amdgcn_cvt_f32_fp8(wi, 1);
    s16x2 v = {0, 0};
    f16x2 h = {(_Float16)in[2 * i], (_Float16)in[2 * i + 1]};
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(v, h, scale, true);
    const int vi = __builtin_bit_cast(int, v);
    out16[2 * i] = __builtin_amdgcn_cvt_f32_fp8(vi, 2);
    out16[2 * i + 1] = __builtin_amdgcn_cvt_f32_fp8(vi, 3);
}
int main() {
    float hin[16] = {1.0f, 1.3f, 100.f, 500.f, 1e-3f, -0.3f, 448.f, 1000.f, 0.0019f, 60000.f, -7.7f, 17.f, 0.f, 1e6f, -1e6f, 0.06f};
    float *din, *d32, *d16;
    hipMalloc(&din, 64); hipMalloc(&d32, 64); hipMalloc(&d16, 64);
    hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 4.0f, 0.25f, 3.0f, 8192.0f}) {
        probe<<<1, 8>>>(din, scale, d32, d16);
        float a[16], b[16];
        hipMemcpy(a, d32, 64, hipMemcpyDeviceToHost);
        hipMemcpy(b, d16, 64, hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int i = 0; i < 16; i++) printf("  %12g -> from f32 %12g   from f16 %12g\n", hin[i], a[i], b[i]);
    }
    return 0;
}
