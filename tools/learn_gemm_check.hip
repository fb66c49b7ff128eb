// learn_gemm_check.hip — the trainer's stream-K GEMM (gemm_sk_kernel + gemm_fixup_kernel) against its one-workgroup-per-tile twin
// (gemm_f32_kernel) on the same operands, all four variants (plain / transposed A, stored / gathered im2col view), at the
// shapes a learn step launches and at shapes with few k-slabs per workgroup.  Same products, other summation order: the two
// must agree to fp32 rounding.  The kernels live in an anonymous namespace, so this file includes the unit itself.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/learn_gemm_check.hip -Ltakzero_amd -ltakzero_hip
//         -Wl,-rpath,$PWD/takzero_amd -o tools/bin/learn_gemm_check && tools/bin/learn_gemm_check
#include "../takzero_amd/csrc/tz_learn.hip"

#include <cstdio>
#include <random>

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

template <bool AT, bool GATHER>
static double one(int M, int N, int K, int gn, int gc, int G, bool accumulate, bool with_bias) {
    // stored A: [M][K] (or [K][M] for AT).  gathered: an NHWC tensor of `pixels` rows x gc channels, pixels = M (or K for AT)
    const int pixels = AT ? K : M;
    const size_t a_elems = GATHER ? (size_t)(pixels + 64) * gc : (size_t)M * K;
    std::mt19937 rng(M * 31 + N * 7 + K + G);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> hA(a_elems), hB((size_t)K * N), hC((size_t)M * N), hbias(N);
    for (auto& v : hA) v = u(rng);
    for (auto& v : hB) v = u(rng);
    for (auto& v : hC) v = u(rng);
    for (auto& v : hbias) v = u(rng);
    float *A, *B, *C0, *C1, *bias, *ws;
    CK(hipMalloc(&A, a_elems * 4));
    CK(hipMalloc(&B, hB.size() * 4));
    CK(hipMalloc(&C0, hC.size() * 4));
    CK(hipMalloc(&C1, hC.size() * 4));
    CK(hipMalloc(&bias, N * 4));
    CK(hipMalloc(&ws, (size_t)G * 2 * 4096 * 4));
    CK(hipMemcpy(A, hA.data(), a_elems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(C0, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(C1, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ws, 0xff, (size_t)G * 2 * 4096 * 4));   // NaN: a slot read without having been written shows
    const int lda = GATHER ? gc : (AT ? M : K), ldb = N, ldc = N, acc = accumulate ? 1 : 0;
    const float* bp = with_bias ? bias : nullptr;
    gemm_f32_kernel<AT, GATHER><<<dim3(M / 64, N / 64), 512>>>(A, B, C0, bp, K, lda, ldb, ldc, acc, gn, gc);
    const int tiles_m = M / 64, tiles = tiles_m * (N / 64), S = K / 32;
    const long long total = (long long)tiles * S;
    const int g = (int)std::min<long long>(total, G);
    gemm_sk_kernel<AT, GATHER><<<g, 256>>>(A, B, C1, bp, K, lda, ldb, ldc, acc, gn, gc, tiles_m, total, ws);
    if (total % g || (total / g) % S) gemm_fixup_kernel<<<tiles, 256>>>(C1, bp, S, ldc, acc, tiles_m, (unsigned)total, (unsigned)g, ws);
    CK(hipDeviceSynchronize());
    std::vector<float> r0(hC.size()), r1(hC.size());
    CK(hipMemcpy(r0.data(), C0, r0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), C1, r1.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < r0.size(); i++) {
        const double d = fabs((double)r0[i] - r1[i]);
        if (!(d <= 1e30)) worst = 1e30;
        worst = d > worst ? d : worst;
        scale = fabs(r0[i]) > scale ? fabs(r0[i]) : scale;
    }
    printf("AT %d GATHER %d  M %5d N %4d K %5d gc %3d G %3d acc %d bias %d: max |diff| %.3g (max |C| %.3g)%s\n", AT, GATHER, M, N, K, gc, g, acc,
           with_bias, worst, scale, worst <= 2e-5 * scale ? "" : "   <-- MISMATCH");
    (void)hipFree(A);
    (void)hipFree(B);
    (void)hipFree(C0);
    (void)hipFree(C1);
    (void)hipFree(bias);
    (void)hipFree(ws);
    return worst <= 2e-5 * scale ? 0.0 : 1.0;
}

int main() {
    double bad = 0;
    for (int G : {512, 96, 7}) {
        // batch 128 on 5x5: forward / data gradient of a tower conv, weight gradient of a tower conv and of the input conv
        bad += one<false, true>(3200, 256, 2304, 5, 256, G, false, true);
        bad += one<true, true>(2304, 256, 3200, 5, 256, G, false, false);
        bad += one<true, true>(320, 256, 3200, 5, 32, G, true, false);
        bad += one<false, true>(3200, 256, 288, 5, 32, G, false, true);
        // the test's batch of 8 (200 pixels in 256 rows) and plain matrices
        bad += one<false, true>(256, 256, 2304, 5, 256, G, true, true);
        bad += one<true, true>(2304, 256, 256, 5, 256, G, false, false);
        bad += one<false, false>(128, 256, 1600, 0, 0, G, false, true);
        bad += one<true, false>(1600, 128, 128, 0, 0, G, true, false);
        bad += one<false, true>(2304, 256, 2304, 6, 256, G, false, false);   // 6x6: 64 boards
        // the parity test's step (batch 64 on 5x5, 128 policy channels): policy conv forward / data gradient / weight gradient, tower and input weight gradients
        bad += one<false, true>(1600, 128, 2304, 5, 256, G, false, true);
        bad += one<false, true>(1600, 256, 1152, 5, 128, G, true, false);
        bad += one<true, true>(2304, 128, 1600, 5, 256, G, false, false);
        bad += one<true, true>(2304, 256, 1600, 5, 256, G, false, false);
        bad += one<true, true>(320, 256, 1600, 5, 32, G, false, false);
        bad += one<false, true>(1600, 256, 320, 5, 32, G, false, false);
    }
    printf(bad ? "FAILED\n" : "all equal to rounding\n");
    return bad ? 1 : 0;
}
