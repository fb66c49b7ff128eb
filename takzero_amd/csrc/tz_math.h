// tz_math.h — deterministic f32 exp / ln / powi shared by the device tree kernels and the CPU
// oracle, so that PUCT scores, softmax priors and Dirichlet logits are bit-identical on both
// sides.  The reference calls Rust's f32::{exp, ln, powi} (policy.rs:10-19,140-156,
// noise.rs:23, eval.rs:97); exp/ln resolve to the host libm, whose last-bit behaviour is
// platform dependent, so there is no single "reference bit pattern" to match.  These
// versions evaluate in IEEE double with only + - * / (no FMA contraction: compile with
// -ffp-contract=off) and round once to float, which is what glibc's expf/logf do as well;
// tests/test_math.py measures the agreement with libm.
//
// powi follows compiler-builtins' __powisf2 (square-and-multiply in f32), which is what
// f32::powi lowers to on x86-64.
#ifndef TZ_MATH_H
#define TZ_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TZ_HD __host__ __device__ __forceinline__
#else
#define TZ_HD static inline
#endif

#if defined(__clang__)
#define TZ_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define TZ_NO_CONTRACT
#endif

TZ_HD double tz_bits_to_double(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}
TZ_HD uint64_t tz_double_to_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof u);
    return u;
}
TZ_HD float tz_bits_to_float(uint32_t u) {
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}
TZ_HD uint32_t tz_float_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}

// e^x for finite float x, result rounded from a double evaluation (error << 1e-15 relative
// before the final rounding).
TZ_HD float tz_expf(float xf) {
    TZ_NO_CONTRACT
    double x = (double)xf;
    if (!(x == x)) return xf;  // NaN
    if (x > 88.9) return tz_bits_to_float(0x7f800000u);
    if (x < -104.5) return 0.0f;
    double t = x * 1.4426950408889634074;  // log2(e)
    double kd = __builtin_floor(t + 0.5);
    // ln2 split so that kd*ln2_hi is exact for |kd| < 2^11
    double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    // Taylor series of e^r, |r| <= 0.347, degree 14 (remainder < 3e-19)
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int64_t k = (int64_t)kd;  // in [-151, 129]
    double scale = tz_bits_to_double((uint64_t)(k + 1023) << 52);
    return (float)(p * scale);
}

// ln(x) for float x.  x<0 -> NaN, x==0 -> -inf, inf -> inf.
TZ_HD float tz_logf(float xf) {
    TZ_NO_CONTRACT
    if (!(xf == xf)) return xf;
    if (xf < 0.0f) return tz_bits_to_float(0x7fc00000u);
    if (xf == 0.0f) return tz_bits_to_float(0xff800000u);
    uint32_t fb = tz_float_to_bits(xf);
    if (fb == 0x7f800000u) return xf;
    double x = (double)xf;  // exact, normal in double even for float denormals
    uint64_t b = tz_double_to_bits(x);
    int64_t e = (int64_t)((b >> 52) & 0x7ff) - 1023;
    double m = tz_bits_to_double((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);  // [1,2)
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e += 1;
    }
    double s = (m - 1.0) / (m + 1.0);  // |s| <= 0.1716
    double z = s * s;
    // atanh series: ln(m) = 2 s (1 + z/3 + z^2/5 + ... ), z <= 0.02944; 13 terms -> < 1e-21
    double q = 1.0 / 27.0;
    q = q * z + 1.0 / 25.0;
    q = q * z + 1.0 / 23.0;
    q = q * z + 1.0 / 21.0;
    q = q * z + 1.0 / 19.0;
    q = q * z + 1.0 / 17.0;
    q = q * z + 1.0 / 15.0;
    q = q * z + 1.0 / 13.0;
    q = q * z + 1.0 / 11.0;
    q = q * z + 1.0 / 9.0;
    q = q * z + 1.0 / 7.0;
    q = q * z + 1.0 / 5.0;
    q = q * z + 1.0 / 3.0;
    q = q * z + 1.0;
    double lm = 2.0 * s * q;
    double ed = (double)e;
    double res = (ed * 6.93147180369123816490e-01 + lm) + ed * 1.90821492927058770002e-10;
    return (float)res;
}

// f32::powi  (compiler-builtins __powisf2)
TZ_HD float tz_powif(float a, int b) {
    TZ_NO_CONTRACT
    unsigned int pw = b < 0 ? 0u - (unsigned int)b : (unsigned int)b;
    float mul = 1.0f;
    for (;;) {
        if (pw & 1u) mul = mul * a;
        pw >>= 1;
        if (pw == 0) break;
        a = a * a;
    }
    return b < 0 ? 1.0f / mul : mul;
}

#define TZ_DISCOUNT 0.997f  // search/mod.rs:7

#endif  // TZ_MATH_H
