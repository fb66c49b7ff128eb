// tz_fp8.h — host-side conversion to OCP FP8 E4M3 (e4m3fn: no infinities, 0x7f / 0xff = NaN, largest finite 448), the format
// gfx950's v_cvt_pk_fp8_f32 writes and v_mfma_f32_16x16x128_f8f6f4 reads with FMT 0.  Round to nearest even, saturating:
// the weight side of TZ_PREC_F16C8 is converted here, the activation side in the kernel's epilogue with the hardware
// instruction (clamped to +-448 first: the instruction turns overflow into NaN).
// tests/test_fp8_host.py checks every code point and a few hundred thousand random values against torch.float8_e4m3fn.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

inline uint8_t tz_f32_to_e4m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    const float a = fabsf(f);
    if (!(a == a)) return (uint8_t)(sign | 0x7f);
    if (a >= 464.0f) return (uint8_t)(sign | 0x7e);   // 464 = halfway between 448 and the next step: everything above saturates
    if (a < ldexpf(1.0f, -6)) {                       // subnormals: multiples of 2^-9 (and the smallest normal, code 8)
        const int qs = (int)nearbyintf(ldexpf(a, 9)); // default rounding mode: to nearest even
        return (uint8_t)(sign | qs);
    }
    int e;
    (void)frexpf(a, &e);                              // a = m * 2^e, m in [0.5, 1)  ->  a = 1.x * 2^(e-1)
    int E = e - 1;
    int q = (int)nearbyintf(ldexpf(a, 3 - E));        // 8 .. 16
    if (q == 16) {
        q = 8;
        E++;
    }
    if (E > 8 || (E == 8 && q > 14)) return (uint8_t)(sign | 0x7e);
    return (uint8_t)(sign | ((E + 7) << 3) | (q - 8));
}

inline float tz_e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    const float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
