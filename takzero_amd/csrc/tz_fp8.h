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
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7f);          // NaN
    if (a >= 0x43e80000u) return (uint8_t)(sign | 0x7e);         // >= 464 = halfway between 448 and the next step (and inf): saturate
    if (a < 0x3c800000u) {                                       // below 2^-6: multiples of 2^-9 (code 8 = the smallest normal)
        float v;
        memcpy(&v, &a, 4);
        v = v * 512.0f + 8388608.0f;                             // 2^23: the add rounds v * 2^9 to an integer, to nearest even
        uint32_t q;
        memcpy(&q, &v, 4);
        return (uint8_t)(sign | (q & 0x7fffffu));
    }
    a += 0x0007ffffu + ((a >> 20) & 1u);                         // round the mantissa to 3 bits, to nearest even; a carry bumps the exponent
    const uint32_t e = (a >> 23) - 120u, m = (a >> 20) & 7u;     // exponent field: unbiased + 7
    if (e > 15u || (e == 15u && m == 7u)) return (uint8_t)(sign | 0x7e);
    return (uint8_t)(sign | (e << 3) | m);
}

inline float tz_e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    const float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
