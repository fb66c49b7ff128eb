// tz_fp6.h — host-side conversion to OCP MX FP6 E2M3 (1 sign, 2 exponent, 3 mantissa bits, bias 1: 0, 0.125 .. 0.875, then
// 1 .. 7.5; no infinities, no NaN), the format v_mfma_scale_f32_16x16x128_f8f6f4 reads with FMT 2 and
// v_cvt_scalef32_pk32_fp6_f16 writes (tools/mfma_f6_probe.hip: round to nearest even, saturating at 7.5, the scale divides).
// The weight side of TZ_PREC_F16C6 is converted here, block of 32 input channels by block, each block with one power-of-two
// scale (an E8M0 byte, 2^(byte - 127)); the activation side in the kernel's epilogue with the hardware instruction.
// tests/test_fp6_host.py checks every code point, every rounding tie and the block scale rule against a numpy restatement.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

// x in units of the block scale -> 6-bit code
inline uint8_t tz_f32_to_e2m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 26) & 0x20);
    float a = fabsf(f);
    if (!(a == a)) return (uint8_t)(sign | 0x1f);       // NaN has no code: the largest magnitude
    if (a >= 7.75f) return (uint8_t)(sign | 0x1f);      // the tie between 7.5 and 8 goes to the even 8, which saturates
    // steps of 1/8 below 2, 1/4 below 4, 1/2 below 8; the add of 1.5 * 2^23-scaled constant rounds to nearest even
    const float step = a < 2.0f ? 0.125f : a < 4.0f ? 0.25f : 0.5f;
    float q = a / step + 12582912.0f;                   // 1.5 * 2^23: the sum's ulp is 1, so the add rounds a / step to an integer (ties to even)
    q -= 12582912.0f;
    const float r = q * step;                           // may have reached the next binade (e.g. 1.97 -> 2.0): recode from the value
    if (r >= 8.0f) return (uint8_t)(sign | 0x1f);
    uint8_t code;
    if (r < 1.0f) code = (uint8_t)(r * 8.0f);                         // subnormals and zero: m / 8
    else if (r < 2.0f) code = (uint8_t)(0x08 | (int)((r - 1.0f) * 8.0f));
    else if (r < 4.0f) code = (uint8_t)(0x10 | (int)((r - 2.0f) * 4.0f));
    else code = (uint8_t)(0x18 | (int)((r - 4.0f) * 2.0f));
    return (uint8_t)(sign | code);
}

inline float tz_e2m3_to_f32(uint8_t c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float f = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -f : f;
}

// Biased exponent (the E8M0 byte) of the block scale s for a block whose largest magnitude is amax: the power of two with
// amax / s in [3.75, 7.5), i.e. s = 2^(floor(log2(amax * 16/15)) - 2), computed the way the kernel's epilogue computes it (one
// fp32 multiply, then the exponent field); never below `min_byte` (a block of zeros, or of values too small to matter).
inline uint32_t tz_e2m3_block_scale_byte(float amax, uint32_t min_byte = 1) {
    const float t = amax * (16.0f / 15.0f);
    uint32_t u;
    memcpy(&u, &t, 4);
    const uint32_t b = (u >> 23) & 0xffu;
    const uint32_t lo = min_byte + 2u;
    return (b > lo ? b : lo) - 2u;
}

// 32 codes -> the 24-byte operand string of one lane: element i in bits [6 i, 6 i + 6)
inline void tz_pack_fp6x32(const uint8_t* codes, uint32_t* out6) {
    for (int w = 0; w < 6; w++) out6[w] = 0;
    for (int i = 0; i < 32; i++) {
        const uint64_t c = codes[i] & 63u;
        const int bit = 6 * i, w = bit >> 5, sh = bit & 31;
        out6[w] |= (uint32_t)(c << sh);
        if (sh > 26) out6[w + 1] |= (uint32_t)(c >> (32 - sh));
    }
}
