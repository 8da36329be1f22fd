// Elementary functions the WGSL shader leaves to the implementation (pow, exp), fixed to ONE
// recipe made of plain IEEE binary32 + - * / (no fma, no libm), so host tables and device code
// agree bit for bit (DESIGN.md "Elementary functions").  Compiled for host and for gfx950 with
// -ffp-contract=off.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VOLYM_HD __host__ __device__ inline
#else
#define VOLYM_HD inline
#endif

namespace volym {

VOLYM_HD uint32_t f32_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
VOLYM_HD float bits_f32(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// log2(x), x positive finite normal.  x = m * 2^e with m in (sqrt(1/2), sqrt(2)],
// s = (m-1)/(m+1), log2(m) = (2/ln2)(s + s^3/3 + s^5/5 + s^7/7 + s^9/9), Horner in s^2.
VOLYM_HD float wgsl_log2(float x)
{
    const uint32_t bits = f32_bits(x);
    int e = static_cast<int>(bits >> 23) - 127;
    float m = bits_f32((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 0x1.6a09e6p+0f) {   // 1.41421356f
        m = m * 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = 0x1.484b14p-2f;          // 2/(9 ln2)  0.3205989
    p = p * s2 + 0x1.a61762p-2f;       // 2/(7 ln2)  0.412198573
    p = p * s2 + 0x1.2776c6p-1f;       // 2/(5 ln2)  0.577078044
    p = p * s2 + 0x1.ec709ep-1f;       // 2/(3 ln2)  0.961796701
    p = p * s2 + 0x1.715476p+1f;       // 2/ln2      2.88539004
    return static_cast<float>(e) + s * p;
}

// 2^z: n = rint(z) (half to even), f = z - n, degree-7 Taylor of 2^f in Horner form, times 2^n.
VOLYM_HD float wgsl_exp2(float z)
{
    if (!(z >= -126.0f)) return 0.0f;
    if (z > 127.0f) z = 127.0f;
    const float n = __builtin_rintf(z);
    const float f = z - n;
    float p = 0x1.ffcbfcp-17f;         // ln2^7/5040
    p = p * f + 0x1.430912p-13f;       // ln2^6/720
    p = p * f + 0x1.5d87fep-10f;       // ln2^5/120
    p = p * f + 0x1.3b2ab6p-7f;        // ln2^4/24
    p = p * f + 0x1.c6b08ep-5f;        // ln2^3/6
    p = p * f + 0x1.ebfbep-3f;         // ln2^2/2
    p = p * f + 0x1.62e43p-1f;         // ln2
    p = p * f + 1.0f;
    const float scale = bits_f32(static_cast<uint32_t>(static_cast<int>(n) + 127) << 23);
    return p * scale;
}

VOLYM_HD float wgsl_pow(float x, float y)
{
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;
    return wgsl_exp2(y * wgsl_log2(x));
}

VOLYM_HD float wgsl_exp(float x) { return wgsl_exp2(x * 0x1.715476p+0f); }   // 1/ln2

}  // namespace volym
