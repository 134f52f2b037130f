/*
 * oracle/orc_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Scalar fp32 math the oracle uses wherever the reference's GLSL calls a built-in
 * (log, sin, cos, acos, asin, atan -- data/shader/include/path_trace.glsl:36,163;
 * dir_gen.glsl:11-12,49; path_trace.glsl:83; nrc/prep_infer_rays.comp:13-15).
 *
 * GLSL built-ins are implementation defined (no bit-exact spec), so this build *defines* them:
 * Cephes-style single-precision polynomials (log: a 128-bin table reduction and a cubic, see orc_logf) whose Horner steps are explicit single-rounding
 * fused multiply-adds (fmaf on the host == v_fma_f32 on the device), correctly rounded / and sqrt.  The HIP product carries its own statement of
 * the same polynomials (nrc-hpm-renderer_amd/csrc/nrc_math.h); tests/test_gpu_math.py checks
 * the two bit-for-bit on the GPU.  Compile with -ffp-contract=off.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

#define ORC_PI 3.14159274101257324f      /* float(PI) of nrc-constants.glsl:33 */
#define ORC_TWO_PI 6.28318548202514648f  /* float(2.0*PI) */
#define ORC_HALF_PI 1.57079637050628662f
#define ORC_QUARTER_PI 0.785398185253143311f


/* single-rounding multiply-add: fmaf on the host, v_fma_f32 on the device -- both IEEE, hence bit-identical */
static inline float orc_fmaf_(float a, float b, float c) { return fmaf(a, b, c); }

static inline uint32_t orc_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float orc_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* natural log for normal positive x (call sites pass 1-u, u in [0,1) => [2^-23, 1]).
 * Spec (round 5): x = m * 2^e with m in [0.5, 1); bin i = top seven mantissa bits; {inv_c, log_c} = ORC_LOG_TAB[i] (tools/make_log_table.py:
 * inv_c = RN32(1/c_i), log_c = RN32(-ln(inv_c)), c the bin's centre, the edges for the first and last bin);
 *   r = fma(m, inv_c, -1);  q = fma(fma(r, 1/3, -1/2), r, 1);  log(x) = fma(q, r, fma(e, LN2, log_c)).
 * e * LN2 and log(m) are both <= 0 on (0, 1): no cancellation; log(1) = fma(1, LN2, -LN2) + 0 = 0 exactly.  At most 1 ulp from the correctly
 * rounded value on all 2^23 arguments the integrator can pass (tests/test_oracle_math.py, exhaustive); GLSL asks for 3 ulp / 2^-21. */
static const float ORC_LOG_TAB[128][2] = {
#include "orc_log_table.inc"
};
#define ORC_LN2 0x1.62e430p-1f
#define ORC_THIRD 0x1.555556p-2f
static inline float orc_logf(float x)
{
    uint32_t ix = orc_f2u(x);
    int e = (int)((ix >> 23) & 0xffu) - 126;
    float m = orc_u2f((ix & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    const float* t = ORC_LOG_TAB[(ix >> 16) & 127u];
    float r = orc_fmaf_(m, t[0], -1.0f);
    float q = orc_fmaf_(r, ORC_THIRD, -0.5f);
    q = orc_fmaf_(q, r, 1.0f);
    float s = orc_fmaf_((float)e, ORC_LN2, t[1]);
    return orc_fmaf_(q, r, s);
}

/* sin and cos of x (radians), |x| up to a few thousand */
static inline void orc_sincosf(float x, float* s_out, float* c_out)
{
    float ax = fabsf(x);
    uint32_t j = (uint32_t)(ax * 1.27323949337005615f); /* 4/pi */
    j = (j + 1u) & ~1u;
    float y = (float)j;
    float r = orc_fmaf_(-y, 0.78515625f, ax);
    r = orc_fmaf_(-y, 2.4187564849853515625e-4f, r);
    r = orc_fmaf_(-y, 3.77489497744594108e-8f, r);
    float z = r * r;
    float ps = -1.9515295891E-4f;
    ps = orc_fmaf_(ps, z, 8.3321608736E-3f);
    ps = orc_fmaf_(ps, z, -1.6666654611E-1f);
    ps = ps * z;
    ps = orc_fmaf_(ps, r, r);
    float pc = 2.443315711809948E-005f;
    pc = orc_fmaf_(pc, z, -1.388731625493765E-003f);
    pc = orc_fmaf_(pc, z, 4.166664568298827E-002f);
    pc = pc * z;
    pc = pc * z;
    pc = orc_fmaf_(-0.5f, z, pc);
    pc = pc + 1.0f;
    uint32_t q = (j >> 1) & 3u;
    float s, c;
    if (q == 0u) { s = ps; c = pc; }
    else if (q == 1u) { s = pc; c = -ps; }
    else if (q == 2u) { s = -ps; c = -pc; }
    else { s = -pc; c = ps; }
    if (x < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}

/* asin for |x| <= 1; NaN outside */
static inline float orc_asinf(float x)
{
    float a = fabsf(x);
    if (!(a <= 1.0f)) return orc_u2f(0x7fc00000u);
    float z, w;
    int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
    else { w = a; z = a * a; }
    float p = 4.2163199048E-2f;
    p = orc_fmaf_(p, z, 2.4181311049E-2f);
    p = orc_fmaf_(p, z, 4.5470025998E-2f);
    p = orc_fmaf_(p, z, 7.4953002686E-2f);
    p = orc_fmaf_(p, z, 1.6666752422E-1f);
    p = p * z;
    p = orc_fmaf_(p, w, w);
    if (big) { p = p + p; p = ORC_HALF_PI - p; }
    return x < 0.0f ? -p : p;
}

/* acos; NaN for |x| > 1 (GLSL: undefined; this is what makes quirk Q5 visible) */
static inline float orc_acosf(float x)
{
    if (!(fabsf(x) <= 1.0f)) return orc_u2f(0x7fc00000u);
    if (x < -0.5f) return ORC_PI - 2.0f * orc_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * orc_asinf(sqrtf(0.5f * (1.0f - x)));
    return ORC_HALF_PI - orc_asinf(x);
}

/* acos with the argument clamped to [-1,1] (dir_gen.glsl:49: cosTheta reaches +-1 up to rounding) */
static inline float orc_acosf_clamped(float x)
{
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    return orc_acosf(x);
}

static inline float orc_atanf(float x)
{
    float sgn = 1.0f;
    if (x < 0.0f) { sgn = -1.0f; x = -x; }
    float y;
    if (x > 2.41421365737915039f) { y = ORC_HALF_PI; x = -(1.0f / x); }
    else if (x > 0.414213567972183228f) { y = ORC_QUARTER_PI; x = (x - 1.0f) / (x + 1.0f); }
    else { y = 0.0f; }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = orc_fmaf_(p, z, -1.38776856032E-1f);
    p = orc_fmaf_(p, z, 1.99777106478E-1f);
    p = orc_fmaf_(p, z, -3.33329491539E-1f);
    p = p * z;
    p = orc_fmaf_(p, x, x);
    y = y + p;
    return sgn * y;
}

/* GLSL atan(y, x) */
static inline float orc_atan2f(float y, float x)
{
    if (x == 0.0f) {
        if (y > 0.0f) return ORC_HALF_PI;
        if (y < 0.0f) return -ORC_HALF_PI;
        return 0.0f;
    }
    float a = orc_atanf(y / x);
    if (x < 0.0f) a = (y >= 0.0f) ? a + ORC_PI : a - ORC_PI;
    return a;
}

/* IEEE binary16 <-> binary32, round-to-nearest-even (what a plain float->half cast does) */
static inline uint16_t orc_f32_to_f16(float f)
{
    uint32_t x = orc_f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
    if (ax < 0x33000001u) return (uint16_t)sign;               /* rounds to zero (<= 2^-25) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    uint32_t h;
    if (e < -14) {
        int shift = -14 - e + 13;              /* denormal half: shift in [14,24] */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q += 1u;
        h = q;
    } else {
        uint32_t q = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
        uint32_t rem = m & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q += 1u;
        h = q;
    }
    return (uint16_t)(sign | h);
}

static inline float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0u) {
        if (m == 0u) return orc_u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f; /* 2^-24 */
        return sign ? -v : v;
    }
    if (e == 31u) return orc_u2f(sign | 0x7f800000u | (m << 13));
    return orc_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

static inline float orc_round_f16(float f) { return orc_f16_to_f32(orc_f32_to_f16(f)); }

#endif
