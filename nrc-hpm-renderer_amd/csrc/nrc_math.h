/*
 * nrc-hpm-renderer_amd/csrc/nrc_math.h -- fp32 math spec of the integrator (host + gfx950 device).
 *
 * The reference calls GLSL built-ins (log, sin, cos, acos, asin, atan: data/shader/include/path_trace.glsl:36,163,
 * dir_gen.glsl:11-12,49, path_trace.glsl:83, nrc/prep_infer_rays.comp:13-15) whose results are implementation
 * defined.  To make per-pixel control flow reproducible between the CPU oracle and the HIP kernels this build
 * defines them: Cephes-style single-precision polynomials (log: a 128-bin table reduction and a cubic, see nrc_logf) whose Horner
 * steps are explicit single-rounding fused multiply-adds (v_fma_f32 == fmaf on the host); translation units that include this header are compiled with
 * -ffp-contract=off so that nothing else is contracted; hipcc's default correctly rounded fp32 / and sqrt are relied upon.  tests/test_gpu_math.py checks these bit-for-bit against the oracle's own statement.
 */
#ifndef NRC_MATH_H
#define NRC_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define NRC_HD __host__ __device__
#else
#define NRC_HD
#endif

#define NRC_PI 3.14159274101257324f      /* float(PI) of nrc-constants.glsl:33 */
#define NRC_TWO_PI 6.28318548202514648f  /* float(2.0*PI) */
#define NRC_HALF_PI 1.57079637050628662f
#define NRC_QUARTER_PI 0.785398185253143311f


/* single-rounding multiply-add: fmaf on the host, v_fma_f32 on the device -- both IEEE, hence bit-identical */
NRC_HD static inline float nrc_fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

NRC_HD static inline uint32_t nrc_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
NRC_HD static inline float nrc_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

/* natural log for normal positive x (call sites pass 1-u, u in [0,1) => [2^-23, 1]).
 * Spec (round 5; oracle/orc_math.h states the same): x = m * 2^e with m in [0.5, 1); bin i = top seven mantissa bits;
 * {inv_c, log_c} = NRC_LOG_TAB[i] (tools/make_log_table.py: inv_c = RN32(1/c_i), log_c = RN32(-ln(inv_c)));
 *   r = fma(m, inv_c, -1);  q = fma(fma(r, 1/3, -1/2), r, 1);  log(x) = fma(q, r, fma(e, LN2, log_c)).
 * Ten vector instructions and one 8-byte LDS read on the device (the Cephes polynomial this replaces: 20; the tracking loops spent a
 * fifth of their instructions in it) and at most 1.7 ulp from the correctly rounded value on every argument the integrator can pass.
 * `tab` is the table wherever the caller keeps it (the kernels: a copy in LDS, fill_log_table). */
#define NRC_LN2 0x1.62e430p-1f
#define NRC_THIRD 0x1.555556p-2f
struct NrcLogBin { float inv_c, log_c; };
NRC_HD static inline float nrc_logf(float x, const NrcLogBin* tab)
{
    const uint32_t ix = nrc_f2u(x);
    const int e = (int)((ix >> 23) & 0xffu) - 126;
    const float m = nrc_u2f((ix & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    const NrcLogBin t = tab[(ix >> 16) & 127u];
    const float r = nrc_fmaf_(m, t.inv_c, -1.0f);
    float q = nrc_fmaf_(r, NRC_THIRD, -0.5f);
    q = nrc_fmaf_(q, r, 1.0f);
    const float s = nrc_fmaf_((float)e, NRC_LN2, t.log_c);
    return nrc_fmaf_(q, r, s);
}

/* sin and cos of x (radians), |x| up to a few thousand */
NRC_HD static inline void nrc_sincosf(float x, float* s_out, float* c_out)
{
    float ax = fabsf(x);
    uint32_t j = (uint32_t)(ax * 1.27323949337005615f); /* 4/pi */
    j = (j + 1u) & ~1u;
    float y = (float)j;
    float r = nrc_fmaf_(-y, 0.78515625f, ax);
    r = nrc_fmaf_(-y, 2.4187564849853515625e-4f, r);
    r = nrc_fmaf_(-y, 3.77489497744594108e-8f, r);
    float z = r * r;
    float ps = -1.9515295891E-4f;
    ps = nrc_fmaf_(ps, z, 8.3321608736E-3f);
    ps = nrc_fmaf_(ps, z, -1.6666654611E-1f);
    ps = ps * z;
    ps = nrc_fmaf_(ps, r, r);
    float pc = 2.443315711809948E-005f;
    pc = nrc_fmaf_(pc, z, -1.388731625493765E-003f);
    pc = nrc_fmaf_(pc, z, 4.166664568298827E-002f);
    pc = pc * z;
    pc = pc * z;
    pc = nrc_fmaf_(-0.5f, z, pc);
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
NRC_HD static inline float nrc_asinf(float x)
{
    float a = fabsf(x);
    if (!(a <= 1.0f)) return nrc_u2f(0x7fc00000u);
    float z, w;
    int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
    else { w = a; z = a * a; }
    float p = 4.2163199048E-2f;
    p = nrc_fmaf_(p, z, 2.4181311049E-2f);
    p = nrc_fmaf_(p, z, 4.5470025998E-2f);
    p = nrc_fmaf_(p, z, 7.4953002686E-2f);
    p = nrc_fmaf_(p, z, 1.6666752422E-1f);
    p = p * z;
    p = nrc_fmaf_(p, w, w);
    if (big) { p = p + p; p = NRC_HALF_PI - p; }
    return x < 0.0f ? -p : p;
}

/* acos; NaN for |x| > 1 (GLSL: undefined; this is what makes quirk Q5 visible) */
NRC_HD static inline float nrc_acosf(float x)
{
    if (!(fabsf(x) <= 1.0f)) return nrc_u2f(0x7fc00000u);
    if (x < -0.5f) return NRC_PI - 2.0f * nrc_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * nrc_asinf(sqrtf(0.5f * (1.0f - x)));
    return NRC_HALF_PI - nrc_asinf(x);
}

/* acos with the argument clamped to [-1,1] (dir_gen.glsl:49: cosTheta reaches +-1 up to rounding) */
NRC_HD static inline float nrc_acosf_clamped(float x)
{
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    return nrc_acosf(x);
}

NRC_HD static inline float nrc_atanf(float x)
{
    float sgn = 1.0f;
    if (x < 0.0f) { sgn = -1.0f; x = -x; }
    float y;
    if (x > 2.41421365737915039f) { y = NRC_HALF_PI; x = -(1.0f / x); }
    else if (x > 0.414213567972183228f) { y = NRC_QUARTER_PI; x = (x - 1.0f) / (x + 1.0f); }
    else { y = 0.0f; }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = nrc_fmaf_(p, z, -1.38776856032E-1f);
    p = nrc_fmaf_(p, z, 1.99777106478E-1f);
    p = nrc_fmaf_(p, z, -3.33329491539E-1f);
    p = p * z;
    p = nrc_fmaf_(p, x, x);
    y = y + p;
    return sgn * y;
}

/* GLSL atan(y, x) */
NRC_HD static inline float nrc_atan2f(float y, float x)
{
    if (x == 0.0f) {
        if (y > 0.0f) return NRC_HALF_PI;
        if (y < 0.0f) return -NRC_HALF_PI;
        return 0.0f;
    }
    float a = nrc_atanf(y / x);
    if (x < 0.0f) a = (y >= 0.0f) ? a + NRC_PI : a - NRC_PI;
    return a;
}

/* IEEE binary16 <-> binary32, round-to-nearest-even (what a plain float->half cast does) */
NRC_HD static inline uint16_t nrc_f32_to_f16(float f)
{
    uint32_t x = nrc_f2u(f);
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

NRC_HD static inline float nrc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0u) {
        if (m == 0u) return nrc_u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f; /* 2^-24 */
        return sign ? -v : v;
    }
    if (e == 31u) return nrc_u2f(sign | 0x7f800000u | (m << 13));
    return nrc_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

NRC_HD static inline float nrc_round_f16(float f) { return nrc_f16_to_f32(nrc_f32_to_f16(f)); }

#endif
