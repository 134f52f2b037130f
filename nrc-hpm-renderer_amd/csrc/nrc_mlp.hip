// nrc_mlp.hip -- fused input encoding + fully fused MLP for gfx950 (CDNA4), hand-written MFMA kernels.
//
// Replaces tiny-cuda-nn v1.6 as used by src/NeuralRadianceCache.cu:16-39 (model config), :142 (inference),
// :153-154 (training_step / loss).  Arithmetic spec: SURVEY.md App. B (Frequency / OneBlob encodings,
// FullyFusedMLP with ReLU, RelativeL2Luminance, EMA{Adam}); precision: fp16 weights and activations,
// fp32 MFMA accumulation (>= the reference's fp16 accumulation), fp32 I/O, fp32 gradients and optimizer.
//
// Kernel design (wave64, v_mfma_f32_32x32x16_f16):
//   * orientation H^T = W * X^T: a wave owns 32 samples (MFMA columns = lanes), neurons run along the MFMA rows.
//     The 32x32 fp32 accumulator of layer l, ReLU'd and rounded to fp16 in registers, IS the B operand of layer
//     l+1 (the sum runs over the accumulator's row index), so activations never leave the register file.
//   * weights are pre-swizzled into "fragment images" ([frag][lane][8 halfs] = one ds_read_b128 per lane per MFMA,
//     conflict free) whose k-order matches the accumulator->operand permutation; each workgroup stages the image
//     into LDS once (54 KB) and then streams sample tiles persistently.
//   * the input encoding is computed straight into B-operand registers (v_fract/v_sin/v_cos hardware, argument
//     reduced exactly), the fp32 RGB output is stored from the accumulator: 20 B in + 12 B out per sample.
//   * training: same chain forward, loss, dgrad chain with W^T images; activations/deltas go to HBM once in
//     neuron-major fp16 so that the weight gradients are split-K MFMA GEMMs with a fixed-order slab reduction
//     (bitwise reproducible; no float atomics).
#include "nrc_mlp.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace nrc {

using half_t = _Float16;
using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

namespace {

constexpr int ENC = 80;          // Frequency-12 x 3 dims (72) + OneBlob-4 x 2 dims (8)
constexpr int WIDTH = 64;
constexpr int KS0 = ENC / 16;    // k-steps of layer 0
constexpr int KSH = WIDTH / 16;  // k-steps of a hidden layer
constexpr int MT = WIDTH / 32;   // 32-row M tiles per layer

// canonical feature index held by (k-step s, lane half h, element j) of the layer-0 B operand.
// s<4: natural order 16s+8h+j = four (sin,cos) pairs of one (dim, frequency quad);
// s=4: both lane halves get two (sin,cos) pairs of dim 2 and the four OneBlob bins of ONE direction dim,
//      so the two halves of a wave execute the same instruction stream.
__host__ __device__ inline int fmap80(int s, int h, int j)
{
    if (s < 4) return 16 * s + 8 * h + j;
    if (j < 4) return 64 + 4 * h + j;
    return 72 + 4 * h + (j - 4);
}
// neuron index held by (k-step s, lane half h, element j) when a 32x32 accumulator pair is re-used as B operand
__host__ __device__ inline int kperm(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ __forceinline__ half8 ld_frag(const uint4* lw, int frag, int lane)
{
    uint4 v = lw[frag * 64 + lane];
    return __builtin_bit_cast(half8, v);
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}

// ReLU + round-to-nearest-even fp16: accumulator registers 0..7 -> lo, 8..15 -> hi.
// One v_cvt_pk_f16_f32 per two values, then ReLU on the packed halfs as a signed 16-bit integer max with 0
// (v_pk_max_i16: negative halfs have the sign bit set; -0 -> +0).  fmaxf() on fp32 would cost two v_max_f32 per value
// (hipcc canonicalises MFMA outputs first); rounding commutes with ReLU, so the result is identical.
using float2v = float __attribute__((ext_vector_type(2)));
using f32x2 = float2v;
using half2v = _Float16 __attribute__((ext_vector_type(2)));
using short2v = short __attribute__((ext_vector_type(2)));
using uint4v = uint32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t relu_pk(float a, float b)
{
    float2v f = {a, b};
    half2v h = __builtin_convertvector(f, half2v);       // v_cvt_pk_f16_f32 (RNE)
    short2v s = __builtin_bit_cast(short2v, h);
    short2v z = {0, 0};
    s = __builtin_elementwise_max(s, z);                  // v_pk_max_i16
    return __builtin_bit_cast(uint32_t, s);
}
__device__ __forceinline__ void relu_pack(const f32x16& acc, half8& lo, half8& hi)
{
    uint4v l, h;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        l[j] = relu_pk(acc[2 * j], acc[2 * j + 1]);
        h[j] = relu_pk(acc[8 + 2 * j], acc[8 + 2 * j + 1]);
    }
    lo = __builtin_bit_cast(half8, l);
    hi = __builtin_bit_cast(half8, h);
}

// OneBlob(4 bins, periodic, quartic kernel of radius 1/4) of one coordinate.  tiny-cuda-nn sums the kernel's CDF over three periodic
// images at each of the five bin edges; the kernel is as wide as a bin, so only the two edges next to x are unsaturated: with
// t = 4x, j = floor(t), A = cdf(edge j), B = cdf(edge j+1) the bin left of x's gets A, x's own B - A, the next 1 - B and the fourth 0
// (indices mod 4).  Equal to the three-image sum over the whole range the renderer's direction coordinates take (-0.5, 1.5] up to fp32
// rounding (tests/test_oracle_nn.py::test_oneblob_two_edge_formula_equals_the_three_image_sum); two polynomials instead of five.
// NaN (quirk Q5): tiny-cuda-nn's fminf/fmaxf clamps give (0, 0, 0, 1).
__device__ __forceinline__ void oneblob4_bins(float xd, float (&out)[4])
{
    const float t = xd * 4.0f;
    const float jf = __builtin_floorf(t);
    const float fr = t - jf;
    const f32x2 u = {-fr, 1.0f - fr};
    const f32x2 u2 = u * u;
    const f32x2 u4 = u2 * u2;
    const f32x2 one = {1.0f, 1.0f};
    f32x2 p = (f32x2{15.0f / 16.0f, 15.0f / 16.0f} * u) * ((one - f32x2{2.0f / 3.0f, 2.0f / 3.0f} * u2) + f32x2{1.0f / 5.0f, 1.0f / 5.0f} * u4) + f32x2{0.5f, 0.5f};
    const float a = fminf(fmaxf(p[0], 0.0f), 1.0f), b = fminf(fmaxf(p[1], 0.0f), 1.0f);
    const int j = (int)jf & 3;
    const bool bad = !(xd == xd);
    const float own = b - a, next = 1.0f - b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = 0.0f;
        v = (k == ((j + 3) & 3)) ? a : v;
        v = (k == j) ? own : v;
        v = (k == ((j + 1) & 3)) ? next : v;
        out[k] = bad ? (k == 3 ? 1.0f : 0.0f) : v;
    }
}

// Input encoding of one sample straight into the five layer-0 B-operand fragments of this lane.
// Frequency: sin(2^f*pi*x + s*pi/2) = v_sin/v_cos(fract(x * 2^(f-1))) (hardware takes revolutions; the
// reduction x*2^(f-1) -> fract is exact in fp32, which defines the value for the large arguments of quirk Q3).
__device__ __forceinline__ void encode80(const float (&x)[5], int h, half8 (&b)[KS0])
{
    // per (dim, frequency quad): one exact (sin, cos) from the hardware units, the next three octaves by angle doubling
    // sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a (3 VALU ops instead of fract + 2 quarter-rate transcendentals; the
    // error grows 2x per octave, i.e. 8x the hardware's ~1e-6, far below fp16 resolution)
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int g0 = 2 * s, g1 = 2 * s + 1;                       // feature groups of the two lane halves
        const float xv = h ? x[g1 / 3] : x[g0 / 3];
        const float sc = h ? (float)(1 << (4 * (g1 % 3))) * 0.5f : (float)(1 << (4 * (g0 % 3))) * 0.5f;
        const float rev = __builtin_amdgcn_fractf(xv * sc);
        float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        b[s][0] = (half_t)sn;
        b[s][1] = (half_t)cs;
#pragma unroll
        for (int i = 1; i < 4; i++) {
            const float t2 = sn + sn;
            const float s2 = t2 * cs;
            cs = __builtin_fmaf(-t2, sn, 1.0f);
            sn = s2;
            b[s][2 * i] = (half_t)sn;
            b[s][2 * i + 1] = (half_t)cs;
        }
    }
    {
        const float rev = __builtin_amdgcn_fractf(x[2] * (h ? 512.0f : 128.0f));      // f = 10,11 | 8,9
        float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        b[4][0] = (half_t)sn;
        b[4][1] = (half_t)cs;
        const float t2 = sn + sn;
        b[4][2] = (half_t)(t2 * cs);
        b[4][3] = (half_t)__builtin_fmaf(-t2, sn, 1.0f);
        const float xd = h ? x[4] : x[3];
        float ob[4];
        oneblob4_bins(xd, ob);
#pragma unroll
        for (int k = 0; k < 4; k++) b[4][4 + k] = (half_t)ob[k];
    }
}

// the same encoding, one layer-0 k-step at a time (s is a compile-time constant after unrolling): lets the inference chain build
// each B fragment right before the MFMAs that consume it instead of holding all five per tile (40 VGPRs for two tiles)
template <int S>
__device__ __forceinline__ half8 encode80_frag(const float (&x)[5], int h)
{
    half8 b;
    if constexpr (S < 4) {
        const int g0 = 2 * S, g1 = 2 * S + 1;
        const float xv = h ? x[g1 / 3] : x[g0 / 3];
        const float sc = h ? (float)(1 << (4 * (g1 % 3))) * 0.5f : (float)(1 << (4 * (g0 % 3))) * 0.5f;
        const float rev = __builtin_amdgcn_fractf(xv * sc);
        float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        b[0] = (half_t)sn;
        b[1] = (half_t)cs;
#pragma unroll
        for (int i = 1; i < 4; i++) {
            const float t2 = sn + sn;
            const float s2 = t2 * cs;
            cs = __builtin_fmaf(-t2, sn, 1.0f);
            sn = s2;
            b[2 * i] = (half_t)sn;
            b[2 * i + 1] = (half_t)cs;
        }
    } else {
        const float rev = __builtin_amdgcn_fractf(x[2] * (h ? 512.0f : 128.0f));      // f = 10,11 | 8,9
        const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        b[0] = (half_t)sn;
        b[1] = (half_t)cs;
        const float t2 = sn + sn;
        b[2] = (half_t)(t2 * cs);
        b[3] = (half_t)__builtin_fmaf(-t2, sn, 1.0f);
        const float xd = h ? x[4] : x[3];
        float ob[4];
        oneblob4_bins(xd, ob);
#pragma unroll
        for (int k = 0; k < 4; k++) b[4 + k] = (half_t)ob[k];
    }
    return b;
}

// the same, s a constant after unrolling
__device__ __forceinline__ half8 encode80_step(const float (&x)[5], int h, int s)
{
    switch (s) {
    case 0: return encode80_frag<0>(x, h);
    case 1: return encode80_frag<1>(x, h);
    case 2: return encode80_frag<2>(x, h);
    case 3: return encode80_frag<3>(x, h);
    default: return encode80_frag<4>(x, h);
    }
}

// fragment bases inside the forward image
constexpr int FRAG_L0 = 0;                       // [mt][s]      MT*KS0
constexpr int FRAG_HID = MT * KS0;               // [l-1][mt][s] (depth-1)*MT*KSH
__host__ __device__ constexpr int frag_out(int depth) { return FRAG_HID + (depth - 1) * MT * KSH; }   // [s] KSH
__host__ __device__ constexpr int n_frag_fwd(int depth) { return frag_out(depth) + KSH; }
// backward image: hidden l=1..depth-1 [l-1][mt][s], then out [mt]
__host__ __device__ constexpr int n_frag_bwd(int depth) { return (depth - 1) * MT * KSH + MT; }

template <int DEPTH, bool KEEP>
struct FwdState {
    half8 enc[KS0];
    half8 act[KEEP ? DEPTH : 1][KSH];
};

// forward chain for one 32-sample tile; returns the output-layer accumulator (rows 0..2 = RGB on lanes h==0)
template <int DEPTH, bool KEEP>
__device__ __forceinline__ f32x16 forward_tile(const uint4* lw, int lane, FwdState<DEPTH, KEEP>& st)
{
    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
    for (int s = 0; s < KS0; s++) {
        acc0 = mfma(ld_frag(lw, FRAG_L0 + 0 * KS0 + s, lane), st.enc[s], acc0);
        acc1 = mfma(ld_frag(lw, FRAG_L0 + 1 * KS0 + s, lane), st.enc[s], acc1);
    }
    half8 b[KSH];
    relu_pack(acc0, b[0], b[1]);
    relu_pack(acc1, b[2], b[3]);
    if (KEEP) {
#pragma unroll
        for (int s = 0; s < KSH; s++) st.act[0][s] = b[s];
    }
#pragma unroll
    for (int l = 1; l < DEPTH; l++) {
        acc0 = zero16();
        acc1 = zero16();
        const int base = FRAG_HID + (l - 1) * MT * KSH;
#pragma unroll
        for (int s = 0; s < KSH; s++) {
            acc0 = mfma(ld_frag(lw, base + s, lane), b[s], acc0);
            acc1 = mfma(ld_frag(lw, base + KSH + s, lane), b[s], acc1);
        }
        relu_pack(acc0, b[0], b[1]);
        relu_pack(acc1, b[2], b[3]);
        if (KEEP) {
#pragma unroll
            for (int s = 0; s < KSH; s++) st.act[l][s] = b[s];
        }
    }
    f32x16 y = zero16();
#pragma unroll
    for (int s = 0; s < KSH; s++) y = mfma(ld_frag(lw, frag_out(DEPTH) + s, lane), b[s], y);
    return y;
}

__device__ __forceinline__ void stage_lds(uint4* dst, const uint4* src, int n16, int tid, int nthreads)
{
    for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ inference
// forward chain for NT independent 32-sample tiles of one wave: every weight fragment read from LDS feeds NT MFMAs, and
// the ReLU/convert VALU work of one tile overlaps the MFMAs of the other inside the wave (in-order issue needs the ILP).
// ABL (diagnostic builds only, never the default): bit 0 = trivial encoding, bit 1 = no ReLU/convert work
template <int ABL>
__device__ __forceinline__ void relu_pack_abl(const f32x16& acc, half8& lo, half8& hi)
{
    if constexpr ((ABL & 2) != 0) {
        uint4v l = {__builtin_bit_cast(uint32_t, acc[0]), __builtin_bit_cast(uint32_t, acc[1]),
                    __builtin_bit_cast(uint32_t, acc[2]), __builtin_bit_cast(uint32_t, acc[3])};
        uint4v h = {__builtin_bit_cast(uint32_t, acc[8]), __builtin_bit_cast(uint32_t, acc[9]),
                    __builtin_bit_cast(uint32_t, acc[10]), __builtin_bit_cast(uint32_t, acc[11])};
        lo = __builtin_bit_cast(half8, l);
        hi = __builtin_bit_cast(half8, h);
    } else {
        relu_pack(acc, lo, hi);
    }
}

// forward chain for NT independent 32-sample tiles of one wave: every weight fragment read from LDS feeds NT MFMAs, and
// the ReLU/convert VALU work of one tile can overlap the MFMAs of the other inside the wave (in-order issue needs the ILP).
template <int DEPTH, int NT, int ABL, int S>
__device__ __forceinline__ void layer0_steps(const uint4* lw, int lane, int h, const float (&x)[NT][5], f32x16 (&acc0)[NT], f32x16 (&acc1)[NT])
{
    if constexpr (S < KS0) {
        const half8 a0 = ld_frag(lw, FRAG_L0 + S, lane);
        const half8 a1 = ld_frag(lw, FRAG_L0 + KS0 + S, lane);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            half8 bs;
            if constexpr ((ABL & 1) != 0) {
#pragma unroll
                for (int j = 0; j < 8; j++) bs[j] = (half_t)x[t][(S + j) % 5];
            } else {
                bs = encode80_frag<S>(x[t], h);
            }
            acc0[t] = mfma(a0, bs, acc0[t]);
            acc1[t] = mfma(a1, bs, acc1[t]);
        }
        layer0_steps<DEPTH, NT, ABL, S + 1>(lw, lane, h, x, acc0, acc1);
    }
}

template <int DEPTH, int NT, int ABL = 0>
__device__ __forceinline__ void forward_tiles(const uint4* lw, int lane, int h, const float (&x)[NT][5], f32x16 (&y)[NT])
{
    f32x16 acc0[NT], acc1[NT];
    half8 b[NT][KSH];
#pragma unroll
    for (int t = 0; t < NT; t++) { acc0[t] = zero16(); acc1[t] = zero16(); }
    layer0_steps<DEPTH, NT, ABL, 0>(lw, lane, h, x, acc0, acc1);      // the encoding is built fragment by fragment beside the MFMAs
#pragma unroll
    for (int t = 0; t < NT; t++) {
        relu_pack_abl<ABL>(acc0[t], b[t][0], b[t][1]);
        relu_pack_abl<ABL>(acc1[t], b[t][2], b[t][3]);
    }
#pragma unroll
    for (int l = 1; l < DEPTH; l++) {
        const int base = FRAG_HID + (l - 1) * MT * KSH;
#pragma unroll
        for (int t = 0; t < NT; t++) { acc0[t] = zero16(); acc1[t] = zero16(); }
#pragma unroll
        for (int s = 0; s < KSH; s++) {
            const half8 a0 = ld_frag(lw, base + s, lane);
            const half8 a1 = ld_frag(lw, base + KSH + s, lane);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acc0[t] = mfma(a0, b[t][s], acc0[t]);
                acc1[t] = mfma(a1, b[t][s], acc1[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            relu_pack_abl<ABL>(acc0[t], b[t][0], b[t][1]);
            relu_pack_abl<ABL>(acc1[t], b[t][2], b[t][3]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) y[t] = zero16();
#pragma unroll
    for (int s = 0; s < KSH; s++) {
        const half8 a = ld_frag(lw, frag_out(DEPTH) + s, lane);
#pragma unroll
        for (int t = 0; t < NT; t++) y[t] = mfma(a, b[t][s], y[t]);
    }
}

// (Measured, tools/infer_micro.hip + profiles/r02_micro_infer_structure.txt: running the two tiles of a wave half a layer apart,
// so that the ReLU/convert of one tile fills the MFMA gaps of the other -- 4 VALU per 32-cycle MFMA, fragments re-read per
// tile, sched_barrier after every gap.  The bare layer stream then keeps the MFMA pipes 96 % busy instead of 84 %, but only
// trivial operands let the chip run it at 2.4 GHz: on random activations it holds 1.63-1.89 GHz, so even that stream tops
// out at 68 % of the nominal peak (58 % with the encoding's VALU share, against 50 % for this order).  In k_infer itself,
// where the encoding sits in front of layer 0 instead of being spread over the gaps, the skewed order ran within 1 % of
// this one at every batch size, so the simpler order stayed.)
// persistent workgroups; NT 32-sample tiles per wave per iteration; the next iteration's queries are loaded (20 B per
// sample, straight from HBM/L2 into registers) before the current tiles are computed, so their latency is hidden.
// nrc/render.comp:23-41 for the pixel of query q (tile-major order): c = primary.rgb + (showNrc && scattered ? max(0, nrc) * primary.w : 0),
// out = blend * c + (1 - blend) * out_prev, alpha likewise towards 1.  The arithmetic is k_composite's, operation for operation
// and without contraction, so the fused and the separate pass give the same bits.
__device__ __forceinline__ void composite_query(const CompositeArgs& ca, uint32_t q, float nr, float ng, float nb)
{
#pragma clang fp contract(off)
    const uint32_t tiles_x = (ca.w + 7u) >> 3;
    const uint32_t tile = q >> 6, in_tile = q & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t lx = tx * 8u + (in_tile & 7u), y = ty * 8u + (in_tile >> 3);
    if (lx >= ca.w || y >= ca.h) return;
    const size_t pix = (size_t)y * ca.w + lx;
    const float4 p = reinterpret_cast<const float4*>(ca.primary)[pix];
    float cr = p.x, cg = p.y, cb = p.z;
    if (ca.show_nrc == 1u && ca.info[pix] == 1.0f) {
        cr += fmaxf(0.0f, nr) * p.w;
        cg += fmaxf(0.0f, ng) * p.w;
        cb += fmaxf(0.0f, nb) * p.w;
    }
    float4* fb = reinterpret_cast<float4*>(ca.framebuffer) + pix;
    const float4 prev = *fb;
    const float ib = 1.0f - ca.blend_factor;
    *fb = make_float4(ca.blend_factor * cr + ib * prev.x, ca.blend_factor * cg + ib * prev.y,
                      ca.blend_factor * cb + ib * prev.z, ca.blend_factor * 1.0f + ib * prev.w);
}

template <int DEPTH, int THREADS, int NT, int ABL = 0, bool FUSE = false, bool LISTED = false>
// (second launch bound.  With ONE wave per SIMD the compiler assumes the 512-register budget is split into 256 architectural + 256
// accumulation registers, puts the MFMA accumulators into AGPRs and copies every one of them out with v_accvgpr_read_b32 before the
// ReLU / pack: 384 of the 1 346 instructions of the renderer-mode loop of the 4-wave form.  Two waves per SIMD = 256 registers, all
// architectural on this target: 122 of them, no copies, 29 % fewer instructions -- and the default preset's frame 1.9 % SLOWER (8 125 ->
// 7 980 Msamples/s, three alternating runs: gen_rays 0.234 -> 0.238 ms beside the denser kernel; the frame is a balance, DESIGN.md
// section 4).  NRC_INFER_MIN_WAVES=2 builds it.  The generic kernels -- k_infer_gen, k_train_gen2, k_wgrad2 -- do ask for two: configs[4]
// + 1 %, HashGrid + 0.6 %.)
#ifndef NRC_INFER_MIN_WAVES
#define NRC_INFER_MIN_WAVES 1
#endif
__global__ __launch_bounds__(THREADS, THREADS == 512 && NT == 2 ? 4 : NRC_INFER_MIN_WAVES) void k_infer(const float* __restrict__ in, float* __restrict__ out, uint32_t n,
                                                  const uint4* __restrict__ image, unsigned long long* __restrict__ stamps = nullptr,
                                                  int skip_zero = 0, CompositeArgs ca = CompositeArgs{},
                                                  const uint32_t* __restrict__ live_list = nullptr,
                                                  const uint32_t* __restrict__ live_count = nullptr)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* lw = reinterpret_cast<uint4*>(smem);
    unsigned long long t0 = 0, r0 = 0;
    if constexpr ((ABL & 4) != 0) {     // diagnostic: shader clock vs 100 MHz wall clock (MI355X_MICROARCH.md DVFS item 6)
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    stage_lds(lw, image, n_frag_fwd(DEPTH) * 64, threadIdx.x, THREADS);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // live_list (renderer inference, round 4): the tiles are 32 entries of the frame's live-query list (DevFrame::live_list: the query index
    // of every pixel that scattered, appended by k_gen_rays) instead of 32 consecutive queries -- nothing is read, computed or stored for a
    // pixel that did not scatter (the list-free renderer mode reads all 2 M queries of a 1080p frame and stores zeros for the dead
    // tiles: 66 MB), and the tiles that run are full (87 % on the bench cloud otherwise, tools/live_tiles.py).  Results are per query.
    // (LISTED is a template parameter: the list-free instantiations are the code they were)
    constexpr bool listed = LISTED && !FUSE;
    uint32_t n_q = n;
    if constexpr (listed) {
        n_q = __builtin_amdgcn_readfirstlane((int)*live_count);
        n_q = n_q < n ? n_q : n;
    }
    auto query_of = [&](uint32_t sx) -> uint32_t {      // (slots beyond the end repeat the last query: in bounds, never stored)
        if constexpr (listed) return live_list[sx < n_q ? sx : (n_q != 0u ? n_q - 1u : 0u)];
        else return sx < n ? sx : n - 1u;
    };
    const uint32_t n_tiles = (n_q + 31u) >> 5;
    const uint32_t stride = gridDim.x * (THREADS / 64) * NT;
    uint32_t tile = (blockIdx.x * (THREADS / 64) + wave) * NT;
    float x[NT][5];
    uint32_t qi[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        qi[t] = query_of((tile + t) * 32u + r);
        const float* p = in + (size_t)qi[t] * 5u;
#pragma unroll
        for (int i = 0; i < 5; i++) x[t][i] = __builtin_nontemporal_load(p + i);
    }
    __syncthreads();
    unsigned long long r_staged = 0;
    if constexpr ((ABL & 4) != 0) r_staged = __builtin_amdgcn_s_memrealtime();
    for (; tile < n_tiles; tile += stride) {
        // the weight image in LDS is loop invariant: keep hipcc from hoisting all 54 fragments (216 VGPRs) out of
        // the tile loop -- fragments are meant to be re-read from LDS, one ds_read_b128 per NT MFMAs
        asm volatile("" ::: "memory");
        float xn[NT][5];
        uint32_t qn[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            qn[t] = query_of((tile + stride + t) * 32u + r);
            const float* p = in + (size_t)qn[t] * 5u;
#pragma unroll
            for (int i = 0; i < 5; i++) xn[t][i] = __builtin_nontemporal_load(p + i);
        }
        f32x16 y[NT];
        // renderer mode WITHOUT the live-query list: gen_rays writes an all-zero query for every pixel that did not scatter (the reference's
        // zero-filled slots, whose network output render.comp never reads); tiles made only of such queries skip the network and store 0
        bool live = true;
        if (skip_zero && !listed) {
            bool nz = false;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int i = 0; i < 5; i++) nz |= (__builtin_bit_cast(uint32_t, x[t][i]) & 0x7fffffffu) != 0u;
            live = __ballot(nz) != 0ull;
        }
        if (live) {
            forward_tiles<DEPTH, NT, ABL>(lw, lane, h, x, y);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) y[t] = zero16();
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const uint32_t sidx = (tile + t) * 32u + r;
            if (sidx < n_q && h == 0 && ((ABL & 8) == 0 || sidx < 64u)) {
                // streamed once: non-temporal, so the 12 B/sample leave the L2 while the kernel runs instead of in the
                // end-of-kernel write-back
                float* o = out + (size_t)(listed ? qi[t] : sidx) * 3u;
                __builtin_nontemporal_store(y[t][0], o);
                __builtin_nontemporal_store(y[t][1], o + 1);
                __builtin_nontemporal_store(y[t][2], o + 2);
                if constexpr (FUSE) composite_query(ca, ca.q_base + sidx, y[t][0], y[t][1], y[t][2]);
            }
#pragma unroll
            for (int i = 0; i < 5; i++) x[t][i] = xn[t][i];
            qi[t] = qn[t];
        }
    }
    if constexpr ((ABL & 4) != 0) {
        if (threadIdx.x == 0) {
            stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
            stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
            stamps[4 * blockIdx.x + 2] = r0;
            stamps[4 * blockIdx.x + 3] = r_staged - r0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ k-group stores
// Activations and deltas go to HBM as [sample/8][row][8 samples] fp16 ("k-groups": the 16 bytes one lane of the weight-gradient
// MFMA loads).  A lane of the forward chain holds the transposed piece -- ONE sample, 8 rows in a half8 -- and used to write it
// with eight 2-byte stores (1 600 scattered short stores and loads per tile of an 8x128 net: the memory pipeline, not the MFMAs,
// set the kernels' time).  Now the wave turns each 16-row x 32-sample block through 1.25 KB of its own LDS (8 ds_write_b16, one
// ds_read_b128 per lane; rows padded to 40 halves so that the two lane halves hit different banks) and stores 16 bytes per lane.
// PERM: the half8's rows are 8(j>>2) + 4h + (j&3) (an accumulator pair re-used as operand, kperm), else 8h + j (natural order).
constexpr int KG_ROW = 40;                          // halves per scratch row
constexpr int KG_SCRATCH = 16 * KG_ROW * 2;         // bytes per wave
template <bool PERM>
__device__ __forceinline__ void store_kgroups(half_t* scratch, const half8& v, int lane, half_t* dst_row0, uint32_t rows)
{
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int rho = PERM ? 8 * (j >> 2) + 4 * h + (j & 3) : 8 * h + j;
        scratch[rho * KG_ROW + r] = v[j];
    }
    __builtin_amdgcn_wave_barrier();                 // same wave, LDS executes in order: only the compiler must not reorder
    const half8 t = *reinterpret_cast<const half8*>(scratch + (lane >> 2) * KG_ROW + 8 * (lane & 3));
    __builtin_amdgcn_wave_barrier();
    // lane -> row lane >> 2 of the block, sample group lane & 3 of the tile
    *reinterpret_cast<half8*>(dst_row0 + ((size_t)(lane & 3) * rows + (size_t)(lane >> 2)) * 8) = t;
}

// ------------------------------------------------------------------------------------------------ training: fwd + loss + dgrad
// tiny-cuda-nn's element-wise losses (the reference passes the name through, src/NeuralRadianceCache.cu:17-19):
// per element value / n_total and dL/dy * loss_scale / n_total, n_total = 3 * (global) batch.
//   0 RelativeL2Luminance (p-t)^2 / (lum(p)^2 + .01)      1 L2 (p-t)^2             2 RelativeL2 (p-t)^2 / (p^2 + .01)
//   3 L1 |p-t|        4 Mape |p-t| / (|t| + .01)        5 Smape |p-t| / ((|p|+|t|)/2 + .01)        6 LogL1 log(1 + |p-t|)
// (the relative losses treat their denominator as a constant in the gradient, as tiny-cuda-nn does)
__device__ __forceinline__ void loss_terms(uint32_t loss_id, const float (&y)[3], const float* __restrict__ t, float inv_n_total,
                                           float& loss_v, float (&dy)[3])
{
    float lum2 = 0.0f;
    if (loss_id == 0u) {
        const float lum = (0.299f * y[0] + 0.587f * y[1]) + 0.114f * y[2];
        lum2 = lum * lum + 0.01f;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float d = y[c] - t[c];
        float value, grad;
        if (loss_id <= 2u) {
            const float den = loss_id == 0u ? lum2 : (loss_id == 1u ? 1.0f : y[c] * y[c] + 0.01f);
            value = d * d / den;
            grad = 2.0f * d / den;
        } else if (loss_id == 6u) {
            const float div = fabsf(d) + 1.0f;
            value = logf(div);
            grad = copysignf(1.0f / div, d);
        } else {
            const float scale = loss_id == 3u ? 1.0f : 1.0f / ((loss_id == 4u ? fabsf(t[c]) : 0.5f * (fabsf(y[c]) + fabsf(t[c]))) + 0.01f);
            value = fabsf(d) * scale;
            grad = copysignf(scale, d);
        }
        loss_v += value * inv_n_total;
        dy[c] = Mlp::kLossScale * (grad * inv_n_total);
    }
}

struct TrainArgs {
    const float* in;
    const float* target;
    uint32_t n;
    float inv_n_total;    // 1 / (3 * n_norm)
    uint32_t loss_id;
    half_t* acts;         // [n/8][ENC + DEPTH*WIDTH][8]   (row = feature / neuron; 8 consecutive samples contiguous)
    half_t* deltas;       // [n/8][DEPTH*WIDTH + 8][8]
    float* loss_part;     // [n/32]
};

// Round 4: the same arithmetic in two phases per round of tiles, so that the workgroup needs ONE weight image in LDS at a time
// (54 KB instead of 98: a CU that holds a k_infer workgroup beside gen_rays' five has 66 KB free) and 128 registers instead of 272:
//   phase 1 (forward image staged): encode, forward chain -- every layer's activations go to HBM the moment they exist and ONE BIT per
//           activation (is it positive: all the dgrad chain needs of it) stays in registers --, loss and dL/dy;
//   phase 2 (W^T image staged over the forward image): the dgrad chain from dL/dy and the bits.
// (Round 3's one-phase kernel, k_train_fwd_bwd -- both images in LDS, 272 registers -- was bit-identical and is gone: git history.)
template <int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void k_train_fwd_bwd_light(TrainArgs a, const uint4* __restrict__ img_fwd,
                                                                   const uint4* __restrict__ img_bwd)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* lw = reinterpret_cast<uint4*>(smem);      // the forward image, then the W^T image
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t n_tiles = a.n >> 5;               // host guarantees n % 32 == 0
    constexpr uint32_t WAVES = THREADS / 64;
    const uint32_t stride = gridDim.x * WAVES;
    constexpr int ROWS_A = ENC + DEPTH * WIDTH;
    constexpr int ROWS_D = DEPTH * WIDTH + 8;
    for (uint32_t t0 = blockIdx.x * WAVES; t0 < n_tiles; t0 += stride) {      // a round: one tile per wave; uniform over the workgroup
        const uint32_t tile = t0 + (uint32_t)wave;
        const bool active = tile < n_tiles;
        if (t0 != blockIdx.x * WAVES) __syncthreads();                       // (the previous round's dgrad chains have read the W^T image)
        stage_lds(lw, img_fwd, n_frag_fwd(DEPTH) * 64, threadIdx.x, THREADS);
        __syncthreads();
        const uint32_t sidx = tile * 32u + r;
        uint32_t relu_bits[DEPTH];
        half8 bo;
#pragma unroll
        for (int j = 0; j < 8; j++) bo[j] = (half_t)0.0f;
#pragma unroll
        for (int l = 0; l < DEPTH; l++) relu_bits[l] = 0u;
        half_t* const pd = a.deltas + ((size_t)(sidx >> 3) * ROWS_D) * 8 + (sidx & 7u);
        half_t* const pd4 = pd + 32 * h;
        if (active) {
            const float* p = a.in + (size_t)sidx * 5u;
            float x[5];
#pragma unroll
            for (int i = 0; i < 5; i++) x[i] = p[i];
            half8 enc[KS0];
            encode80(x, h, enc);
            // ---- activations -> HBM in [sample/8][row][8] order: the 16-byte k-groups the weight-gradient GEMM reads; row offsets are
            //      compile-time immediates on two per-lane bases (+8h / +4h rows)
            half_t* const pa = a.acts + ((size_t)(sidx >> 3) * ROWS_A) * 8 + (sidx & 7u);
            half_t* const pa8 = pa + 64 * h;      // rows + 8h
            half_t* const pa4 = pa + 32 * h;      // rows + 4h
#pragma unroll
            for (int s = 0; s < KS0; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (s < 4) pa8[(16 * s + j) * 8] = enc[s][j];
                    else pa4[(j < 4 ? 64 + j : 72 + (j - 4)) * 8] = enc[s][j];
                }
            f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
            for (int s = 0; s < KS0; s++) {
                acc0 = mfma(ld_frag(lw, FRAG_L0 + 0 * KS0 + s, lane), enc[s], acc0);
                acc1 = mfma(ld_frag(lw, FRAG_L0 + 1 * KS0 + s, lane), enc[s], acc1);
            }
            half8 b[KSH];
#pragma unroll
            for (int l = 0; l < DEPTH; l++) {
                asm volatile("" ::: "memory");      // a layer's LDS fragment reads stay in the layer (hoisted, they cost 4 VGPRs each)
                if (l > 0) {
                    acc0 = zero16();
                    acc1 = zero16();
                    const int base = FRAG_HID + (l - 1) * MT * KSH;
#pragma unroll
                    for (int s = 0; s < KSH; s++) {
                        acc0 = mfma(ld_frag(lw, base + s, lane), b[s], acc0);
                        acc1 = mfma(ld_frag(lw, base + KSH + s, lane), b[s], acc1);
                    }
                }
                relu_pack(acc0, b[0], b[1]);
                relu_pack(acc1, b[2], b[3]);
                uint32_t bits = 0u;
#pragma unroll
                for (int s = 0; s < KSH; s++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        pa4[(ENC + WIDTH * l + kperm(s, 0, j)) * 8] = b[s][j];
                        bits |= (b[s][j] > (half_t)0.0f) ? (1u << (8 * s + j)) : 0u;
                    }
                // computed HERE: sunk to its use in the dgrad chain (where the scheduler puts it), the word keeps the layer's 16 VGPRs of
                // activations alive -- 305 VGPRs instead of 140
                asm volatile("" : "+v"(bits));
                relu_bits[l] = bits;
            }
            f32x16 y = zero16();
#pragma unroll
            for (int s = 0; s < KSH; s++) y = mfma(ld_frag(lw, frag_out(DEPTH) + s, lane), b[s], y);

            // ---- loss + dL/dy (lanes h==0 hold y)
            float loss_v = 0.0f;
            if (h == 0) {
                const float* t = a.target + (size_t)sidx * 3u;
                const float yv[3] = {y[0], y[1], y[2]};
                float dy[3];
                loss_terms(a.loss_id, yv, t, a.inv_n_total, loss_v, dy);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    bo[c] = (half_t)dy[c];
                    pd[(DEPTH * WIDTH + c) * 8] = bo[c];
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) loss_v += __shfl_xor(loss_v, off);
            if (lane == 0) a.loss_part[tile] = loss_v;
        }
        __syncthreads();                                                     // every forward chain has read the forward image
        stage_lds(lw, img_bwd, n_frag_bwd(DEPTH) * 64, threadIdx.x, THREADS);
        __syncthreads();
        if (active) {
            // ---- dgrad chain: delta_{l-1} = relu'(a_{l-1}) * (W_l^T delta_l)
            const int bout = (DEPTH - 1) * MT * KSH;
            f32x16 d0 = mfma(ld_frag(lw, bout + 0, lane), bo, zero16());
            f32x16 d1 = mfma(ld_frag(lw, bout + 1, lane), bo, zero16());
#pragma unroll
            for (int l = DEPTH - 1; l >= 0; l--) {
                half8 dl[KSH];
                const uint32_t bits = relu_bits[l];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    dl[0][j] = ((bits >> j) & 1u) ? (half_t)d0[j] : (half_t)0.0f;
                    dl[1][j] = ((bits >> (8 + j)) & 1u) ? (half_t)d0[8 + j] : (half_t)0.0f;
                    dl[2][j] = ((bits >> (16 + j)) & 1u) ? (half_t)d1[j] : (half_t)0.0f;
                    dl[3][j] = ((bits >> (24 + j)) & 1u) ? (half_t)d1[8 + j] : (half_t)0.0f;
                }
#pragma unroll
                for (int s = 0; s < KSH; s++)
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        pd4[(WIDTH * l + kperm(s, 0, j)) * 8] = dl[s][j];
                if (l > 0) {
                    asm volatile("" ::: "memory");
                    d0 = zero16();
                    d1 = zero16();
                    const int base = (l - 1) * MT * KSH;
#pragma unroll
                    for (int s = 0; s < KSH; s++) {
                        d0 = mfma(ld_frag(lw, base + s, lane), dl[s], d0);
                        d1 = mfma(ld_frag(lw, base + KSH + s, lane), dl[s], d1);
                    }
                }
            }
        }
    }
}

// ================================================================================================ generic path
// Any supported encoding, width 64 or 128, any depth: the encoding is a separate kernel writing fp16 features [n][E16]
// (E16 = encoded dims padded to a multiple of 16 with 1.0, as tiny-cuda-nn does), the MLP kernels read layer-0 B operands
// from there and weight fragments straight from the L2-resident images (an 8x128 network's 250 KB do not fit the LDS
// budget of resident workgroups).  Same accumulator-as-operand chain, same numerics as the fused 6x64 kernels.
__device__ __forceinline__ float tri_wave(float x, int f)      // TriangleWave: |((2^f x) mod 2) - 1|
{
    const float t = x * (float)(1 << f);
    const float r = t - 2.0f * floorf(t * 0.5f);
    return fabsf(r - 1.0f);
}

__device__ __forceinline__ void oneblob4(float xd, float (&out)[4]) { oneblob4_bins(xd, out); }

// one thread per sample; features in tiny-cuda-nn order: position encoding (dims 0-2) then direction encoding (dims 3-4).
// POS / DIR are compile-time so that every feature has a static slot.
template <int POS, int DIR>
__global__ __launch_bounds__(256) void k_encode(const float* __restrict__ in, half_t* __restrict__ feat, uint32_t n, int skip_zero)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int NP = POS == 3 ? 72 : (POS == 1 ? 3 : 36);
    constexpr int ND = DIR == 1 ? 2 : 8;
    constexpr int E16 = (NP + ND + 15) / 16 * 16;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float x[5];
#pragma unroll
    for (int d = 0; d < 5; d++) x[d] = in[(size_t)i * 5u + d];
    // renderer inference: the all-zero query of an unscattered pixel gets no features -- its MFMA column is computed from
    // whatever the buffer holds (columns are independent) and its output is never read; saves 64 % of this kernel's stores
    if (skip_zero != 0 && x[0] == 0.0f && x[1] == 0.0f && x[2] == 0.0f && x[3] == 0.0f && x[4] == 0.0f) return;
    half_t* o = feat + (size_t)i * E16;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (POS == 3) {
            float t = x[d] * 0.5f;
#pragma unroll
            for (int f = 0; f < 12; f++) {
                const float rev = __builtin_amdgcn_fractf(t);
                o[d * 24 + 2 * f] = (half_t)__builtin_amdgcn_sinf(rev);
                o[d * 24 + 2 * f + 1] = (half_t)__builtin_amdgcn_cosf(rev);
                t = t + t;
            }
        } else if (POS == 1) {
            o[d] = (half_t)x[d];
        } else {
#pragma unroll
            for (int f = 0; f < 12; f++) o[d * 12 + f] = (half_t)tri_wave(x[d], f);
        }
    }
#pragma unroll
    for (int d = 0; d < 2; d++) {
        if (DIR == 0) {
            float b[4];
            oneblob4(x[3 + d], b);
#pragma unroll
            for (int q = 0; q < 4; q++) o[NP + 4 * d + q] = (half_t)b[q];
        } else if (DIR == 1) {
            o[NP + d] = (half_t)x[3 + d];
        } else {
#pragma unroll
            for (int f = 0; f < 4; f++) o[NP + 4 * d + f] = (half_t)tri_wave(x[3 + d], f);
        }
    }
#pragma unroll
    for (int k = NP + ND; k < E16; k++) o[k] = (half_t)1.0f;
}

template <int POS>
static void launch_encode_pos(uint32_t dir_id, dim3 g, hipStream_t s, const float* in, half_t* feat, uint32_t n, int skip)
{
    if (dir_id == 0) hipLaunchKernelGGL((k_encode<POS, 0>), g, dim3(256), 0, s, in, feat, n, skip);
    else if (dir_id == 1) hipLaunchKernelGGL((k_encode<POS, 1>), g, dim3(256), 0, s, in, feat, n, skip);
    else hipLaunchKernelGGL((k_encode<POS, 2>), g, dim3(256), 0, s, in, feat, n, skip);
}
static void launch_encode(uint32_t pos_id, uint32_t dir_id, hipStream_t s, const float* in, half_t* feat, uint32_t n, bool skip_zero)
{
    const dim3 g(ceil_div(n, 256));
    const int skip = skip_zero ? 1 : 0;
    if (pos_id == 3) launch_encode_pos<3>(dir_id, g, s, in, feat, n, skip);
    else if (pos_id == 1) launch_encode_pos<1>(dir_id, g, s, in, feat, n, skip);
    else launch_encode_pos<2>(dir_id, g, s, in, feat, n, skip);
}

// ---- HashGrid position encoding (AppConfig posID 0, src/AppConfig.cpp:19-27: 16 levels x 2 features, 2^19 entries per hashed
// level, base resolution 16, per-level scale 2).  tiny-cuda-nn v1.6 grid semantics (PARITY UNPINNED):
// pos = fma(scale, x, 0.5), trilinear weights over the 8 corners, dense index while res^3 fits the level else the coherent
// prime hash, index % level size.  The table is trainable: fp32 master + EMA in the parameter vector, fp16 (half2 per
// entry) copies for the gathers.
constexpr uint32_t HG_LEVELS = 16;
struct HashLevels {
    uint32_t off[HG_LEVELS + 1];      // per-level entry offsets
};

#pragma clang fp contract(off)
__device__ __forceinline__ void hg_corners(const HashLevels& lv, uint32_t level, const float (&x)[3], uint32_t (&idx)[8], float (&w8)[8])
{
    const uint32_t res = 16u << level;                      // ceil(scale) + 1
    const float scale = (float)res - 1.0f;                  // exp2(level) * 16 - 1
    const uint32_t base = lv.off[level], hsize = lv.off[level + 1] - base;
    float pos[3];
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = __builtin_fmaf(scale, x[d], 0.5f);
        const float tmp = floorf(pos[d]);
        pg[d] = (uint32_t)(int)tmp;
        pos[d] -= tmp;
    }
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        float w = 1.0f;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if ((c & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        uint32_t stride = 1, index = 0;
#pragma unroll
        for (int d = 0; d < 3; d++)
            if (stride <= hsize) { index += pl[d] * stride; stride *= res; }
        if (hsize < stride) index = (pl[0] * 1u) ^ (pl[1] * 2654435761u) ^ (pl[2] * 805459861u);
        idx[c] = base + index % hsize;
        w8[c] = w;
    }
}

// The 8 corner gathers of one (sample, level).  Round 4: x-neighbours in ONE 8-byte load wherever they share an aligned pair of
// entries.  The x coordinate enters the dense index with stride 1 and the hash with the prime 1, so for an EVEN cell coordinate the
// corners x and x + 1 differ in bit 0 of the entry index only (dense: index + 1, hash: index ^ 1; level sizes and offsets are even) --
// half of all (sample, level) pairs.  The test is made on the indices themselves (idx[c + 1] == idx[c] ^ 1), so nothing is assumed
// about the level: a lane whose four x-pairs all pass does 4 gathers of 8 bytes, the others 4 + 4.  Same entries, same order of the
// weighted sum -- bit-identical features, 25 % fewer lane-gathers.  (-DNRC_HG_NO_PAIR_GATHER is the A/B build.)
__device__ __forceinline__ void hg_gather(const uint32_t* __restrict__ table16, const uint32_t (&idx)[8], const float (&w8)[8], float& r0, float& r1)
{
    uint32_t v[8];
#ifdef NRC_HG_NO_PAIR_GATHER
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = table16[idx[c]];
#else
    bool paired = true;
#pragma unroll
    for (int c = 0; c < 8; c += 2) paired = paired && idx[c + 1] == (idx[c] ^ 1u);
    uint32_t u[4] = {0u, 0u, 0u, 0u};
    if (!paired) {            // issued first: the lanes that need them have all eight gathers in flight at once
#pragma unroll
        for (int c = 0; c < 4; c++) u[c] = table16[idx[2 * c + 1]];
    }
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        const uint2 a = *reinterpret_cast<const uint2*>(table16 + (idx[c] & ~1u));
        const bool hi = (idx[c] & 1u) != 0u;
        v[c] = hi ? a.y : a.x;
        v[c + 1] = paired ? (hi ? a.x : a.y) : u[c >> 1];
    }
#endif
    r0 = 0.0f; r1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const half2v hv = __builtin_bit_cast(half2v, v[c]);
        r0 = __builtin_fmaf(w8[c], (float)hv[0], r0);
        r1 = __builtin_fmaf(w8[c], (float)hv[1], r1);
    }
}

// 16 lanes per sample, one level each (64 contiguous bytes of features per sample); lane 0 also writes the direction encoding
template <int DIR>
__global__ __launch_bounds__(256) void k_encode_hash(const float* __restrict__ in, const uint32_t* __restrict__ table16,
                                                    half_t* __restrict__ feat, uint32_t n, HashLevels lv, int skip_zero)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int ND = DIR == 1 ? 2 : 8, E16 = 48;
    // skip_zero & 2 (round 4): a level's table is gathered from ONE XCD -- workgroup b (on XCD b & 7) takes the levels (b & 7) and (b & 7) + 8
    // of 128 samples (see k_encode_hash_list); otherwise 16 lanes per sample, one level each
    uint32_t sample, level;
    if (skip_zero & 2) {
        sample = (blockIdx.x >> 3) * 128u + (threadIdx.x >> 1);
        level = (blockIdx.x & 7u) + 8u * (threadIdx.x & 1u);
    } else {
        const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
        sample = gid >> 4; level = gid & 15u;
    }
    skip_zero &= 1;
    if (sample >= n) return;
    const float* p = in + (size_t)sample * 5u;
    const float x[3] = {p[0], p[1], p[2]};
    float r0 = 0.0f, r1 = 0.0f;
    // renderer inference: the query of a pixel that did not scatter is all zero and its output is never read -- no gathers
    const bool unused = skip_zero != 0 && x[0] == 0.0f && x[1] == 0.0f && x[2] == 0.0f && p[3] == 0.0f && p[4] == 0.0f;
    if (!unused) {
        uint32_t idx[8];
        float w8[8];
        hg_corners(lv, level, x, idx, w8);
        hg_gather(table16, idx, w8, r0, r1);
    }
    half_t* o = feat + (size_t)sample * E16;
    o[2 * level] = (half_t)r0;
    o[2 * level + 1] = (half_t)r1;
    if (level == 0) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
            if (DIR == 0) {
                float b[4];
                oneblob4(p[3 + d], b);
#pragma unroll
                for (int q = 0; q < 4; q++) o[32 + 4 * d + q] = (half_t)b[q];
            } else if (DIR == 1) {
                o[32 + d] = (half_t)p[3 + d];
            } else {
#pragma unroll
                for (int f = 0; f < 4; f++) o[32 + 4 * d + f] = (half_t)tri_wave(p[3 + d], f);
            }
        }
#pragma unroll
        for (int k = 32 + ND; k < E16; k++) o[k] = (half_t)1.0f;
    }
}

// Inference variant, level-major: blockIdx.y = feature slot (0..15 hash levels, 16.. direction features / padding, one half2
// each), a thread = one sample.  Workgroups are dispatched slot by slot, so the chip gathers from ONE level's table at a time
// (2 MB at the default 2^19 entries: resident in every XCD's 4 MB L2) instead of from all 16 at once (28 MB, every gather a
// fabric round trip for 4 useful bytes).  Output layout [slot][n] half2 -- coalesced stores; k_infer_gen<..., true> builds its
// layer-0 operands from it.
template <int DIR>
__global__ __launch_bounds__(256) void k_encode_hash_lm(const float* __restrict__ in, const uint32_t* __restrict__ table16,
                                                       uint32_t* __restrict__ feat_lm, uint32_t n, HashLevels lv, int skip_zero)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int ND = DIR == 1 ? 2 : 8;
    const uint32_t sample = blockIdx.x * 256u + threadIdx.x, slot = blockIdx.y;
    if (sample >= n) return;
    const float* p = in + (size_t)sample * 5u;
    const float x[3] = {p[0], p[1], p[2]};
    const float d0 = p[3], d1 = p[4];
    const bool unused = skip_zero != 0 && x[0] == 0.0f && x[1] == 0.0f && x[2] == 0.0f && d0 == 0.0f && d1 == 0.0f;
    if (unused) return;                  // unscattered pixel: its column of the MLP is never read
    float r0 = 1.0f, r1 = 1.0f;          // padding slots hold ones
    if (slot < HG_LEVELS) {
        uint32_t idx[8];
        float w8[8];
        hg_corners(lv, slot, x, idx, w8);
        hg_gather(table16, idx, w8, r0, r1);
    } else {
        const int k0 = 2 * ((int)slot - (int)HG_LEVELS);         // direction feature index of r0 (r1 = k0 + 1)
        if (k0 < ND) {
            if (DIR == 0) {
                float b[4];
                oneblob4(k0 < 4 ? d0 : d1, b);
                r0 = b[k0 & 3]; r1 = b[(k0 & 3) + 1];
            } else if (DIR == 1) {
                r0 = d0; r1 = d1;
            } else {
                const float v = k0 < 4 ? d0 : d1;
                r0 = tri_wave(v, k0 & 3); r1 = tri_wave(v, (k0 & 3) + 1);
            }
        }
    }
    float2v f = {r0, r1};
    feat_lm[(size_t)slot * n + sample] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
}

// The same over a LIST of live queries (renderer inference: k_gen_rays lists the pixels that scattered, DevFrame::live_list).  The
// one-thread-per-(query, slot) launch above starts 41 M threads for a 1080p frame of which 78 % read a 20-byte query, find it all zero
// and leave -- 650 k waves and 0.8 GB of query reads beside gen_rays, whose launch went from 0.24 to 0.6 ms with them in the way.  Here a
// fixed grid of workgroups per slot strides over the list; dead pixels cost nothing.
template <int DIR>
__global__ __launch_bounds__(256) void k_encode_hash_list(const float* __restrict__ in, const uint32_t* __restrict__ table16,
                                                         uint32_t* __restrict__ feat_lm, uint32_t n, HashLevels lv,
                                                         const uint32_t* __restrict__ live_list, const uint32_t* __restrict__ live_count,
                                                         uint32_t xcd_chunks)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int ND = DIR == 1 ? 2 : 8;
    // Which (slot, chunk of the list) a workgroup takes.  gridDim.y == 1 (round 4): a level's table is gathered from ONE XCD.  Workgroups are
    // dealt to the eight XCDs round-robin by their linear index, so workgroup b runs on XCD b & 7; it takes the slots x, x + 8, x + 16, ... of
    // its XCD x in turn (b >> 3 = chunk + chunks * turn): every XCD's L2 then holds the one or two 2 MB tables of its own levels for the whole
    // launch instead of all sixteen one after the other (28 MB through each of the eight 4 MB L2s per frame, evicting the volume gen_rays
    // reads beside it).  gridDim.y == n_slots: the level-major order of the dense variant above (every XCD gathers from every level).
    uint32_t slot, chunk, chunks;
    if (gridDim.y == 1u) {
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
        chunks = xcd_chunks;
        slot = xcd + 8u * (k / chunks);
        chunk = k % chunks;
    } else {
        slot = blockIdx.y; chunk = blockIdx.x; chunks = gridDim.x;
    }
    const uint32_t count = min(*live_count, n);
    for (uint32_t i = chunk * 256u + threadIdx.x; i < count; i += chunks * 256u) {
        const uint32_t sample = live_list[i];
        if (sample >= n) continue;           // (a list of another launch: never written here)
        const float* p = in + (size_t)sample * 5u;
        float r0 = 1.0f, r1 = 1.0f;          // padding slots hold ones
        if (slot < HG_LEVELS) {
            const float x[3] = {p[0], p[1], p[2]};
            uint32_t idx[8];
            float w8[8];
            hg_corners(lv, slot, x, idx, w8);
            hg_gather(table16, idx, w8, r0, r1);
        } else {
            const float d0 = p[3], d1 = p[4];
            const int k0 = 2 * ((int)slot - (int)HG_LEVELS);         // direction feature index of r0 (r1 = k0 + 1)
            if (k0 < ND) {
                if (DIR == 0) {
                    float b[4];
                    oneblob4(k0 < 4 ? d0 : d1, b);
                    r0 = b[k0 & 3]; r1 = b[(k0 & 3) + 1];
                } else if (DIR == 1) {
                    r0 = d0; r1 = d1;
                } else {
                    const float v = k0 < 4 ? d0 : d1;
                    r0 = tri_wave(v, k0 & 3); r1 = tri_wave(v, (k0 & 3) + 1);
                }
            }
        }
        float2v f = {r0, r1};
        feat_lm[(size_t)slot * n + sample] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2v));
    }
}

// ---- dL/d(table): every (sample, level) sends weight * dL/d(feature) -- rounded to fp16 pairs, as tiny-cuda-nn's packed atomics do (the
// values carry the loss scale) -- to its 8 corners.
// Round 4 (k_grid_scatter + k_grid_gather): no memory-side float atomics for the large levels.  2.1 M packed-fp16 atomics per step took 112 us
// alone and cost the frame beside them 41 %.  A level of at least 8 x 2 048 entries is cut into BINS of 2 048 entries (4 096 in round 4): pass 1 (one level and
// 256 samples per workgroup) appends its (entry, value) pairs to the level's bin lists -- a per-workgroup LDS histogram, ONE global atomic per
// (workgroup, non-empty bin) to reserve the run --, pass 2 (one workgroup per bin) adds a bin's pairs into accumulators in LDS and stores the
// entries that were touched.  The bin size is a residency matter: with bins of 16 384 entries (128 KB of LDS per gather workgroup) the frame
// gained 2.6 %, with 4 096 another 6 % -- the workgroups find room on CUs that gen_rays occupies.
// Round 5: the sums are EXACT, hence independent of the order the pairs arrive in -- the table gradient, and with it HashGrid training, is
// bitwise repeatable (test_backward_is_bitwise_reproducible[hashgrid]).  An fp16 value is an integer multiple of 2^-24 below
// 2^16: as a 64-bit fixed-point number it adds without rounding, and integer addition is associative.  The LDS accumulators are int64 (bins
// of 2 048 entries now: the same 32 KB), what does not go through a bin list -- the levels too small for bins, pairs
// beyond a list's capacity -- is added with 64-bit integer atomics into a fixed-point shadow of the table (all zero between two steps: the
// gather pass folds it into the entries it finishes and clears it), and every entry is rounded ONCE, from its exact sum, to the fp16 pair
// the optimizer and the list exchange read (round 4 summed in fp32 in arrival order; tiny-cuda-nn's atomics round every partial sum).
// A list holds twice the pairs the level sends a bin on average (a dense coarse level's bins are crowded: 16 384 pairs each for 32^3).
#ifndef NRC_GB_BIN_LOG2
#define NRC_GB_BIN_LOG2 11
#endif
#ifndef NRC_GB_GATHER_THREADS
#define NRC_GB_GATHER_THREADS 512
#endif
constexpr uint32_t GB_BIN_LOG2 = NRC_GB_BIN_LOG2, GB_BIN = 1u << GB_BIN_LOG2, GB_MAX_BINS = 1u << (20 - NRC_GB_BIN_LOG2), GB_MIN_BINS = 8;
struct GridBins {
    uint32_t first[HG_LEVELS], count[HG_LEVELS];      // per level: index of its first bin, number of bins (0: no lists, the level adds into the shadow)
    uint32_t cap[HG_LEVELS], list0[HG_LEVELS];        // pairs a bin's list holds; pair offset of the level's first list
};
// fp16 bits -> value * 2^24 (exact; inf / nan read as 2^15 * 2^25: deterministic, and the loss is not finite then anyway)
__device__ __forceinline__ long long half_to_fix(uint32_t h)
{
    const uint32_t e = (h >> 10) & 31u, f = h & 1023u;
    const unsigned long long m = e ? (1024u | f) : f;
    const long long v = (long long)(m << (e ? e - 1u : 0u));
    return (h & 0x8000u) ? -v : v;
}
// the exact 64-bit sum -> fp16, rounded ONCE (int64 -> fp32 -> fp16 were two roundings, and a sum within half an fp32 ulp of an
// fp16 tie could land on the wrong side).  The fp32 step rounds to ODD -- truncate, then set the last bit if anything was cut off -- which
// keeps every sum strictly between two fp16 candidates strictly between them, and exactly representable sums exact; fp32's 24 bits are 13
// more than fp16 keeps, so the final round-to-nearest-even sees what the infinitely precise value would have shown it.
__device__ __forceinline__ float fix_to_float_odd(long long a)
{
    const float f = __ll2float_rz(a);
    const uint32_t sticky = ((long long)f != a) ? 1u : 0u;      // (|f| <= |a| < 2^63: the conversion back is exact)
    return __uint_as_float(__float_as_uint(f) | sticky) * 5.9604644775390625e-8f;      // x 2^-24 (exact)
}
__device__ __forceinline__ uint32_t fix_to_half2(long long a0, long long a1)
{
    const float2v a = {fix_to_float_odd(a0), fix_to_float_odd(a1)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(a, half2v));
}
__device__ __forceinline__ void fix_add_global(long long* __restrict__ fix, uint32_t entry, uint32_t hbits2)
{
    const uint32_t h0 = hbits2 & 0xffffu, h1 = hbits2 >> 16;
    if ((h0 & 0x7fffu) != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(fix + 2 * (size_t)entry), (unsigned long long)half_to_fix(h0));
    if ((h1 & 0x7fffu) != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(fix + 2 * (size_t)entry + 1), (unsigned long long)half_to_fix(h1));
}
__global__ __launch_bounds__(256) void k_grid_scatter(const float* __restrict__ in, const half_t* __restrict__ d_enc, long long* __restrict__ fix,
                                                     uint32_t n, HashLevels lv, GridBins gb, uint32_t* __restrict__ counters,
                                                     uint2* __restrict__ lists)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    __shared__ uint32_t s_cnt[GB_MAX_BINS], s_base[GB_MAX_BINS];
    // gridDim.y == 1: a level's lists are appended to from ONE XCD (workgroup b runs on XCD b & 7 and takes level (b & 7) + 8 * turn, see
    // k_encode_hash_list): the partial lines of a list's neighbouring runs meet in one L2
    uint32_t level, sample;
    if (gridDim.y == 1u) {
        const uint32_t k = blockIdx.x >> 3, chunks = gridDim.x / HG_LEVELS;
        level = (blockIdx.x & 7u) + 8u * (k / chunks);
        sample = (k % chunks) * 256u + threadIdx.x;
    } else {
        level = blockIdx.y; sample = blockIdx.x * 256u + threadIdx.x;
    }
    const uint32_t nb = gb.count[level];              // (workgroup-uniform)
    for (uint32_t i = threadIdx.x; i < GB_MAX_BINS; i += 256u) s_cnt[i] = 0u;
    __syncthreads();
    bool act = sample < n;
    float de0 = 0.0f, de1 = 0.0f;
    if (act) {
        de0 = (float)d_enc[(size_t)sample * 32u + 2u * level];
        de1 = (float)d_enc[(size_t)sample * 32u + 2u * level + 1u];
        act = !(de0 == 0.0f && de1 == 0.0f);
    }
    uint32_t idx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
        const float* p = in + (size_t)sample * 5u;
        const float x[3] = {p[0], p[1], p[2]};
        hg_corners(lv, level, x, idx, w8);
    }
    if (nb != 0u) {
        if (act) {
#pragma unroll
            for (int c = 0; c < 8; c++) slot[c] = atomicAdd(&s_cnt[(idx[c] - lv.off[level]) >> GB_BIN_LOG2], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += 256u) {
            const uint32_t k = s_cnt[i];
            s_base[i] = k != 0u ? atomicAdd(&counters[gb.first[level] + i], k) : 0u;
        }
        __syncthreads();
    }
    if (!act) return;
    const uint32_t cap = gb.cap[level];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float2v g = {w8[c] * de0, w8[c] * de1};
        const uint32_t gh = __builtin_bit_cast(uint32_t, __builtin_convertvector(g, half2v));
        if (nb != 0u) {
            const uint32_t bin = (idx[c] - lv.off[level]) >> GB_BIN_LOG2;
            const uint32_t pos = s_base[bin] + slot[c];
            if (pos < cap) {      // (which pairs of a crowded bin miss its list depends on the order of arrival; the entry's exact sum does not)
                lists[(size_t)gb.list0[level] + (size_t)bin * cap + pos] = make_uint2(idx[c], gh);
                continue;
            }
        }
        fix_add_global(fix, idx[c], gh);
    }
}
// pass 2.  Workgroups 0 .. n_bins - 1: one bin each -- bin_info[b] = {table entry of the bin's first slot, pair offset of its list, the list's
// capacity, 0}; resets the bin's counter for the next step.  The workgroups behind them walk the entries of the levels WITHOUT bins
// (loose[k] = {first entry, count}, n_loose ranges, GB_BIN entries per workgroup) and turn their shadow sums into the gradient.
struct LooseRanges {
    uint32_t first[HG_LEVELS], count[HG_LEVELS], n;
};
__global__ __launch_bounds__(NRC_GB_GATHER_THREADS) void k_grid_gather(uint32_t* __restrict__ grad16, const uint4* __restrict__ bin_info, uint32_t n_bins,
                                                    uint32_t* __restrict__ counters, const uint2* __restrict__ lists, long long* __restrict__ fix,
                                                    LooseRanges loose)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    extern __shared__ long long s_acc[];      // [GB_BIN][2]
    const uint32_t b = blockIdx.x;
    if (b >= n_bins) {      // entries of the levels without bins: shadow -> gradient
        uint32_t chunk = b - n_bins;
        for (uint32_t r = 0; r < loose.n; r++) {
            const uint32_t chunks = (loose.count[r] + GB_BIN - 1u) / GB_BIN;
            if (chunk < chunks) {
                const uint32_t end = loose.first[r] + loose.count[r];
                for (uint32_t e = loose.first[r] + chunk * GB_BIN + threadIdx.x; e < min(end, loose.first[r] + (chunk + 1u) * GB_BIN); e += (uint32_t)NRC_GB_GATHER_THREADS) {
                    const long long a0 = fix[2 * (size_t)e], a1 = fix[2 * (size_t)e + 1];
                    if ((a0 | a1) == 0) continue;
                    fix[2 * (size_t)e] = 0;
                    fix[2 * (size_t)e + 1] = 0;
                    grad16[e] = fix_to_half2(a0, a1);
                }
                return;
            }
            chunk -= chunks;
        }
        return;
    }
    const uint32_t total = counters[b];
    if (total == 0u) return;              // (workgroup-uniform; nothing was added to the shadow for this bin either: a list fills before it overflows)
    const uint4 info = bin_info[b];
    const uint32_t e0 = info.x, cap = info.z;
    const uint32_t cnt = total < cap ? total : cap;
    const bool overflow = total > cap;    // some of the bin's pairs went to the shadow
    const uint2* L = lists + info.y;
    for (uint32_t i = threadIdx.x; i < GB_BIN * 2u; i += (uint32_t)NRC_GB_GATHER_THREADS) s_acc[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += (uint32_t)NRC_GB_GATHER_THREADS) {
        const uint2 pr = L[i];
        const uint32_t li = (pr.x - e0) & (GB_BIN - 1u);
        const uint32_t h0 = pr.y & 0xffffu, h1 = pr.y >> 16;
        if ((h0 & 0x7fffu) != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[2u * li]), (unsigned long long)half_to_fix(h0));
        if ((h1 & 0x7fffu) != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(&s_acc[2u * li + 1u]), (unsigned long long)half_to_fix(h1));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < GB_BIN; i += (uint32_t)NRC_GB_GATHER_THREADS) {
        const uint32_t e = e0 + i;
        long long a0 = s_acc[2u * i], a1 = s_acc[2u * i + 1u];
        if (overflow) {
            const long long f0 = fix[2 * (size_t)e], f1 = fix[2 * (size_t)e + 1];
            if ((f0 | f1) != 0) {
                fix[2 * (size_t)e] = 0;
                fix[2 * (size_t)e + 1] = 0;
                a0 += f0;
                a1 += f1;
            }
        }
        if ((a0 | a1) == 0) continue;
        grad16[e] = fix_to_half2(a0, a1);
    }
    if (threadIdx.x == 0u) counters[b] = 0u;
}

__global__ void k_grid_grad_f32(const uint32_t* __restrict__ grad16, float* __restrict__ grad, uint32_t n_entries)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const half2v h = __builtin_bit_cast(half2v, grad16[i]);
    grad[2 * (size_t)i] = (float)h[0];
    grad[2 * (size_t)i + 1] = (float)h[1];
}

// ---- sparse multi-GPU exchange of the table gradient.  A training batch touches a small part of the table (a rank's share of
// 16 384 rays x 16 levels x 8 corners against 7 M entries), so instead of all-reducing the dense fp32 vector every rank packs the
// entries its batch touched into a list {count, 0, (entry, half2 bits) x capacity}, the lists are all-gathered, and every rank
// adds them into a zeroed fp32 gradient in RANK ORDER: an entry occurs at most once per list, so a list is applied without
// atomics, the sum of an entry is ((0 + r0) + r1) + ... on every rank, and the replicas stay bit-identical.
// (a wave packs 1 024 entries and reserves its list slots with ONE atomic: an atomic per 64 entries -- 111 k same-address atomics from
// eight XCDs -- took 1.04 ms for the 7.1 M-entry table)
__global__ __launch_bounds__(256) void k_grid_pack(const uint32_t* __restrict__ grad16, uint32_t n_entries,
                                                  uint32_t* __restrict__ list, uint32_t cap)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t e0 = wave * 1024u;
    uint32_t w[16], cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t e = e0 + (uint32_t)k * 256u + lane * 4u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (e + 3u < n_entries) v = *reinterpret_cast<const uint4*>(grad16 + e);
        else {
            if (e < n_entries) v.x = grad16[e];
            if (e + 1u < n_entries) v.y = grad16[e + 1u];
            if (e + 2u < n_entries) v.z = grad16[e + 2u];
        }
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        cnt += (v.x != 0u) + (v.y != 0u) + (v.z != 0u) + (v.w != 0u);      // untouched entries still hold the memset's zero
    }
    uint32_t scan = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(scan, off);
        if ((int)lane >= off) scan += up;
    }
    const uint32_t total = __shfl(scan, 63);
    if (total == 0u) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&list[0], total);
    uint32_t pos = __shfl(base, 0) + scan - cnt;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (w[k] != 0u) {
            if (pos < cap) {
                list[2u + 2u * pos] = e0 + (uint32_t)(k >> 2) * 256u + lane * 4u + (uint32_t)(k & 3);
                list[3u + 2u * pos] = w[k];
            }
            pos++;
        }
    }
}

__global__ __launch_bounds__(256) void k_grid_apply(const uint32_t* __restrict__ list, uint32_t cap, float* __restrict__ grad,
                                                   uint32_t n_entries)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= min(list[0], cap)) return;
    const uint32_t e = list[2u + 2u * i];
    if (e >= n_entries) return;
    const half2v h = __builtin_bit_cast(half2v, list[3u + 2u * i]);
    grad[2 * (size_t)e] += (float)h[0];
    grad[2 * (size_t)e + 1] += (float)h[1];
}
#pragma clang fp contract(fast)

// fp16 gather copies of the table: training weights and EMA weights (half2 per entry)
__global__ void k_pack_grid(const float* __restrict__ w, const float* __restrict__ ema, uint32_t* __restrict__ t_train,
                            uint32_t* __restrict__ t_ema, uint32_t n_entries)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    float2v a = {w[2 * (size_t)i], w[2 * (size_t)i + 1]}, b = {ema[2 * (size_t)i], ema[2 * (size_t)i + 1]};
    t_train[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, half2v));
    t_ema[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, half2v));
}

__device__ __forceinline__ half8 ld_frag_g(const uint4* __restrict__ img, int frag, int lane)
{
    const uint4 v = img[frag * 64 + lane];
    return __builtin_bit_cast(half8, v);
}

// Generic inference (any encoding, width 64 / 128, any depth).  The weight image of an 8x128 net (250 KB) does not fit a
// workgroup's LDS, and fetching every fragment from L2 per 32-sample tile made the kernel L2-bound (7.8 KB per sample).  Here a
// workgroup (4 waves x NT tiles = 256 samples) streams the image through LDS one layer at a time: while the waves run layer
// l's MFMAs out of one buffer, layer l+1 arrives in the other one by direct global->LDS loads issued before the MFMAs -- one
// barrier per layer, no staging registers, 0.5-1 KB of L2 traffic per sample.  Same accumulator-as-
// operand chain and numerics as k_infer.  skip_in (renderer inference): tiles whose 32 queries are all zero are not computed,
// a workgroup whose 256 queries are all zero does not even stream the weights.
#ifndef NRC_GEN128_WPS
#define NRC_GEN128_WPS 1
#endif
#ifndef NRC_GEN128_THREADS
#define NRC_GEN128_THREADS 512
#endif
#ifndef NRC_GEN128_WG4_PER_CU
#define NRC_GEN128_WG4_PER_CU 2      // persistent 4-wave workgroups per CU of the renderer-mode 128-wide inference launch
#endif
// ENC80: the input is the raw 5-float query and the Frequency(12) + OneBlob(4) encoding is computed here, k-step by k-step, as
// k_infer does (no k_encode pass, no 160 B/sample feature buffer); feat is unused, raw_in required, the image is in fmap80 order.
template <int WIDTH, int THREADS, bool FEAT_LM, int NT = 2, bool ENC80 = false>
__global__ __launch_bounds__(THREADS, WIDTH == 128 && NRC_GEN128_WPS > 2 ? NRC_GEN128_WPS : 2) void k_infer_gen(const half_t* __restrict__ feat, float* __restrict__ out, uint32_t n,
                                                      const uint4* __restrict__ img, int depth, int ks0,
                                                      const float* __restrict__ skip_in, const float* __restrict__ raw_in = nullptr,
                                                      const uint32_t* __restrict__ live_list = nullptr,
                                                      const uint32_t* __restrict__ live_count = nullptr)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int MTG = WIDTH / 32, KSG = WIDTH / 16, WAVES = THREADS / 64;
    constexpr int STAGE_FRAGS = MTG * (KSG > 5 ? KSG : 5);        // largest stage (layer 0 has ks0 <= 5 k-steps)
    constexpr int PF = (STAGE_FRAGS * 64 + THREADS - 1) / THREADS;      // uint4 per thread per stage
    extern __shared__ uint4 lds_w[];                              // [2][STAGE_FRAGS * 64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // live_list (renderer inference, round 4): the frame's live queries as a dense list (k_gen_rays appends the query index of every pixel
    // that scattered, DevFrame::live_list) -- the tiles are 32 LIST entries, so every tile that runs is full but the last one; without it
    // a tile is 32 consecutive queries = four rows of eight pixels, computed whole when any of them is live (68 % of the computed
    // queries are live on configs[4]'s smoke plume, 87 % on the default cloud: tools/live_tiles.py).  Results are per query: the
    // (unordered) list changes no bit.
    const bool listed = live_list != nullptr;
    uint32_t n_q = n;
    if (listed) {
        n_q = __builtin_amdgcn_readfirstlane((int)*live_count);
        n_q = n_q < n ? n_q : n;
    }
    const uint32_t n_tiles = (n_q + 31u) >> 5;
    // the query lane r of tile slot sx works on (slots beyond the end repeat the last one: in bounds, never stored)
    auto query_of = [&](uint32_t sx) -> uint32_t {
        const uint32_t c = sx < n_q ? sx : n_q - 1u;
        return listed ? live_list[c] : c;
    };
    const uint32_t n_groups = (n_tiles + WAVES * NT - 1u) / (WAVES * NT);
    const uint32_t e16 = (uint32_t)ks0 * 16u;
    const int hid_base = MTG * ks0;
    // stage st: layer-0 fragments | hidden layer st | output fragments; first fragment and size (in uint4) in the image
#define NRC_STAGE_FIRST(st) ((st) == 0 ? 0 : hid_base + ((st) - 1) * MTG * KSG)
#define NRC_STAGE_COUNT(st) (((st) == 0 ? MTG * ks0 : ((st) == depth ? KSG : MTG * KSG)) * 64)
    // Stage prefetch through registers: the global loads of stage st+1 are issued at the top of stage st (PF x 16 B per thread),
    // their data is written to the other LDS buffer at the bottom, in front of the barrier.  (The direct global->LDS loads used
    // before -- global_load_lds_dwordx4, no staging registers -- did not overlap anything: the compiler cannot tell which LDS
    // buffer an LDS-DMA writes and put s_waitcnt vmcnt(0) in front of the first ds_read of the CURRENT stage, so every layer paid
    // the L2 latency of the next one: 4 700 cycles per layer instead of the 2 048 its MFMAs need.)
    uint4v pf[PF];
#define NRC_STAGE_FETCH(st)                                                                  \
    do {                                                                                     \
        const uint4* src_ = img + (size_t)NRC_STAGE_FIRST(st) * 64;                          \
        const int cnt_ = NRC_STAGE_COUNT(st);                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < PF; k_++) {                                  \
            const int i_ = k_ * THREADS + (int)threadIdx.x;                                  \
            if (i_ < cnt_) pf[k_] = reinterpret_cast<const uint4v*>(src_)[i_];                  \
        }                                                                                    \
    } while (0)
#define NRC_STAGE_COMMIT(st, buf)                                                            \
    do {                                                                                     \
        uint4* dst_ = lds_w + (buf) * (STAGE_FRAGS * 64);                                    \
        const int cnt_ = NRC_STAGE_COUNT(st);                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < PF; k_++) {                                  \
            const int i_ = k_ * THREADS + (int)threadIdx.x;                                  \
            if (i_ < cnt_) reinterpret_cast<uint4v*>(dst_)[i_] = pf[k_];                        \
        }                                                                                    \
    } while (0)

    // layer-0 operands (the encoded features of this wave's samples, ks0 <= 5 k-steps): all loaded at the top of the group --
    // fetched inside the k-loop, each of the five load -> MFMA rounds exposed a full memory latency, a quarter of a group's time
    half8 f0[NT][5];
    auto load_features = [&](uint32_t g) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const size_t si = query_of(((g * WAVES + (uint32_t)wave) * NT + (uint32_t)t) * 32u + (uint32_t)r);
            if constexpr (ENC80) {
                const float* qv = raw_in + si * 5u;
                const float x[5] = {qv[0], qv[1], qv[2], qv[3], qv[4]};
#pragma unroll
                for (int k = 0; k < 5; k++) f0[t][k] = encode80_step(x, h, k);
                continue;
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int kc = k < ks0 ? k : ks0 - 1;      // k-steps beyond ks0: load the last one again, use zeros -- no branches
                if (FEAT_LM) {           // [slot][n] half2 (k_encode_hash_lm): features 16k+8h+2i, +1 = slot 8k+4h+i
                    const uint32_t* fl = reinterpret_cast<const uint32_t*>(feat) + (size_t)(8 * kc + 4 * h) * n + si;
                    uint4v v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fl[(size_t)i * n];
                    f0[t][k] = __builtin_bit_cast(half8, v);
                } else {
                    f0[t][k] = *reinterpret_cast<const half8*>(feat + si * e16 + 8 * h + 16 * kc);
                }
                if (k >= ks0) f0[t][k] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };

    int q = 0;                       // stage counter: stage data lives in buffer q & 1
    NRC_STAGE_FETCH(0);
    NRC_STAGE_COMMIT(0, 0);
    __syncthreads();
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        uint32_t sidx[NT];
        bool valid[NT], run[NT], mine[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const uint32_t tile = (group * WAVES + (uint32_t)wave) * NT + (uint32_t)t;
            valid[t] = tile * 32u + (uint32_t)r < n_q;
            sidx[t] = n_q != 0u ? query_of(tile * 32u + (uint32_t)r) : 0u;
            bool used = valid[t];
            if (!listed && skip_in != nullptr && valid[t]) {
                const float* qv = skip_in + (size_t)sidx[t] * 5u;
                used = qv[0] != 0.0f || qv[1] != 0.0f || qv[2] != 0.0f || qv[3] != 0.0f || qv[4] != 0.0f;
            }
            mine[t] = used;                           // this lane's own query is live
            run[t] = __ballot(used) != 0ull;          // wave-uniform
        }
        bool wave_runs = false;
#pragma unroll
        for (int t = 0; t < NT; t++) wave_runs |= run[t];
        if (__syncthreads_or(wave_runs ? 1 : 0) == 0) continue;       // nothing to do for this group: stage 0 stays staged
        if (wave_runs) load_features(group);

        half8 b[NT][KSG];                 // the only state carried from stage to stage
#pragma unroll 1
        for (int st = 0; st <= depth; st++) {
            const uint4* lw = lds_w + (q & 1) * (STAGE_FRAGS * 64);
            const bool last = st == depth;
            const bool more = !last || group + gridDim.x < n_groups;       // next stage to stream (stage 0 of the next group)
            const int next = last ? 0 : st + 1;
            if (more) NRC_STAGE_FETCH(next);
            if (wave_runs) {
                if (st == 0) {
                    f32x16 acc[NT][MTG];
                    half8 fr0[2][MTG];                 // the k-step's fragments, read one k-step ahead
#pragma unroll
                    for (int m = 0; m < MTG; m++) fr0[0][m] = ld_frag(lw, m * ks0, lane);
#pragma unroll
                    for (int k = 0; k < 5; k++) {      // always five k-steps: beyond ks0 the operand is zero (and the fragment a repeat)
                        if (k + 1 < 5) {
#pragma unroll
                            for (int m = 0; m < MTG; m++) fr0[(k + 1) & 1][m] = ld_frag(lw, m * ks0 + (k + 1 < ks0 ? k + 1 : ks0 - 1), lane);
                        }
#pragma unroll
                        for (int m = 0; m < MTG; m++) {
#pragma unroll
                            for (int t = 0; t < NT; t++) acc[t][m] = mfma(fr0[k & 1][m], f0[t][k], k == 0 ? zero16() : acc[t][m]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int m = 0; m < MTG; m++) relu_pack(acc[t][m], b[t][2 * m], b[t][2 * m + 1]);
                } else if (!last) {
                    if constexpr (WIDTH == 128) {
                        // one 32-row block at a time, converted as soon as the next block's MFMAs are issued: two accumulators
                        // live instead of four (the four-block form needed 167 VGPRs and spilled under a 128-register cap, so a
                        // workgroup only fitted on a CU that gen_rays had all but left)
                        // The fragments come through a two-deep ring of four (a "region" = four k-steps of one block): the reads of
                        // region i+1 are issued in front of region i's MFMAs.  With only the current region's reads in flight every
                        // region began with an exposed LDS latency (~150 cycles per 128 cycles of MFMA: the pipes were 43 % busy).
                        half8 bn[NT][KSG];
                        f32x16 acc[NT][2];
                        half8 fr[2][4];
                        constexpr int REGIONS = MTG * (KSG / 4);
#pragma unroll
                        for (int j = 0; j < 4; j++) fr[0][j] = ld_frag(lw, j, lane);
#pragma unroll
                        for (int i = 0; i < REGIONS; i++) {
                            const int m = i / (KSG / 4), k0 = (i % (KSG / 4)) * 4;
                            if (i + 1 < REGIONS) {
#pragma unroll
                                for (int j = 0; j < 4; j++) fr[(i + 1) & 1][j] = ld_frag(lw, (i + 1) * 4 + j, lane);      // = m' * KSG + k'
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) {
#pragma unroll
                                for (int t = 0; t < NT; t++)
                                    acc[t][m & 1] = mfma(fr[i & 1][j], b[t][k0 + j], k0 + j == 0 ? zero16() : acc[t][m & 1]);
                            }
                            if (k0 == 0 && m > 0) {      // the block before this one is complete: convert it behind this block's first MFMAs
#pragma unroll
                                for (int t = 0; t < NT; t++) relu_pack(acc[t][(m - 1) & 1], bn[t][2 * (m - 1)], bn[t][2 * (m - 1) + 1]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int t = 0; t < NT; t++) relu_pack(acc[t][(MTG - 1) & 1], bn[t][2 * (MTG - 1)], bn[t][2 * (MTG - 1) + 1]);
#pragma unroll
                        for (int t = 0; t < NT; t++)
#pragma unroll
                            for (int k = 0; k < KSG; k++) b[t][k] = bn[t][k];
                    } else {
                        f32x16 acc[NT][MTG];
#pragma unroll
                        for (int m = 0; m < MTG; m++) {
#pragma unroll
                            for (int t = 0; t < NT; t++) acc[t][m] = zero16();
#pragma unroll
                            for (int k = 0; k < KSG; k++) {
                                const half8 a = ld_frag(lw, m * KSG + k, lane);
#pragma unroll
                                for (int t = 0; t < NT; t++) acc[t][m] = mfma(a, b[t][k], acc[t][m]);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < NT; t++)
#pragma unroll
                            for (int m = 0; m < MTG; m++) relu_pack(acc[t][m], b[t][2 * m], b[t][2 * m + 1]);
                    }
                } else {
                    f32x16 y[NT];
#pragma unroll
                    for (int t = 0; t < NT; t++) y[t] = zero16();
#pragma unroll
                    for (int k = 0; k < KSG; k++) {
                        const half8 a = ld_frag(lw, k, lane);
#pragma unroll
                        for (int t = 0; t < NT; t++) y[t] = mfma(a, b[t][k], y[t]);
                    }
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        if (mine[t] && h == 0) {      // unscattered pixels keep their previous output (their features were skipped)
                            float* o = out + (size_t)sidx[t] * 3u;
                            o[0] = y[t][0];
                            o[1] = y[t][1];
                            o[2] = y[t][2];
                        }
                    }
                }
            }
            if (more) NRC_STAGE_COMMIT(next, (q + 1) & 1);
            __syncthreads();
            q++;
        }
    }
#undef NRC_STAGE_FIRST
#undef NRC_STAGE_COUNT
#undef NRC_STAGE_FETCH
#undef NRC_STAGE_COMMIT
}

struct TrainArgsGen {
    const half_t* feat;   // [n][e16]
    const float* target;
    uint32_t n;
    float inv_n_total;
    uint32_t loss_id;
    half_t* acts;         // [n/8][e16 + depth*WIDTH][8]
    half_t* deltas;       // [n/8][depth*WIDTH + 8][8]
    float* loss_part;
    int depth, ks0;
    half_t* d_enc;        // [n][32] dL/d(first 32 encoded dims) for a trainable encoding (HashGrid), or nullptr
};

// Generic training forward + loss + dgrad (any encoding, width 32 / 64 / 128, any depth).  Like k_infer_gen the workgroup streams the
// weight images through LDS one layer at a time -- the forward image front to back, then the transposed (W^T) image back to
// front -- with the next stage arriving by direct global->LDS loads while the current one feeds the MFMAs: every fragment is
// fetched from L2 once per workgroup (WAVES tiles) instead of once per tile (a wave used to pull all 500 KB of an 8x128 net's
// two images through its own dependent loads: 284 us for a 16 384-ray batch beside gen_rays).  One tile per wave; activations and
// deltas leave for HBM in the k-group layout the weight-gradient GEMM reads, as before.
//   stages: 0 layer 0 | 1..depth-1 hidden layers | depth output layer | depth+1 output layer^T | depth+1+j hidden layer^T
//   l = depth-j (j = 1..depth-1) | 2*depth+1 layer 0^T rows of a trainable encoding
template <int WIDTH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_train_gen(TrainArgsGen a, const uint4* __restrict__ img_fwd,
                                                         const uint4* __restrict__ img_bwd)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int MTG = WIDTH / 32, KSG = WIDTH / 16, THREADS = WAVES * 64;
    constexpr int STAGE_FRAGS = MTG * (KSG > 5 ? KSG : 5);
    constexpr int PF = (STAGE_FRAGS * 64 + THREADS - 1) / THREADS;
    extern __shared__ uint4 lds_w[];                              // [2][STAGE_FRAGS * 64] | k-group scratch per wave | ReLU masks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int depth = a.depth, ks0 = a.ks0;
    const uint32_t e16 = (uint32_t)ks0 * 16u;
    const uint32_t rows_a = e16 + (uint32_t)depth * WIDTH, rows_d = (uint32_t)depth * WIDTH + 8u;
    const uint32_t n_tiles = a.n >> 5;
    const uint32_t n_groups = (n_tiles + WAVES - 1u) / WAVES;
    half_t* const kg = reinterpret_cast<half_t*>(lds_w + 2 * STAGE_FRAGS * 64) + wave * (KG_SCRATCH / 2);
    // ReLU masks of every layer's output (1 bit per value, KSG * 8 values per lane and layer), kept for the dgrad chain
    uint2* const masks = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds_w + 2 * STAGE_FRAGS * 64) + WAVES * KG_SCRATCH) +
                         (size_t)wave * depth * 64 + lane;
    constexpr int MASK_BITS = KSG * 4 >= 16 ? 16 : KSG * 4;       // operand dwords folded into one mask word (two values each)
    const int hid_base = MTG * ks0;
    const int n_stages = 2 * depth + 1 + (a.d_enc != nullptr ? 1 : 0);
    // stage st of the images: source and size in uint4; prefetched through registers (see k_infer_gen): loaded at the top of the
    // stage before, written to the other LDS buffer at its bottom
    uint4v pf[PF];
    auto stage_src = [&](int st, const uint4*& src, int& cnt) {
        if (st <= depth) {
            src = img_fwd + (size_t)(st == 0 ? 0 : hid_base + (st - 1) * MTG * KSG) * 64;
            cnt = (st == 0 ? MTG * ks0 : (st == depth ? KSG : MTG * KSG)) * 64;
        } else {
            const int j = st - depth - 1;
            if (j == 0) { src = img_bwd + (size_t)((depth - 1) * MTG * KSG) * 64; cnt = MTG * 64; }
            else if (j < depth) { src = img_bwd + (size_t)((depth - j - 1) * MTG * KSG) * 64; cnt = MTG * KSG * 64; }
            else { src = img_bwd + (size_t)((depth - 1) * MTG * KSG + MTG) * 64; cnt = KSG * 64; }
        }
    };
    auto stage_fetch = [&](int st) {
        const uint4* src;
        int cnt;
        stage_src(st, src, cnt);
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i = k * THREADS + (int)threadIdx.x;
            if (i < cnt) pf[k] = reinterpret_cast<const uint4v*>(src)[i];
        }
    };
    auto stage_commit = [&](int st, int buf) {
        const uint4* src;
        int cnt;
        stage_src(st, src, cnt);
        uint4* dst = lds_w + buf * (STAGE_FRAGS * 64);
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i = k * THREADS + (int)threadIdx.x;
            if (i < cnt) reinterpret_cast<uint4v*>(dst)[i] = pf[k];
        }
    };

    // layer-0 operands (encoded features, ks0 <= 5 k-steps) of this wave's tile, all loaded at the top of the group (see k_infer_gen)
    half8 f0[5];
    auto load_features = [&](uint32_t g) {
        const uint32_t tl = g * WAVES + (uint32_t)wave;
        const half_t* fq = a.feat + (size_t)((tl < n_tiles ? tl : 0u) * 32u + (uint32_t)r) * e16 + 8 * h;
#pragma unroll
        for (int k = 0; k < 5; k++) {      // k-steps beyond ks0: load the last one again, use zeros -- no branches
            f0[k] = *reinterpret_cast<const half8*>(fq + 16 * (k < ks0 ? k : ks0 - 1));
            if (k >= ks0) f0[k] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };

    int q = 0;
    stage_fetch(0);
    stage_commit(0, 0);
    __syncthreads();
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const uint32_t tile = group * WAVES + (uint32_t)wave;
        const bool active = tile < n_tiles;                         // wave-uniform; idle waves still stage and meet the barriers
        if (active) load_features(group);
        const uint32_t sidx = (active ? tile : 0u) * 32u + (uint32_t)r;
        half_t* const ta = a.acts + (size_t)(active ? tile : 0u) * 4u * rows_a * 8;        // this tile's four sample groups, row 0
        half_t* const td = a.deltas + (size_t)(active ? tile : 0u) * 4u * rows_d * 8;
        half_t* const pd = a.deltas + ((size_t)(sidx >> 3) * rows_d) * 8 + (sidx & 7u);
        f32x16 acc[MTG];                       // forward accumulators, then the dgrad accumulators
        half8 b[KSG];                          // forward operands, then the deltas
        half8 bo;
#pragma unroll 1
        for (int st = 0; st < n_stages; st++) {
            const uint4* lw = lds_w + (q & 1) * (STAGE_FRAGS * 64);
            const bool last = st == n_stages - 1;
            const bool more = !last || group + gridDim.x < n_groups;
            const int next = last ? 0 : st + 1;
            if (more) stage_fetch(next);
            if (active) {
                if (st < depth) {
                    // ---- forward layer st; every layer's input goes to HBM in the k-group layout the weight-gradient GEMM reads
#pragma unroll
                    for (int m = 0; m < MTG; m++) acc[m] = zero16();
                    if (st == 0) {
#pragma unroll
                        for (int s = 0; s < 5; s++) {      // always five k-steps: beyond ks0 the operand is zero
                            if (s < ks0) store_kgroups<false>(kg, f0[s], lane, ta + (size_t)(16 * s) * 8, rows_a);
#pragma unroll
                            for (int m = 0; m < MTG; m++) acc[m] = mfma(ld_frag(lw, m * ks0 + (s < ks0 ? s : ks0 - 1), lane), f0[s], acc[m]);
                            __builtin_amdgcn_sched_barrier(0);      // one k-step's fragments in flight, not all five
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < MTG; m++)
#pragma unroll
                            for (int s = 0; s < KSG; s++) acc[m] = mfma(ld_frag(lw, m * KSG + s, lane), b[s], acc[m]);
                    }
#pragma unroll
                    for (int m = 0; m < MTG; m++) relu_pack(acc[m], b[2 * m], b[2 * m + 1]);
                    half_t* const tl = ta + (size_t)(e16 + (uint32_t)st * WIDTH) * 8;
                    uint32_t mk[2] = {0u, 0u};
#pragma unroll
                    for (int s = 0; s < KSG; s++) {
                        store_kgroups<true>(kg, b[s], lane, tl + (size_t)(16 * s) * 8, rows_a);
                        const uint4v w = __builtin_bit_cast(uint4v, b[s]);
#pragma unroll
                        for (int i = 0; i < 4; i++) {      // ReLU output > 0 <=> its fp16 bits are non-zero
                            // (plain integer tests: hipcc 7.2 folded a packed min(bits, 1) over the four dwords into ONE dword's result)
                            const uint32_t wi = w[i];
                            const uint32_t nz = ((wi & 0x0000ffffu) != 0u ? 1u : 0u) | ((wi & 0xffff0000u) != 0u ? 0x10000u : 0u);
                            const int k = 4 * s + i;
                            mk[k >> 4] = (mk[k >> 4] << 1) | nz;
                        }
                    }
                    masks[(size_t)st * 64] = make_uint2(mk[0], mk[1]);
                } else if (st == depth) {
                    // ---- output layer, loss, dL/dy
                    f32x16 y = zero16();
#pragma unroll
                    for (int s = 0; s < KSG; s++) y = mfma(ld_frag(lw, s, lane), b[s], y);
                    float loss_v = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; j++) bo[j] = (half_t)0.0f;
                    if (h == 0) {
                        const float* t = a.target + (size_t)sidx * 3u;
                        const float yv[3] = {y[0], y[1], y[2]};
                        float dy[3];
                        loss_terms(a.loss_id, yv, t, a.inv_n_total, loss_v, dy);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            bo[c] = (half_t)dy[c];
                            pd[(size_t)((uint32_t)depth * WIDTH + c) * 8] = bo[c];
                        }
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) loss_v += __shfl_xor(loss_v, off);
                    if (lane == 0) a.loss_part[tile] = loss_v;
                } else if (st <= 2 * depth) {
                    // ---- dgrad: W_{l+1}^T delta_{l+1} (the output layer's W^T first), then delta_l = relu'(a_l) * that; the ReLU
                    //      mask is re-read from this lane's own activation stores
                    const int l = 2 * depth - st;      // the layer whose delta this stage produces: depth-1 ... 0
                    if (st == depth + 1) {
#pragma unroll
                        for (int m = 0; m < MTG; m++) acc[m] = mfma(ld_frag(lw, m, lane), bo, zero16());
                    } else {
#pragma unroll
                        for (int m = 0; m < MTG; m++) {
                            acc[m] = zero16();
#pragma unroll
                            for (int s = 0; s < KSG; s++) acc[m] = mfma(ld_frag(lw, m * KSG + s, lane), b[s], acc[m]);
                        }
                    }
                    const uint2 mw = masks[(size_t)l * 64];
                    half_t* const tdl = td + (size_t)((uint32_t)l * WIDTH) * 8;
#pragma unroll
                    for (int s = 0; s < KSG; s++) {
                        uint4v w;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int k = 4 * s + i;
                            const uint32_t bits = ((k >> 4) ? mw.y : mw.x) >> (MASK_BITS - 1 - (k & 15));
                            const short2v keep = -__builtin_bit_cast(short2v, bits & 0x00010001u);      // 0xffff where the ReLU passed
                            const float2v dv = {acc[s >> 1][8 * (s & 1) + 2 * i], acc[s >> 1][8 * (s & 1) + 2 * i + 1]};
                            const half2v dh = __builtin_convertvector(dv, half2v);
                            w[i] = __builtin_bit_cast(uint32_t, dh) & __builtin_bit_cast(uint32_t, keep);
                        }
                        b[s] = __builtin_bit_cast(half8, w);
                        store_kgroups<true>(kg, b[s], lane, tdl + (size_t)(16 * s) * 8, rows_d);
                    }
                } else {
                    // ---- dL/d(encoded input rows 0..31) = W0^T delta_0 for a trainable encoding
                    f32x16 de = zero16();
#pragma unroll
                    for (int s = 0; s < KSG; s++) de = mfma(ld_frag(lw, s, lane), b[s], de);
                    half_t* const po = a.d_enc + (size_t)sidx * 32u + 4 * h;
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) po[(reg & 3) + 8 * (reg >> 2)] = (half_t)de[reg];
                }
            }
            if (more) stage_commit(next, (q + 1) & 1);
            __syncthreads();
            q++;
        }
    }
}

// ---- round 4: the same training step with every layer's ROWS split over the waves of a workgroup (k_train_gen2).
// k_train_gen costs configs[4]'s frame a fifth of its rate (tools/ab_skip.sh: 4 720 -> 5 740 Msamples/s without it) although it
// runs only 51 us alone: a wave carries a whole 32-sample tile through every layer -- MTG accumulators, the layer's operand and two
// staged weight images: 219 + 64 registers and 80 KB of LDS per workgroup --, so a workgroup needs a CU that gen_rays has all but
// left, and holds it for 17 barrier-to-barrier stages of 32 dependent MFMAs.  Here wave w of a sample group computes rows
// 32w .. 32w+31 of the layer for NT tiles: it reads ITS weight fragments straight from L2 into registers one stage ahead (a fragment
// has exactly one reader: no weight staging in LDS at all), the layer's input arrives as MFMA B operands through a 2 x 8 KB (per tile)
// ping-pong in LDS that the waves fill with their own row blocks of the previous layer (one barrier per stage), and the epilogue
// work (ReLU / mask / fp16 pack, the k-group stores of activations and deltas) is split the same way.  152 registers and 37 KB of LDS
// (one tile per sample group; 179 and 53 KB with two) for an 8-layer net.  Every output element is the same sequence of MFMAs as in k_train_gen: activations, deltas and loss are
// bit-identical (tests/test_gpu_mlp.py::test_training_kernels_agree).
template <int WIDTH, int NT>
#ifndef NRC_TRAIN_GEN2_MIN_WAVES
#define NRC_TRAIN_GEN2_MIN_WAVES 2
#endif
__global__ __launch_bounds__(256, NRC_TRAIN_GEN2_MIN_WAVES) void k_train_gen2(TrainArgsGen a, const uint4* __restrict__ img_fwd, const uint4* __restrict__ img_bwd)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    constexpr int MTG = WIDTH / 32, KSG = WIDTH / 16, WAVES = 4, SG = WAVES / MTG;      // SG sample groups of MTG waves
    constexpr int KA = KSG > 5 ? KSG : 5;                                               // weight fragments a wave holds per stage
    constexpr int BBUF = SG * NT * KSG * 64;                                            // uint4 per B-operand buffer
    extern __shared__ uint4 lds2[];      // B operands [2][SG][NT][KSG][64] | k-group scratch per wave | ReLU masks [WAVES][depth][NT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int sg = wave / MTG, w = wave % MTG;
    const int depth = a.depth, ks0 = a.ks0;
    const uint32_t e16 = (uint32_t)ks0 * 16u;
    const uint32_t rows_a = e16 + (uint32_t)depth * WIDTH, rows_d = (uint32_t)depth * WIDTH + 8u;
    const uint32_t n_tiles = a.n >> 5;
    const uint32_t n_groups = (n_tiles + SG * NT - 1u) / (SG * NT);
    uint4* const bbuf = lds2 + (size_t)sg * NT * KSG * 64;
    half_t* const kg = reinterpret_cast<half_t*>(lds2 + 2 * BBUF) + wave * (KG_SCRATCH / 2);
    uint32_t* const masks = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds2 + 2 * BBUF) + WAVES * KG_SCRATCH) +
                            (size_t)wave * depth * NT * 64 + lane;
    const int hid_base = MTG * ks0;
    const int n_stages = 2 * depth + 1 + (a.d_enc != nullptr ? 1 : 0);
    // stage st: this wave's run of fragments in the images (first fragment, count) -- see k_train_gen for the stages
    auto stage_frags = [&](int st, const uint4*& src, int& cnt) {
        if (st == 0) { src = img_fwd + (size_t)(w * ks0) * 64; cnt = ks0; }
        else if (st < depth) { src = img_fwd + (size_t)(hid_base + (st - 1) * MTG * KSG + w * KSG) * 64; cnt = KSG; }
        else if (st == depth) { src = img_fwd + (size_t)(hid_base + (depth - 1) * MTG * KSG) * 64; cnt = KSG; }      // (wave 0 uses it)
        else {
            const int j = st - depth - 1;
            if (j == 0) { src = img_bwd + (size_t)((depth - 1) * MTG * KSG + w) * 64; cnt = 1; }
            else if (j < depth) { src = img_bwd + (size_t)((depth - j - 1) * MTG * KSG + w * KSG) * 64; cnt = KSG; }
            else { src = img_bwd + (size_t)((depth - 1) * MTG * KSG + MTG) * 64; cnt = KSG; }                        // (wave 0 uses it)
        }
    };
    // branch-free: slots beyond the run load its last fragment again (in bounds, never multiplied with a non-zero operand)
    half8 cur[KA], nxt[KA];
    auto load_frags = [&](int st, half8 (&dst)[KA]) {
        const uint4* src;
        int cnt;
        stage_frags(st, src, cnt);
#pragma unroll
        for (int s = 0; s < KA; s++) dst[s] = ld_frag_g(src, s < cnt ? s : cnt - 1, lane);
    };
    load_frags(0, nxt);
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        uint32_t tile[NT], sidx[NT];
        bool act[NT];
        half_t *ta[NT], *td[NT], *pd[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            tile[t] = (group * SG + (uint32_t)sg) * NT + (uint32_t)t;
            act[t] = tile[t] < n_tiles;                              // wave-uniform
            const uint32_t tl = act[t] ? tile[t] : 0u;
            sidx[t] = tl * 32u + (uint32_t)r;
            ta[t] = a.acts + (size_t)tl * 4u * rows_a * 8;           // the tile's four sample groups, row 0
            td[t] = a.deltas + (size_t)tl * 4u * rows_d * 8;
            pd[t] = a.deltas + ((size_t)(sidx[t] >> 3) * rows_d) * 8 + (sidx[t] & 7u);
        }
        f32x16 acc[NT];
        // forward epilogue of layer st: this wave's 32 rows as the next layer's k-steps 2w, 2w+1 (LDS), as activations (HBM), as mask bits
        auto fwd_epilogue = [&](int st, uint4* bn) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (!act[t]) continue;
                half8 lo, hi;
                relu_pack(acc[t], lo, hi);
                bn[(t * KSG + 2 * w) * 64 + lane] = __builtin_bit_cast(uint4, lo);
                bn[(t * KSG + 2 * w + 1) * 64 + lane] = __builtin_bit_cast(uint4, hi);
                half_t* const tl = ta[t] + (size_t)(e16 + (uint32_t)st * WIDTH) * 8;
                store_kgroups<true>(kg, lo, lane, tl + (size_t)(16 * (2 * w)) * 8, rows_a);
                store_kgroups<true>(kg, hi, lane, tl + (size_t)(16 * (2 * w + 1)) * 8, rows_a);
                const uint4v wl = __builtin_bit_cast(uint4v, lo), wh = __builtin_bit_cast(uint4v, hi);
                uint32_t mk = 0u;
#pragma unroll
                for (int i = 0; i < 8; i++) {      // ReLU output > 0 <=> its fp16 bits are non-zero; dword i: bit i low half, bit 16+i high half
                    const uint32_t wi = i < 4 ? wl[i] : wh[i - 4];
                    mk |= (((wi & 0x0000ffffu) != 0u ? 1u : 0u) | ((wi & 0xffff0000u) != 0u ? 0x10000u : 0u)) << i;
                }
                masks[(size_t)(st * NT + t) * 64] = mk;
            }
        };
        // ---- stage 0: layer 0 on the encoded features (B operands straight from memory, every wave its own copy)
        {
#pragma unroll
            for (int s = 0; s < KA; s++) cur[s] = nxt[s];
            load_frags(1, nxt);
            __syncthreads();                       // the previous group's last stage has read its B operands
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acc[t] = zero16();
                if (!act[t]) continue;
                const half_t* fq = a.feat + (size_t)sidx[t] * e16 + 8 * h;
                half8 f0[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {      // k-steps beyond ks0: load the last one again, use zeros -- no branches
                    f0[k] = *reinterpret_cast<const half8*>(fq + 16 * (k < ks0 ? k : ks0 - 1));
                    if (k >= ks0) f0[k] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (k < ks0 && (k % MTG) == w) store_kgroups<false>(kg, f0[k], lane, ta[t] + (size_t)(16 * k) * 8, rows_a);
                    acc[t] = mfma(cur[k], f0[k], acc[t]);
                }
            }
            fwd_epilogue(0, bbuf + BBUF);
        }
#pragma unroll 1
        for (int st = 1; st < n_stages; st++) {
            const bool last = st == n_stages - 1;
            const bool more = !last || group + gridDim.x < n_groups;
#pragma unroll
            for (int s = 0; s < KA; s++) cur[s] = nxt[s];
            if (more) load_frags(last ? 0 : st + 1, nxt);
            __syncthreads();                       // B(st), written by the stage before, is complete
            const uint4* bc = bbuf + (st & 1) * BBUF;
            uint4* bn = bbuf + ((st + 1) & 1) * BBUF;
            if (st < depth) {
                // ---- forward layer st
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = zero16();
#pragma unroll
                for (int s = 0; s < KSG; s++) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (act[t]) acc[t] = mfma(cur[s], ld_frag(bc, t * KSG + s, lane), acc[t]);
                }
                fwd_epilogue(st, bn);
            } else if (st == depth) {
                // ---- output layer, loss, dL/dy: one row block, wave 0 of the sample group
                if (w == 0) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        if (!act[t]) continue;
                        f32x16 y = zero16();
#pragma unroll
                        for (int s = 0; s < KSG; s++) y = mfma(cur[s], ld_frag(bc, t * KSG + s, lane), y);
                        float loss_v = 0.0f;
                        half8 bo;
#pragma unroll
                        for (int j = 0; j < 8; j++) bo[j] = (half_t)0.0f;
                        if (h == 0) {
                            const float* tg = a.target + (size_t)sidx[t] * 3u;
                            const float yv[3] = {y[0], y[1], y[2]};
                            float dy[3];
                            loss_terms(a.loss_id, yv, tg, a.inv_n_total, loss_v, dy);
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                bo[c] = (half_t)dy[c];
                                pd[t][(size_t)((uint32_t)depth * WIDTH + c) * 8] = bo[c];
                            }
                        }
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) loss_v += __shfl_xor(loss_v, off);
                        if (lane == 0) a.loss_part[tile[t]] = loss_v;
                        bn[(t * KSG) * 64 + lane] = __builtin_bit_cast(uint4, bo);      // k-step 0 of the next stage's operand
                    }
                }
            } else if (st <= 2 * depth) {
                // ---- dgrad: W_{l+1}^T delta_{l+1} (the output layer's W^T first), then delta_l = relu'(a_l) * that
                const int l = 2 * depth - st;      // the layer whose delta this stage produces: depth-1 ... 0
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = zero16();
                if (st == depth + 1) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (act[t]) acc[t] = mfma(cur[0], ld_frag(bc, t * KSG, lane), acc[t]);
                } else {
#pragma unroll
                    for (int s = 0; s < KSG; s++) {
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            if (act[t]) acc[t] = mfma(cur[s], ld_frag(bc, t * KSG + s, lane), acc[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (!act[t]) continue;
                    const uint32_t mk = masks[(size_t)(l * NT + t) * 64];
                    half_t* const tdl = td[t] + (size_t)((uint32_t)l * WIDTH) * 8;
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        uint4v wv;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t bits = mk >> (4 * hf + i);
                            const short2v keep = -__builtin_bit_cast(short2v, bits & 0x00010001u);      // 0xffff where the ReLU passed
                            const float2v dv = {acc[t][8 * hf + 2 * i], acc[t][8 * hf + 2 * i + 1]};
                            const half2v dh = __builtin_convertvector(dv, half2v);
                            wv[i] = __builtin_bit_cast(uint32_t, dh) & __builtin_bit_cast(uint32_t, keep);
                        }
                        const half8 bd = __builtin_bit_cast(half8, wv);
                        bn[(t * KSG + 2 * w + hf) * 64 + lane] = __builtin_bit_cast(uint4, wv);
                        store_kgroups<true>(kg, bd, lane, tdl + (size_t)(16 * (2 * w + hf)) * 8, rows_d);
                    }
                }
            } else {
                // ---- dL/d(encoded input rows 0..31) = W0^T delta_0 for a trainable encoding: one row block, wave 0
                if (w == 0) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        if (!act[t]) continue;
                        f32x16 de = zero16();
#pragma unroll
                        for (int s = 0; s < KSG; s++) de = mfma(cur[s], ld_frag(bc, t * KSG + s, lane), de);
                        half_t* const po = a.d_enc + (size_t)sidx[t] * 32u + 4 * h;
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) po[(reg & 3) + 8 * (reg >> 2)] = (half_t)de[reg];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ training: weight gradients
// dW_l = delta_l (rows: out neurons, k: samples) * a_{l-1}^T.  One workgroup per K-chunk of samples, one 32x32
// output tile per wave iteration, partial result to this chunk's slab (fixed-order reduction afterwards).
struct WgradTile {
    uint32_t a_row0, b_row0, m_valid, n_valid, param_off, in_dim;
};
#ifndef NRC_WGRAD_WAVES
#define NRC_WGRAD_WAVES 7
#endif
constexpr int WGRAD_WAVES = NRC_WGRAD_WAVES;
constexpr uint32_t WGRAD_CHUNK = 128;

__global__ __launch_bounds__(WGRAD_WAVES * 64) void k_wgrad(const half_t* __restrict__ deltas,
                                                           const half_t* __restrict__ acts, uint32_t n,
                                                           uint32_t rows_d, uint32_t rows_a,
                                                           const WgradTile* __restrict__ tiles, int n_tiles,
                                                           float* __restrict__ slabs, uint32_t n_params)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t k0 = blockIdx.x * WGRAD_CHUNK;
    const uint32_t left = n - k0;
    const int ksteps = (int)((left < WGRAD_CHUNK ? left : WGRAD_CHUNK) >> 4);
    float* slab = slabs + (size_t)blockIdx.x * n_params;
    half8 zero;
#pragma unroll
    for (int j = 0; j < 8; j++) zero[j] = (half_t)0.0f;
    for (int t = wave; t < n_tiles; t += WGRAD_WAVES) {
        const WgradTile T = tiles[t];
        const bool av = (uint32_t)r < T.m_valid, bv = (uint32_t)r < T.n_valid;
        // operand k-groups: [sample/8][rows][8]; lane (r,h) reads row r of group 2s+h -> 512 contiguous bytes per half wave
        const half_t* ap = deltas + ((size_t)((k0 >> 3) + h) * rows_d + T.a_row0 + (av ? r : 0)) * 8;
        const half_t* bp = acts + ((size_t)((k0 >> 3) + h) * rows_a + T.b_row0 + (bv ? r : 0)) * 8;
        f32x16 acc = zero16();
#pragma unroll 4
        for (int s = 0; s < ksteps; s++) {
            half8 av8 = *reinterpret_cast<const half8*>(ap + (size_t)s * (16 * rows_d));
            half8 bv8 = *reinterpret_cast<const half8*>(bp + (size_t)s * (16 * rows_a));
            acc = mfma(av ? av8 : zero, bv ? bv8 : zero, acc);
        }
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const uint32_t row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < T.m_valid && bv) slab[T.param_off + row * T.in_dim + r] = acc[reg];
        }
    }
}

// ---- round 4: one workgroup per (K-chunk, four row-block tasks) instead of one per K-chunk with every tile of the network.
// k_wgrad above is latency-bound: its 896 waves walk ~19 output tiles each, every tile two dependent rounds of global loads in
// front of 8 MFMAs, every operand block fetched once per tile that uses it (61 us stand-alone for an 8x128 net, 183 us beside
// gen_rays -- the longest kernel of configs[4]'s training stream).  Here a wave owns ONE 32-row block of a layer's delta (the A
// operand) and up to four 32-column blocks of the layer's input (B): every k-step is one A load + NTL B loads for NTL MFMAs into
// NTL live accumulators, the loads of the next two k-steps in flight behind the current two.  The K range of a workgroup (`chunk`
// samples) is chosen by the host so that the launch is ~2 workgroups per CU; the partial sums go to the chunk's slab and are added
// in the fixed order of k_reduce_grads as before (bitwise reproducible; for a 6x64 net the chunk is the old 128 samples and the
// gradient is bit-identical to k_wgrad's).
struct WgradTask {
    uint32_t a_row0, b_row0, m_valid, n_tiles, n_valid_last, param_off, in_dim, pad;
};
#ifndef NRC_WGRAD2_NTL
#define NRC_WGRAD2_NTL 4            // 32-column blocks (accumulators) per task, at most 4
#endif
#ifndef NRC_WGRAD2_U
#define NRC_WGRAD2_U 2              // k-steps per pipeline stage
#endif
constexpr int WGRAD2_WAVES = 4;
constexpr int WGRAD2_NTL = NRC_WGRAD2_NTL;
constexpr int WGRAD2_U = NRC_WGRAD2_U;
template <int NTL>
__device__ __forceinline__ void wgrad_task(const WgradTask& T, const half_t* __restrict__ deltas, const half_t* __restrict__ acts, uint32_t k0,
                                           int ksteps, uint32_t rows_d, uint32_t rows_a, float* __restrict__ slab, int lane)
{
    constexpr int U = WGRAD2_U;
    const int r = lane & 31, h = lane >> 5;
    const bool av = (uint32_t)r < T.m_valid;
    // operand k-groups: [sample/8][rows][8]; lane (r,h) reads row r of group 2s+h -> 512 contiguous bytes per half wave
    const half_t* ap = deltas + ((size_t)((k0 >> 3) + h) * rows_d + T.a_row0 + (av ? r : 0)) * 8;
    const half_t* bp[NTL];
    bool bv[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) {
        bv[t] = (uint32_t)r < (t == NTL - 1 ? T.n_valid_last : 32u);
        bp[t] = acts + ((size_t)((k0 >> 3) + h) * rows_a + T.b_row0 + 32u * (uint32_t)t + (bv[t] ? r : 0)) * 8;
    }
    const size_t a_step = (size_t)16 * rows_d, b_step = (size_t)16 * rows_a;
    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; t++) acc[t] = zero16();
    half8 a0[U], b0[U][NTL], a1[U], b1[U][NTL];
    // rows / columns beyond the tile's valid ones read row 0 of the block and are never stored; k-steps beyond the chunk's end load the
    // last valid step again (in bounds) and are not multiplied
#define NRC_WG_LOAD(A, B, S0)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                                                               \
        const int sc = (S0) + u < ksteps ? (S0) + u : ksteps - 1;                                                  \
        A[u] = *reinterpret_cast<const half8*>(ap + (size_t)sc * a_step);                                          \
        _Pragma("unroll") for (int t = 0; t < NTL; t++) B[u][t] = *reinterpret_cast<const half8*>(bp[t] + (size_t)sc * b_step); \
    }
#define NRC_WG_MMA(A, B, S0)                                                                                      \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                                                               \
        if ((S0) + u < ksteps) {                                                                                   \
            _Pragma("unroll") for (int t = 0; t < NTL; t++) acc[t] = mfma(A[u], B[u][t], acc[t]);                  \
        }                                                                                                          \
    }
    NRC_WG_LOAD(a0, b0, 0)
    for (int s = 0; s < ksteps; s += 2 * U) {
        NRC_WG_LOAD(a1, b1, s + U)
        NRC_WG_MMA(a0, b0, s)
        NRC_WG_LOAD(a0, b0, s + 2 * U)
        NRC_WG_MMA(a1, b1, s + U)
    }
#undef NRC_WG_LOAD
#undef NRC_WG_MMA
#pragma unroll
    for (int t = 0; t < NTL; t++) {
        float* o = slab + T.param_off + 32u * (uint32_t)t + (uint32_t)r;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const uint32_t row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < T.m_valid && bv[t]) o[(size_t)row * T.in_dim] = acc[t][reg];
        }
    }
}
#ifndef NRC_WGRAD2_MIN_WAVES
#define NRC_WGRAD2_MIN_WAVES 2
#endif
__global__ __launch_bounds__(WGRAD2_WAVES * 64, NRC_WGRAD2_MIN_WAVES) void k_wgrad2(const half_t* __restrict__ deltas, const half_t* __restrict__ acts, uint32_t n,
                                                             uint32_t chunk, uint32_t rows_d, uint32_t rows_a,
                                                             const WgradTask* __restrict__ tasks, int n_tasks, float* __restrict__ slabs,
                                                             uint32_t n_params)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = __builtin_amdgcn_readfirstlane((int)blockIdx.y * WGRAD2_WAVES + wave);
    if (task >= n_tasks) return;
    const WgradTask T = tasks[task];
    const uint32_t k0 = blockIdx.x * chunk;
    const uint32_t left = n - k0;
    const int ksteps = (int)((left < chunk ? left : chunk) >> 4);
    float* slab = slabs + (size_t)blockIdx.x * n_params;
    if constexpr (WGRAD2_NTL == 1) {
        wgrad_task<1>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane);
    } else if constexpr (WGRAD2_NTL == 2) {
        if (T.n_tiles == 1) wgrad_task<1>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane);
        else wgrad_task<2>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane);
    } else {
        switch (T.n_tiles) {
        case 1: wgrad_task<1>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane); break;
        case 2: wgrad_task<2>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane); break;
        case 3: wgrad_task<3>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane); break;
        default: wgrad_task<4>(T, deltas, acts, k0, ksteps, rows_d, rows_a, slab, lane); break;
        }
    }
}

// grad[i] = sum over chunk slabs in a fixed order (bitwise reproducible): a 256-thread block owns 64 parameters, four
// thread groups each add a contiguous quarter of the slabs in chunk order, the four partials are combined in group order.
// Block 0 also folds the loss partials.
__global__ __launch_bounds__(256) void k_reduce_grads(const float* __restrict__ slabs, uint32_t n_chunks, uint32_t n_params,
                                                     float* __restrict__ grad, const float* __restrict__ loss_part,
                                                     uint32_t n_loss, float* __restrict__ loss)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    __shared__ float part[4][64];
    __shared__ float red[256];
    const uint32_t p = threadIdx.x & 63u, g = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64u + p;
    const uint32_t per = (n_chunks + 3u) / 4u;
    const uint32_t c0 = g * per, c1 = min(c0 + per, n_chunks);
    float s = 0.0f;
    if (i < n_params)
        for (uint32_t c = c0; c < c1; c++) s += slabs[(size_t)c * n_params + i];
    part[g][p] = s;
    __syncthreads();
    if (g == 0 && i < n_params) grad[i] = ((part[0][p] + part[1][p]) + part[2][p]) + part[3][p];
    if (blockIdx.x == 0) {
        float l = 0.0f;
        for (uint32_t k = threadIdx.x; k < n_loss; k += 256u) l += loss_part[k];
        red[threadIdx.x] = l;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) { loss[0] = red[0]; loss[1] = 0.0f; }
    }
}

// EMA{Adam} (tiny-cuda-nn defaults: beta1 .9, beta2 .999, eps 1e-8, l2_reg 1e-8), SURVEY App. B
struct AdamArgs {
    float lr_t, inv_loss_scale, ema_old, ema_new, ema_div;
};
// (no contraction in the two update rules: k_adam_ema / k_sgd_ema and the one-launch k_opt_pack then round identically, and like the
// CPU oracle, which is built with -ffp-contract=off)
#pragma clang fp contract(off)
__device__ __forceinline__ void adam_ema_update(uint32_t i, float* __restrict__ w, float* __restrict__ ema, float* __restrict__ m,
                                                float* __restrict__ v, float graw, bool matrix, const AdamArgs& a, float* w_new,
                                                float* ema_new)
{
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, l2 = 1e-8f;
    float wi = w[i];
    if (!matrix && graw == 0.0f) {      // tiny-cuda-nn: grid entries with a zero gradient keep weight and moments
        const float e = (ema[i] * a.ema_old + wi * a.ema_new) / a.ema_div;
        ema[i] = e;
        *w_new = wi; *ema_new = e;
        return;
    }
    float g = graw * a.inv_loss_scale + (matrix ? l2 * wi : 0.0f);
    float mi = b1 * m[i] + (1.0f - b1) * g;
    float vi = b2 * v[i] + (1.0f - b2) * (g * g);
    m[i] = mi;
    v[i] = vi;
    wi = wi - a.lr_t * mi / (sqrtf(vi) + eps);
    w[i] = wi;
    const float e = (ema[i] * a.ema_old + wi * a.ema_new) / a.ema_div;
    ema[i] = e;
    *w_new = wi; *ema_new = e;
}
// tiny-cuda-nn sgd.h nested in the EMA wrapper: w -= lr * (g / loss_scale + l2 * w), l2_reg 1e-8, every parameter
__device__ __forceinline__ void sgd_ema_update(uint32_t i, float* __restrict__ w, float* __restrict__ ema, float graw, float lr,
                                               const AdamArgs& a, float* w_new, float* ema_new)
{
    const float l2 = 1e-8f;
    float wi = w[i];
    const float g = graw * a.inv_loss_scale + l2 * wi;
    wi = wi - lr * g;
    w[i] = wi;
    const float e = (ema[i] * a.ema_old + wi * a.ema_new) / a.ema_div;
    ema[i] = e;
    *w_new = wi; *ema_new = e;
}
#pragma clang fp contract(fast)

__global__ void k_adam_ema(float* __restrict__ w, float* __restrict__ ema, float* __restrict__ m,
                           float* __restrict__ v, const float* __restrict__ grad, uint32_t n, uint32_t n_matrix, AdamArgs a)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wn, en;
    adam_ema_update(i, w, ema, m, v, grad[i], i < n_matrix, a, &wn, &en);
}

__global__ void k_sgd_ema(float* __restrict__ w, float* __restrict__ ema, const float* __restrict__ grad, uint32_t n, float lr,
                          AdamArgs a)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float wn, en;
    sgd_ema_update(i, w, ema, grad[i], lr, a, &wn, &en);
}

// The optimizer step of a model without a trainable encoding as ONE launch: every thread updates its parameter and stores the
// fp16 copies straight into the three fragment images (dst: image slot of parameter i in the forward / EMA inference / backward
// image, -1 = not in that image; the images' padding slots are zero since construction), thread 0 also publishes the step's loss
// (the 8-byte {loss, sequence number} store of k_publish_loss).  Replaces k_adam_ema + k_pack + k_publish_loss: two dependent
// launches (~17 us each on this stack) less on the serial chain backward -> exchange -> optimizer -> next backward.
struct PackDst {
    const int32_t *fwd, *inf, *bwd;
    half_t *pk_fwd, *pk_inf, *pk_bwd;
};
template <bool SGD>
__global__ __launch_bounds__(256) void k_opt_pack(float* __restrict__ w, float* __restrict__ ema, float* __restrict__ m,
                                                 float* __restrict__ v, const float* __restrict__ grad, uint32_t n, float lr,
                                                 AdamArgs a, PackDst d, const float* __restrict__ loss, uint32_t loss_seq,
                                                 unsigned long long* __restrict__ loss_cell)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && loss_cell != nullptr) {
        const unsigned long long bits = (unsigned long long)__builtin_bit_cast(uint32_t, loss[0]) | ((unsigned long long)loss_seq << 32);
        __hip_atomic_store(loss_cell, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i >= n) return;
    float wn, en;
    if (SGD) sgd_ema_update(i, w, ema, grad[i], lr, a, &wn, &en);
    else adam_ema_update(i, w, ema, m, v, grad[i], true, a, &wn, &en);
    const int32_t df = d.fwd[i], di = d.inf[i], db = d.bwd[i];
    if (df >= 0) d.pk_fwd[df] = (half_t)wn;
    if (di >= 0) d.pk_inf[di] = (half_t)en;
    if (db >= 0) d.pk_bwd[db] = (half_t)wn;
}

// The trainable table's share of the step, one thread per entry (two features): the gradient straight from the packed fp16 table
// the atomics accumulated into (FROM16; no fp32 widening pass) or from the fp32 vector (after an exchange / a caller's hook), the
// update of k_adam_ema / k_sgd_ema for parameters n_matrix + 2e, + 1, and the two fp16 gather copies (training weights, EMA set
// `next`) that k_pack_grid would write.  Replaces k_grid_grad_f32 + the table part of k_adam_ema + k_pack_grid.
template <bool SGD, bool FROM16>
__global__ __launch_bounds__(256) void k_grid_opt(float* __restrict__ w, float* __restrict__ ema, float* __restrict__ m,
                                                 float* __restrict__ v, const float* __restrict__ grad,
                                                 uint32_t* __restrict__ grad16, uint32_t n_matrix, uint32_t n_entries, float lr,
                                                 AdamArgs a, uint32_t* __restrict__ t_train, uint32_t* __restrict__ t_ema)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n_entries) return;
    const uint32_t i0 = n_matrix + 2u * e;
    float g0, g1;
    if (FROM16) {
        const uint32_t word = grad16[e];
        const half2v h = __builtin_bit_cast(half2v, word);
        g0 = (float)h[0]; g1 = (float)h[1];
        // the table the atomics of the next step accumulate into is cleared here, entry by touched entry, instead of by a 28 MB memset in
        // front of every backward pass (the runtime's fill kernel runs at wave priority 0 beside this library's kernels at 3: ~170 us in
        // the frame for 7 us of work)
        if (word != 0u) grad16[e] = 0u;
    } else {
        g0 = grad[i0]; g1 = grad[i0 + 1u];
    }
    float2v wn, en;
    float x, y;
    if (SGD) sgd_ema_update(i0, w, ema, g0, lr, a, &x, &y);
    else adam_ema_update(i0, w, ema, m, v, g0, false, a, &x, &y);
    wn[0] = x; en[0] = y;
    if (SGD) sgd_ema_update(i0 + 1u, w, ema, g1, lr, a, &x, &y);
    else adam_ema_update(i0 + 1u, w, ema, m, v, g1, false, a, &x, &y);
    wn[1] = x; en[1] = y;
    t_train[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(wn, half2v));
    t_ema[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(en, half2v));
}

// Round 4: the same step with TWO entries (four parameters) per thread and 16-byte accesses.  k_grid_opt's one entry per thread reads and
// writes its two parameters as separate 4-byte accesses at an 8-byte stride and moves 36 B per entry at 2.3 TB/s (110 us for the 7 M
// entries of the reference-default table -- the longest kernel of the training step).  Here a thread whose four gradients are all zero --
// three entries in four at 16 384 train rays -- loads w and ema (float4), stores ema and the EMA gather copy, and leaves the weight
// copies alone: w does not change, so the training gather copy already holds fp16(w) (every writer of w also writes the copy: this
// kernel, k_pack_grid after set_params / construction).  The others run the per-parameter update of adam_ema_update / sgd_ema_update
// (same expressions, same rounding: the functions below restate them on values; fp contraction is off here too).
#pragma clang fp contract(off)
__device__ __forceinline__ float ema_value(float e_old, float wi, const AdamArgs& a) { return (e_old * a.ema_old + wi * a.ema_new) / a.ema_div; }
__device__ __forceinline__ void adam_value(float& wi, float& mi, float& vi, float graw, const AdamArgs& a)      // non-matrix parameter, graw != 0
{
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float g = graw * a.inv_loss_scale + 0.0f;
    mi = b1 * mi + (1.0f - b1) * g;
    vi = b2 * vi + (1.0f - b2) * (g * g);
    wi = wi - a.lr_t * mi / (sqrtf(vi) + eps);
}
__device__ __forceinline__ void sgd_value(float& wi, float graw, float lr, const AdamArgs& a)
{
    const float l2 = 1e-8f;
    const float g = graw * a.inv_loss_scale + l2 * wi;
    wi = wi - lr * g;
}
template <bool SGD, bool FROM16>
__global__ __launch_bounds__(256) void k_grid_opt2(float* __restrict__ w, float* __restrict__ ema, float* __restrict__ m,
                                                  float* __restrict__ v, const float* __restrict__ grad,
                                                  uint32_t* __restrict__ grad16, uint32_t n_matrix, uint32_t n_pairs, float lr,
                                                  AdamArgs a, uint32_t* __restrict__ t_train, uint32_t* __restrict__ t_ema)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t e = 2u * p, i0 = n_matrix + 4u * p;      // (n_matrix % 4 == 0 and an even number of entries: checked by the caller)
    float g[4];
    if (FROM16) {
        const uint2 word = *reinterpret_cast<const uint2*>(grad16 + e);
        const half2v h0 = __builtin_bit_cast(half2v, word.x), h1 = __builtin_bit_cast(half2v, word.y);
        g[0] = (float)h0[0]; g[1] = (float)h0[1]; g[2] = (float)h1[0]; g[3] = (float)h1[1];
        if ((word.x | word.y) != 0u) *reinterpret_cast<uint2*>(grad16 + e) = make_uint2(0u, 0u);      // (see k_grid_opt: the table is left clean)
    } else {
        const float4 gv = *reinterpret_cast<const float4*>(grad + i0);
        g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
    }
    const float4 wv = *reinterpret_cast<const float4*>(w + i0), ev = *reinterpret_cast<const float4*>(ema + i0);
    float wi[4] = {wv.x, wv.y, wv.z, wv.w}, ei[4] = {ev.x, ev.y, ev.z, ev.w};
    const bool untouched = !SGD && g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f && g[3] == 0.0f;
    if (!untouched) {
        if (SGD) {
#pragma unroll
            for (int k = 0; k < 4; k++) sgd_value(wi[k], g[k], lr, a);
        } else {
            const float4 mv = *reinterpret_cast<const float4*>(m + i0), vv = *reinterpret_cast<const float4*>(v + i0);
            float mi[4] = {mv.x, mv.y, mv.z, mv.w}, vi[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (g[k] != 0.0f) adam_value(wi[k], mi[k], vi[k], g[k], a);      // a zero gradient keeps weight and moments, parameter by parameter
            *reinterpret_cast<float4*>(m + i0) = make_float4(mi[0], mi[1], mi[2], mi[3]);
            *reinterpret_cast<float4*>(v + i0) = make_float4(vi[0], vi[1], vi[2], vi[3]);
        }
        *reinterpret_cast<float4*>(w + i0) = make_float4(wi[0], wi[1], wi[2], wi[3]);
        const float2v w0 = {wi[0], wi[1]}, w1 = {wi[2], wi[3]};
        *reinterpret_cast<uint2*>(t_train + e) = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(w0, half2v)),
                                                           __builtin_bit_cast(uint32_t, __builtin_convertvector(w1, half2v)));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) ei[k] = ema_value(ei[k], wi[k], a);
    *reinterpret_cast<float4*>(ema + i0) = make_float4(ei[0], ei[1], ei[2], ei[3]);
    const float2v e0 = {ei[0], ei[1]}, e1 = {ei[2], ei[3]};
    *reinterpret_cast<uint2*>(t_ema + e) = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(e0, half2v)),
                                                     __builtin_bit_cast(uint32_t, __builtin_convertvector(e1, half2v)));
}
#pragma clang fp contract(fast)

// fragment images from the canonical fp32 vectors
__global__ void k_pack(const float* __restrict__ w, const float* __restrict__ ema, const int32_t* __restrict__ src_fwd,
                       const int32_t* __restrict__ src_inf, uint32_t n_fwd, const int32_t* __restrict__ src_bwd, uint32_t n_bwd,
                       half_t* __restrict__ pk_infer, half_t* __restrict__ pk_fwd, half_t* __restrict__ pk_bwd)
{
    NRC_RAISE_WAVE_PRIORITY(1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_fwd) {
        const int32_t s = src_fwd[i], si = src_inf[i];      // the EMA (inference) image may order layer 0's inputs differently
        pk_infer[i] = si < 0 ? (half_t)0.0f : (half_t)ema[si];
        pk_fwd[i] = s < 0 ? (half_t)0.0f : (half_t)w[s];
    }
    if (i < n_bwd) {
        const int32_t s = src_bwd[i];
        pk_bwd[i] = s < 0 ? (half_t)0.0f : (half_t)w[s];
    }
}

}  // namespace

// ================================================================================================ host
static uint32_t pos_enc_dims(uint32_t id) { return id == 0 ? 32u : id == 1 ? 3u : id == 2 ? 36u : id == 3 ? 72u : 0u; }
static uint32_t dir_enc_dims(uint32_t id) { return id == 0 ? 8u : id == 1 ? 2u : id == 2 ? 8u : ~0u; }

Mlp::Mlp(const nrc_config& cfg) : cfg_(cfg)
{
    width_ = cfg.nn_width;
    depth_ = cfg.nn_depth;
    if (pos_enc_dims(cfg.pos_id) == 0 || dir_enc_dims(cfg.dir_id) == ~0u) fail("NNEncodingConfig posID/dirID is invalid");
    if (width_ != 16 && width_ != 32 && width_ != 64 && width_ != 128)
        fail("nnWidth must be 16, 32, 64 or 128 -- tiny-cuda-nn's FullyFusedMLP widths (got " + std::to_string(width_) + ")");
    // a 16-wide network runs on the 32-row MFMA tiles of the 32-wide kernels: neurons 16..31 of every hidden layer have all-zero
    // fragment entries, so they stay exactly 0 through ReLU and contribute exactly 0 downstream; the parameter vector, gradients
    // and optimizer see the true 16-wide shapes (half of the MFMA work is padding: the model is 10x cheaper than 6x64 anyway)
    kw_ = width_ < 32 ? 32 : width_;
    xcd8_ = device_xcds() == 8;
    if (depth_ < 1 || depth_ > 16) fail("nnDepth must be in 1..16 (got " + std::to_string(depth_) + ")");
    if (std::strcmp(cfg.optimizer, "Adam") == 0) sgd_ = false;
    else if (std::strcmp(cfg.optimizer, "SGD") == 0) sgd_ = true;
    else fail(std::string("unsupported optimizer ") + cfg.optimizer + " (Adam and SGD are built; both run inside the EMA wrapper)");
    if (std::strcmp(cfg.loss_fn, "RelativeL2Luminance") == 0) loss_id_ = 0;
    else if (std::strcmp(cfg.loss_fn, "L2") == 0) loss_id_ = 1;
    else if (std::strcmp(cfg.loss_fn, "RelativeL2") == 0) loss_id_ = 2;
    else if (std::strcmp(cfg.loss_fn, "L1") == 0) loss_id_ = 3;
    else if (std::strcmp(cfg.loss_fn, "Mape") == 0) loss_id_ = 4;
    else if (std::strcmp(cfg.loss_fn, "Smape") == 0) loss_id_ = 5;
    else if (std::strcmp(cfg.loss_fn, "LogL1") == 0) loss_id_ = 6;
    else
        fail(std::string("unsupported loss ") + cfg.loss_fn +
             " (built: RelativeL2Luminance, L2, RelativeL2, L1, Mape, Smape, LogL1; tiny-cuda-nn's CrossEntropy and Variance need a "
             "sample pdf the NRC path does not have)");
    // encoded width, padded to a multiple of 16 with 1.0 (tiny-cuda-nn); the fused kernels cover the north-star model
    enc_dims_ = (pos_enc_dims(cfg.pos_id) + dir_enc_dims(cfg.dir_id) + 15u) / 16u * 16u;
    fused_ = cfg.pos_id == 3 && cfg.dir_id == 0 && width_ == 64 && depth_ == 6;

    uint32_t off = 0;
    for (uint32_t l = 0; l <= depth_; l++) {
        MlpLayer L;
        L.in = l == 0 ? enc_dims_ : width_;
        L.out = l == depth_ ? 3u : width_;
        L.off = off;
        off += L.in * L.out;
        layers_.push_back(L);
    }
    n_mlp_ = off;
    hash_ = cfg.pos_id == 0;
    uint32_t n_grid = 0;
    if (hash_) {      // per-level tables: dense while res^3 (rounded up to 8) fits, else 2^log2_hashmap_size entries
        const uint32_t log2_size = cfg.hashgrid_log2_size ? cfg.hashgrid_log2_size : 19u;
        if (log2_size < 4 || log2_size > 24) fail("hashgrid_log2_size must be in 4..24");
        uint32_t o = 0;
        for (uint32_t l = 0; l < HG_LEVELS; l++) {
            const double dense = std::pow((double)(16u << l), 3.0);
            uint32_t cnt = dense > 2147483647.0 ? 2147483647u : (uint32_t)dense;
            cnt = (cnt + 7u) / 8u * 8u;
            if (cnt > (1u << log2_size)) cnt = 1u << log2_size;
            hg_off_[l] = o;
            o += cnt;
        }
        hg_off_[HG_LEVELS] = o;
        n_grid_entries_ = o;
        n_grid = o * 2u;
    }
    n_params_ = n_mlp_ + n_grid;

    // tiny-cuda-nn v1.6's initialisation (src/NeuralRadianceCache.cu:39 -> tcnn::create_from_config -> Trainer::initialize_params;
    // recalled from upstream, the submodule is absent -- DESIGN.md section 2): pcg32{seed} on stream 1; the network's matrices first
    // -- each Xavier-uniform under the bound sqrt(6 / (rows + columns)) of its STORED shape, element = next_float() * 2 * scale - scale
    // in that order of operations, the output matrix stored with 16 rows (rows 3..15 feed the padded outputs nobody reads: drawn,
    // kept for the tcnn-layout dump, never part of this model) --, then the table: generate_random_uniform's GPU order (thread i of
    // ceil(n / 4) rounded up to 128 writes draw 4 i + j to element i + n_threads * j), value = fma(u, 2e-4, -1e-4).
    std::vector<float> w(n_params_);
    Pcg32 rng;
    rng.seed(cfg.seed, 1);
    for (int k = 0; k < 4; k++) tcnn_dead_rows_[k].assign((size_t)13 * width_, 0.0f);
    for (uint32_t l = 0; l <= depth_; l++) {
        const MlpLayer& L = layers_[l];
        const uint32_t rows = l == depth_ ? 16u : L.out;
        const float scale = 1.0f * sqrtf(6.0f / (float)(L.in + rows));
        for (uint32_t i = 0; i < L.in * rows; i++) {
            const float x = rng.nextf() * 2.0f * scale - scale;
            if (i < L.in * L.out) w[L.off + i] = x;
            else tcnn_dead_rows_[0][i - L.in * L.out] = x;
        }
    }
    tcnn_dead_rows_[1] = tcnn_dead_rows_[0];      // the EMA copy starts as the weights
    if (n_grid > 0) {
        const size_t n_threads = (((size_t)n_grid + 3) / 4 + 127) / 128 * 128;
        const float lower = -1e-4f, upper = 1e-4f;
        for (size_t k = 0; k < 4 * n_threads; k++) {
            const float u = rng.nextf();
            const size_t idx = k / 4 + n_threads * (k % 4);
            if (idx < n_grid) w[n_mlp_ + idx] = fmaf(u, upper - lower, lower);
        }
    }

    const size_t pb = (size_t)n_params_ * sizeof(float);
    dev_alloc(&d_w_, pb, "d_w_");
    dev_alloc(&d_ema_, pb, "d_ema_");
    dev_alloc(&d_m_, pb, "d_m_");
    dev_alloc(&d_v_, pb, "d_v_");
    // gradient vector and the {loss, pad} cell share one allocation so that the multi-GPU driver all-reduces both at once
    dev_alloc(&d_grad_, pb + 4 * sizeof(float), "d_grad_");
    d_loss_ = d_grad_ + n_params_;
    NRC_HIP(hipMemcpy(d_w_, w.data(), pb, hipMemcpyHostToDevice));
    NRC_HIP(hipMemcpy(d_ema_, w.data(), pb, hipMemcpyHostToDevice));
    NRC_HIP(hipMemset(d_m_, 0, pb));
    NRC_HIP(hipMemset(d_v_, 0, pb));
    NRC_HIP(hipMemset(d_grad_, 0, pb + 4 * sizeof(float)));

    // fragment-image gather tables: [frag][lane][8]; layer 0 [mt][s], hidden layer l [mt][s], output [s];
    // backward image: hidden layer l (W^T) [mt over inputs][s over outputs], output layer [mt] (one k-step, 3 live rows)
    const int D = (int)depth_, mt_n = (int)kw_ / 32, ksh = (int)kw_ / 16, ks0 = (int)enc_dims_ / 16, W = (int)width_;
    const int E = (int)enc_dims_;
    const int hid_base = mt_n * ks0, out_base = hid_base + (D - 1) * mt_n * ksh;
    n_frag_fwd_ = (uint32_t)(out_base + ksh);
    const int din_base = (D - 1) * mt_n * ksh + mt_n;      // W0^T fragments (rows = encoded dims 0..31) for dL/d(input)
    n_frag_bwd_ = (uint32_t)(din_base + (hash_ ? ksh : 0));
    std::vector<int32_t> sf((size_t)n_frag_fwd_ * 512, -1), sb((size_t)n_frag_bwd_ * 512, -1);
    auto slot = [](int frag, int lane, int j) { return ((size_t)frag * 64 + lane) * 8 + j; };
    for (int lane = 0; lane < 64; lane++) {
        const int r = lane & 31, h = lane >> 5;
        for (int j = 0; j < 8; j++) {
            for (int mt = 0; mt < mt_n; mt++) {
                const bool row_ok = 32 * mt + r < W;          // rows / columns beyond the true width stay -1 = zero (width 16)
                for (int s = 0; s < ks0; s++) {
                    const int k = fused_ ? fmap80(s, h, j) : 16 * s + 8 * h + j;
                    if (row_ok) sf[slot(mt * ks0 + s, lane, j)] = (int32_t)(layers_[0].off + (32 * mt + r) * E + k);
                }
                for (int l = 1; l < D; l++)
                    for (int s = 0; s < ksh; s++) {
                        if (!row_ok || kperm(s, h, j) >= W) continue;
                        sf[slot(hid_base + (l - 1) * mt_n * ksh + mt * ksh + s, lane, j)] =
                            (int32_t)(layers_[l].off + (32 * mt + r) * W + kperm(s, h, j));
                        sb[slot((l - 1) * mt_n * ksh + mt * ksh + s, lane, j)] =
                            (int32_t)(layers_[l].off + kperm(s, h, j) * W + (32 * mt + r));
                    }
                const int k = 8 * h + j;
                if (k < 3 && row_ok) sb[slot((D - 1) * mt_n * ksh + mt, lane, j)] = (int32_t)(layers_[D].off + k * W + (32 * mt + r));
            }
            for (int s = 0; s < ksh; s++) {
                if (kperm(s, h, j) >= W) continue;
                if (r < 3) sf[slot(out_base + s, lane, j)] = (int32_t)(layers_[D].off + r * W + kperm(s, h, j));
                if (hash_) sb[slot(din_base + s, lane, j)] = (int32_t)(layers_[0].off + kperm(s, h, j) * E + r);
            }
        }
    }
    dev_alloc(&d_src_fwd_, sf.size() * 4, "d_src_fwd_");
    dev_alloc(&d_src_bwd_, sb.size() * 4, "d_src_bwd_");
    NRC_HIP(hipMemcpy(d_src_fwd_, sf.data(), sf.size() * 4, hipMemcpyHostToDevice));
    // Generic models with the Frequency(12) + OneBlob(4) input (configs[4]'s 8x128): EMA inference encodes inside k_infer_gen as
    // k_infer does, so its image takes layer 0's inputs in that encoder's order (fmap80); training keeps k_encode's natural order
    enc80_generic_ = !fused_ && !hash_ && cfg.pos_id == 3 && cfg.dir_id == 0;
    d_src_inf_ = d_src_fwd_;
    if (enc80_generic_) {
        std::vector<int32_t> si = sf;
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++)
                for (int mt = 0; mt < mt_n; mt++)
                    for (int s = 0; s < ks0; s++)
                        if (32 * mt + (lane & 31) < W)
                            si[slot(mt * ks0 + s, lane, j)] = (int32_t)(layers_[0].off + (32 * mt + (lane & 31)) * E + fmap80(s, lane >> 5, j));
        dev_alloc(&d_src_inf_, si.size() * 4, "d_src_inf_");
        NRC_HIP(hipMemcpy(d_src_inf_, si.data(), si.size() * 4, hipMemcpyHostToDevice));
    }
    NRC_HIP(hipMemcpy(d_src_bwd_, sb.data(), sb.size() * 4, hipMemcpyHostToDevice));
    for (auto& p : d_pk_infer_) {
        dev_alloc(&p, sf.size() * 2, "p");
        NRC_HIP(hipMemset(p, 0, sf.size() * 2));      // k_opt_pack never writes the padding slots
    }
    // inverse maps for the one-launch optimizer step (k_opt_pack): parameter -> its slot in each image
    fused_opt_ = !debug_switch("no_fused_opt");
    if (fused_opt_) {
        std::vector<int32_t> dst((size_t)3 * n_mlp_, -1);
        auto invert = [&](const int32_t* src, size_t n_slots, int32_t* out) {
            for (size_t j = 0; j < n_slots; j++) {
                if (src[j] < 0) continue;
                if ((uint32_t)src[j] >= n_mlp_ || out[src[j]] >= 0) { fused_opt_ = false; return; }      // not a one-to-one image
                out[src[j]] = (int32_t)j;
            }
        };
        std::vector<int32_t> si_host(sf.size());
        NRC_HIP(hipMemcpy(si_host.data(), d_src_inf_, sf.size() * 4, hipMemcpyDeviceToHost));
        invert(sf.data(), sf.size(), dst.data());
        if (fused_opt_) invert(si_host.data(), si_host.size(), dst.data() + n_mlp_);
        if (fused_opt_) invert(sb.data(), sb.size(), dst.data() + 2 * (size_t)n_mlp_);
        if (fused_opt_) {
            dev_alloc(&d_dst_, dst.size() * 4, "d_dst_");
            NRC_HIP(hipMemcpy(d_dst_, dst.data(), dst.size() * 4, hipMemcpyHostToDevice));
        }
    }
    dev_alloc(&d_pk_fwd_, sf.size() * 2, "d_pk_fwd_");
    dev_alloc(&d_pk_bwd_, sb.size() * 2, "d_pk_bwd_");
    if (hash_) {
        dev_alloc(&d_t16_train_, (size_t)n_grid_entries_ * 4, "d_t16_train_");
        for (auto& p : d_t16_ema_) dev_alloc(&p, (size_t)n_grid_entries_ * 4, "p");
        dev_alloc(&d_grad16_, (size_t)n_grid_entries_ * 4, "d_grad16_");
    }
    repack(nullptr);
    NRC_HIP(hipStreamSynchronize(nullptr));
}

Mlp::~Mlp()
{
    if (d_src_inf_ != d_src_fwd_ && d_src_inf_) dev_free(d_src_inf_);
    if (d_dst_) dev_free(d_dst_);
    void* ptrs[] = {d_w_, d_ema_, d_m_, d_v_, d_grad_, d_pk_infer_[0], d_pk_infer_[1], d_pk_fwd_, d_pk_bwd_, d_src_fwd_,
                    d_src_bwd_, d_acts_, d_deltas_, d_slabs_, d_loss_part_, d_tiles_, d_tasks_, d_feat_[0], d_feat_[1], d_feat_[2], d_feat_[3], d_t16_train_,
                    d_t16_ema_[0], d_t16_ema_[1], d_denc_, d_grad16_, d_grid_lists_, d_grid_counters_, d_grid_bin_entry0_, d_grid_fix_};
    for (void* p : ptrs)
        if (p) dev_free(p);
}

float* Mlp::buffer(int which)
{
    switch (which) {
    case 0: return d_w_;
    case 1: return d_ema_;
    case 2: return d_m_;
    case 3: return d_v_;
    case 4: return d_grad_;
    default: fail("bad parameter buffer id");
    }
}

void Mlp::to_tcnn_layout(int which, const float* own, float* tcnn) const
{
    if (which < 0 || which > 4) fail("bad parameter buffer id");
    const size_t n_out = (size_t)3 * width_, head = n_mlp_ - n_out, dead = (size_t)13 * width_;
    std::memcpy(tcnn, own, (head + n_out) * sizeof(float));
    if (which < 4) std::memcpy(tcnn + n_mlp_, tcnn_dead_rows_[which].data(), dead * sizeof(float));
    else std::memset(tcnn + n_mlp_, 0, dead * sizeof(float));
    std::memcpy(tcnn + n_mlp_ + dead, own + n_mlp_, (size_t)(n_params_ - n_mlp_) * sizeof(float));
}

void Mlp::from_tcnn_layout(int which, const float* tcnn, float* own)
{
    if (which < 0 || which > 4) fail("bad parameter buffer id");
    const size_t dead = (size_t)13 * width_;
    std::memcpy(own, tcnn, (size_t)n_mlp_ * sizeof(float));
    if (which < 4) std::memcpy(tcnn_dead_rows_[which].data(), tcnn + n_mlp_, dead * sizeof(float));
    std::memcpy(own + n_mlp_, tcnn + n_mlp_ + dead, (size_t)(n_params_ - n_mlp_) * sizeof(float));
}

// The inference (EMA) image and table are double-buffered: this writes the set inference is NOT reading and then makes it
// current for every inference enqueued from now on.  An inference pass enqueued earlier keeps reading the other set, so the
// optimizer of frame N never has to wait for frame N's inference (only for frame N-1's, which read the set written here).
void Mlp::repack(hipStream_t s)
{
    const uint32_t nf = n_frag_fwd_ * 512, nb = n_frag_bwd_ * 512;
    const uint32_t nmax = nf > nb ? nf : nb;
    const int next = infer_set_ ^ 1;
    hipLaunchKernelGGL(k_pack, dim3(ceil_div(nmax, 256)), dim3(256), 0, s, d_w_, d_ema_, d_src_fwd_, d_src_inf_, nf, d_src_bwd_, nb,
                       (half_t*)d_pk_infer_[next], (half_t*)d_pk_fwd_, (half_t*)d_pk_bwd_);
    if (hash_)
        hipLaunchKernelGGL(k_pack_grid, dim3(ceil_div(n_grid_entries_, 256)), dim3(256), 0, s, d_w_ + n_mlp_, d_ema_ + n_mlp_,
                           (uint32_t*)d_t16_train_, (uint32_t*)d_t16_ema_[next], n_grid_entries_);
    NRC_HIP(hipGetLastError());
    infer_set_ = next;
}

// generic-path encoding launch: HashGrid gathers from the fp16 table copy that belongs to the weight set in use
// `slot` 0 = inference, 1 = training: the two may run concurrently on different streams and own separate feature buffers
void Mlp::launch_features(const float* d_in, uint32_t n, bool use_ema, int slot, hipStream_t s, bool skip_zero, const uint32_t* live_list,
                          const uint32_t* live_count)
{
    ensure_features(n, slot);
    half_t* feat = (half_t*)d_feat_[slot];
    if (!hash_) {
        launch_encode(cfg_.pos_id, cfg_.dir_id, s, d_in, feat, n, skip_zero);
        return;
    }
    HashLevels lv;
    for (uint32_t l = 0; l <= HG_LEVELS; l++) lv.off[l] = hg_off_[l];
    const uint32_t* tab = (const uint32_t*)(use_ema ? d_t16_ema_[infer_set_] : d_t16_train_);
    if (slot == 0 && live_list != nullptr && skip_zero) {      // renderer inference with the frame's live-query list
        const uint32_t n_slots = enc_dims_ / 2;
        dim3 g((uint32_t)num_cus() * 2u, n_slots);
        uint32_t xc = 0;
        if (xcd8_ && n_slots % 8u == 0u) {      // a level's table is gathered from one XCD (see the kernel)
            xc = (uint32_t)num_cus() * 2u;               // chunks of the list per slot: as many workgroups per slot as before
            g = dim3(8u * xc * (n_slots / 8u), 1);
        }
        if (cfg_.dir_id == 0) hipLaunchKernelGGL(k_encode_hash_list<0>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, live_list, live_count, xc);
        else if (cfg_.dir_id == 1) hipLaunchKernelGGL(k_encode_hash_list<1>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, live_list, live_count, xc);
        else hipLaunchKernelGGL(k_encode_hash_list<2>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, live_list, live_count, xc);
        return;
    }
    if (slot == 0) {         // inference: level-major gathers and feature layout (k_encode_hash_lm / k_infer_gen<..., true>)
        const dim3 g(ceil_div(n, 256), enc_dims_ / 2);
        const int sk = skip_zero ? 1 : 0;
        if (cfg_.dir_id == 0) hipLaunchKernelGGL(k_encode_hash_lm<0>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, sk);
        else if (cfg_.dir_id == 1) hipLaunchKernelGGL(k_encode_hash_lm<1>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, sk);
        else hipLaunchKernelGGL(k_encode_hash_lm<2>, g, dim3(256), 0, s, d_in, tab, (uint32_t*)feat, n, lv, sk);
        return;
    }
    const bool level_major2 = !xcd8_;      // (the level-per-XCD mapping is written for eight XCDs; otherwise every XCD gathers from every level)
    const dim3 g(level_major2 ? ceil_div(n * 16u, 256) : ceil_div(n, 128) * 8u);
    const int mode = (skip_zero ? 1 : 0) | (level_major2 ? 0 : 2);
    if (cfg_.dir_id == 0) hipLaunchKernelGGL(k_encode_hash<0>, g, dim3(256), 0, s, d_in, tab, feat, n, lv, mode);
    else if (cfg_.dir_id == 1) hipLaunchKernelGGL(k_encode_hash<1>, g, dim3(256), 0, s, d_in, tab, feat, n, lv, mode);
    else hipLaunchKernelGGL(k_encode_hash<2>, g, dim3(256), 0, s, d_in, tab, feat, n, lv, mode);
}

// this translation unit's copy of the run-time priority switch (nrc_common.hpp)
void mlp_set_wave_priority_raise(int on)
{
    NRC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_raise_wave_priority), &on, sizeof(int)));
}

// CU count of the device this Mlp lives on (one instance per GPU: no process-wide cache)
int Mlp::num_cus()
{
    if (num_cus_ == 0) {
        int dev = 0;
        NRC_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        NRC_HIP(hipGetDeviceProperties(&prop, dev));
        num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return num_cus_;
}

template <int THREADS, int NT>
static void launch_infer(uint32_t blocks, size_t lds, hipStream_t s, const float* d_in, float* d_out, uint32_t n, const uint4* img,
                         int skip_zero, const CompositeArgs* composite = nullptr, const uint32_t* live_list = nullptr,
                         const uint32_t* live_count = nullptr)
{
    if (composite != nullptr) {
        launch_last(k_infer<6, THREADS, NT, 0, true>, dim3(blocks), dim3(THREADS), (uint32_t)lds, s, d_in, d_out, n, img,
                    (unsigned long long*)nullptr, skip_zero, *composite, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        return;
    }
    if (live_list != nullptr)
        launch_last(k_infer<6, THREADS, NT, 0, false, true>, dim3(blocks), dim3(THREADS), (uint32_t)lds, s, d_in, d_out, n, img,
                           (unsigned long long*)nullptr, skip_zero, CompositeArgs{}, live_list, live_count);
    else
        launch_last(k_infer<6, THREADS, NT>, dim3(blocks), dim3(THREADS), (uint32_t)lds, s, d_in, d_out, n, img,
                           (unsigned long long*)nullptr, skip_zero, CompositeArgs{}, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
}

// fp16 feature buffer of the generic path ([n][E16]); grows on demand (never inside a captured region: first use sizes it)
void Mlp::ensure_features(uint32_t n, int slot)
{
    if (n <= feat_n_[slot]) return;
    if (d_feat_[slot]) {
        NRC_HIP(hipDeviceSynchronize());      // a kernel on another stream may still read the old buffer
        dev_free(d_feat_[slot]);
    }
    d_feat_[slot] = nullptr;
    dev_alloc(&d_feat_[slot], (size_t)n * enc_dims_ * 2, "d_feat_[slot]");
    feat_n_[slot] = n;
}

void Mlp::infer(const float* d_in, float* d_out, uint32_t n, bool use_ema, hipStream_t s, bool skip_zero_queries, const CompositeArgs* composite,
                const uint32_t* live_list, const uint32_t* live_count)
{
    if (n == 0) return;
    if (composite != nullptr && !fused_) throw std::logic_error("SkyRenderer ERROR: compositing epilogue asked of a generic model");
    const uint4* img = (const uint4*)(use_ema ? d_pk_infer_[infer_set_] : d_pk_fwd_);
    if (!fused_) {
        // EMA inference of a Frequency(12) + OneBlob(4) model encodes inside the MLP kernel (image in the encoder's input order);
        // everything else runs the encoding kernel first
        const bool enc80 = enc80_generic_ && use_ema;
        if (!enc80) launch_features(d_in, n, use_ema, 0, s, skip_zero_queries, live_list, live_count);
        const float* skip_in = skip_zero_queries ? d_in : nullptr;
        const uint32_t* list = skip_zero_queries ? live_list : nullptr;      // renderer inference with the frame's live-query list (k_infer_gen)
        const half_t* feat = enc80 ? nullptr : (const half_t*)d_feat_[0];
        const int ks0 = (int)enc_dims_ / 16;
        uint32_t blocks = ceil_div(ceil_div(n, 32), 8);
        auto launch = [&](auto kernel, uint32_t threads, size_t lds, uint32_t per_cu) {
            const uint32_t cap = (uint32_t)num_cus() * per_cu;
            launch_last(kernel, dim3(blocks > cap ? cap : blocks), dim3(threads), (uint32_t)lds, s, feat, d_out, n, img, (int)depth_, ks0, skip_in, d_in,
                        list, list != nullptr ? live_count : nullptr);      // (the stage's last launch: takes the armed event along, nrc_common.hpp)
        };
        if (kw_ == 32) {
            if (enc80) launch(k_infer_gen<32, 256, false, 2, true>, 256, 2 * 5 * 1024, 4);
            else if (hash_) launch(k_infer_gen<32, 256, true>, 256, 2 * 5 * 1024, 4);
            else launch(k_infer_gen<32, 256, false>, 256, 2 * 5 * 1024, 4);
        } else if (kw_ == 64) {           // 4 waves x 2 tiles = 256 samples per workgroup pass, 20 KB of LDS
            if (enc80) launch(k_infer_gen<64, 256, false, 2, true>, 256, 2 * 10 * 1024, 4);
            else if (hash_) launch(k_infer_gen<64, 256, true>, 256, 2 * 10 * 1024, 4);
            else launch(k_infer_gen<64, 256, false>, 256, 2 * 10 * 1024, 4);
        } else {
            // 8 waves x 1 tile = 256 samples per staged layer (32 KB), 64 KB of LDS.  One tile per wave and one 32-row block of a
            // layer at a time keep the kernel at 134-154 VGPRs without scratch (two tiles per wave spilled 27 VGPRs): a workgroup
            // fits on a CU beside two gen_rays waves per SIMD.  The staged layers come out of the L2 (the 250 KB image is resident
            // there): 1 KB per sample instead of 0.5 KB, still far below what per-tile fragment fetches cost (7.8 KB).
            // Renderer inference (skip_in: the launch runs BESIDE gen_rays, whose five waves per SIMD leave room for one 168-VGPR wave
            // per SIMD only after two of them have retired) uses 4-wave workgroups -- one wave per SIMD instead of two: configs[4]'s frame
            // 0.456 -> 0.439 ms -- although they are 12 % slower on a dense launch (0.54 -> 0.61 ms: half the samples per staged
            // layer), which keeps the 8-wave form.
            const size_t lds = 2 * 32 * 1024;
            auto launch128 = [&](auto threads_c) {
                constexpr int T128 = decltype(threads_c)::value;
                bool& attr_set = T128 == 256 ? attr_infer4_set_ : attr_infer_set_;
                if (!attr_set) {      // per instance = per device: the attribute belongs to the device's code object
                    NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_infer_gen<128, T128, false, 1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_infer_gen<128, T128, true, 1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_infer_gen<128, T128, false, 1, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    attr_set = true;
                }
                blocks = ceil_div(ceil_div(n, 32), T128 / 64);
                constexpr uint32_t per_cu = T128 == 256 ? NRC_GEN128_WG4_PER_CU : 2;
                if (enc80) launch(k_infer_gen<128, T128, false, 1, true>, T128, lds, per_cu);
                else if (hash_) launch(k_infer_gen<128, T128, true, 1>, T128, lds, per_cu);
                else launch(k_infer_gen<128, T128, false, 1>, T128, lds, per_cu);
            };
#ifdef NRC_GEN128_THREADS_FORCED
            launch128(std::integral_constant<int, NRC_GEN128_THREADS>{});
#else
            if (skip_in != nullptr) launch128(std::integral_constant<int, 256>{});
            else launch128(std::integral_constant<int, 512>{});
#endif
        }
        NRC_HIP(hipGetLastError());
        return;
    }
    // persistent workgroups sharing one 54 KB weight image in LDS: 8 waves x 2 tiles per iteration, two workgroups per CU.
    // The shape sweeps and ablations that led here (256...1024 threads, 1...4 tiles, no-encode / no-ReLU variants, in-kernel
    // clock stamps) exist only in the diagnostic build (make EXTRA=-DNRC_DIAG OUT=../lib_diag; tools/bench_mlp.py).
    int threads = 512, nt = 2, bpc = 2;
#ifdef NRC_DIAG
    static const int e_threads = [] { const char* e = getenv("NRC_INFER_THREADS"); return e ? atoi(e) : 512; }();
    static const int e_nt = [] { const char* e = getenv("NRC_INFER_NT"); return e ? atoi(e) : 2; }();
    static const int e_bpc = [] { const char* e = getenv("NRC_INFER_BPC"); return e ? atoi(e) : 2; }();
    threads = e_threads; nt = e_nt; bpc = e_bpc;
#endif
    const uint32_t n_tiles = ceil_div(n, 32);
    const uint32_t max_blocks = (uint32_t)num_cus() * (uint32_t)bpc;
    uint32_t blocks = ceil_div(n_tiles, (uint32_t)(threads / 64) * (uint32_t)nt);
    if (blocks > max_blocks) blocks = max_blocks;
    const size_t lds = (size_t)n_frag_fwd_ * 1024;
    const int sz = skip_zero_queries ? 1 : 0;
#ifdef NRC_DIAG
    static const int abl = [] { const char* e = getenv("NRC_INFER_ABL"); return e ? atoi(e) : 0; }();
    if (abl != 0) {       // ablations / in-kernel clock: tools/bench_mlp.py
        infer_diagnostic(abl, blocks, lds, s, d_in, d_out, n, img);
        return;
    }
    if (threads == 1024 && nt == 1) launch_infer<1024, 1>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 256 && nt == 1) launch_infer<256, 1>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 256 && nt == 2) launch_infer<256, 2>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 512 && nt == 1) launch_infer<512, 1>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 1024 && nt == 2) launch_infer<1024, 2>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 512 && nt == 3) launch_infer<512, 3>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 256 && nt == 3) launch_infer<256, 3>(blocks, lds, s, d_in, d_out, n, img, sz);
    else if (threads == 256 && nt == 4) launch_infer<256, 4>(blocks, lds, s, d_in, d_out, n, img, sz);
    else
#endif
    // Renderer-mode launches (skip_zero_queries: the launch runs BESIDE gen_rays) of up to a 1080p frame's worth of queries use 4-wave
    // workgroups -- one 116-register wave per SIMD instead of two: a workgroup finds room once ONE of a SIMD's five camera waves has
    // retired (the same reason the 128-wide kernel does it, above).  Same box, default preset: 7 890-7 940 -> 7 990-8 020 Msamples/s;
    // 2-wave workgroups lose (7 500: twice the weight-image traffic), and the 8.3 M queries of a whole 4K frame on one GPU prefer the
    // 8-wave form (7 970 against 7 820).  NRC_INFER_RENDER_THREADS=512 restores it everywhere.
#ifndef NRC_INFER_RENDER_THREADS
#define NRC_INFER_RENDER_THREADS 256
#endif
    if (NRC_INFER_RENDER_THREADS != 512 && skip_zero_queries && composite == nullptr && n <= (3u << 20)) {
        constexpr int T = NRC_INFER_RENDER_THREADS;
        uint32_t b2 = ceil_div(n_tiles, (uint32_t)(T / 64) * 2u);
        const uint32_t cap = (uint32_t)num_cus() * 2u;
        launch_infer<T, 2>(b2 > cap ? cap : b2, lds, s, d_in, d_out, n, img, sz, nullptr, live_list, live_count);
        NRC_HIP(hipGetLastError());
        return;
    }
    launch_infer<512, 2>(blocks, lds, s, d_in, d_out, n, img, sz, composite, skip_zero_queries && composite == nullptr ? live_list : nullptr, live_count);
    NRC_HIP(hipGetLastError());
}

#ifdef NRC_DIAG
// diagnostics of the fused inference kernel (never on the product path): ABL 1/2/3 drop the encoding / ReLU-convert work,
// ABL 4 stamps s_memtime / s_memrealtime per workgroup and prints the in-kernel clock and inter-kernel gaps
void Mlp::infer_diagnostic(int abl, uint32_t blocks, size_t lds, hipStream_t s, const float* d_in, float* d_out, uint32_t n,
                           const void* image)
{
    const uint4* img = (const uint4*)image;
    if (abl == 4) {
        static unsigned long long* d_st = nullptr;
        constexpr int SLOTS = 16;
        if (!d_st) dev_alloc(&d_st, (size_t)SLOTS * 4 * 2048 * sizeof(unsigned long long), "d_st");
        static int count = 0;
        const int slot = count % SLOTS;
        hipLaunchKernelGGL((k_infer<6, 512, 2, 4>), dim3(blocks), dim3(512), lds, s, d_in, d_out, n, img, d_st + (size_t)slot * 4 * 2048);
        if (++count % 64 == 0) {
            std::vector<unsigned long long> h((size_t)SLOTS * 4 * 2048);
            NRC_HIP(hipStreamSynchronize(s));
            NRC_HIP(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<std::pair<unsigned long long, unsigned long long>> span;
            std::vector<double> clk;
            for (int k = 0; k < SLOTS; k++) {
                unsigned long long first = ~0ull, last = 0;
                for (uint32_t b = 0; b < blocks; b++) {
                    const unsigned long long* e = &h[((size_t)k * 2048 + b) * 4];
                    first = std::min(first, e[2]);
                    last = std::max(last, e[2] + e[1]);
                    clk.push_back((double)e[0] / (double)e[1] * 100.0);
                }
                span.push_back({first, last});
            }
            std::sort(span.begin(), span.end());
            std::sort(clk.begin(), clk.end());
            fprintf(stderr, "[nrc diag] clock MHz med %.0f | per launch (wave-0 span us, gap to next us):", clk[clk.size() / 2]);
            for (int k = 0; k + 1 < SLOTS; k++)
                fprintf(stderr, " (%.1f, %.1f)", (double)(span[k].second - span[k].first) / 100.0,
                        (double)(span[k + 1].first - span[k].second) / 100.0);
            fprintf(stderr, "\n");
            // one launch in detail: when workgroups start, how long the LDS staging takes, when they end (us from the first start)
            {
                std::vector<double> st, sg, en;
                const unsigned long long* base = &h[0];
                unsigned long long first = ~0ull;
                for (uint32_t b = 0; b < blocks; b++) first = std::min(first, base[4 * b + 2]);
                for (uint32_t b = 0; b < blocks; b++) {
                    st.push_back((double)(base[4 * b + 2] - first) / 100.0);
                    sg.push_back((double)base[4 * b + 3] / 100.0);
                    en.push_back((double)(base[4 * b + 2] + base[4 * b + 1] - first) / 100.0);
                }
                std::sort(st.begin(), st.end()); std::sort(sg.begin(), sg.end()); std::sort(en.begin(), en.end());
                auto q = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (double)(v.size() - 1))]; };
                fprintf(stderr, "[nrc diag] workgroup start us: med %.1f p90 %.1f max %.1f | staging us: min %.1f med %.1f max %.1f | end us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f\n",
                        q(st, .5), q(st, .9), q(st, 1), q(sg, 0), q(sg, .5), q(sg, 1), q(en, 0), q(en, .1), q(en, .5), q(en, .9), q(en, 1));
            }
        }
    } else if (abl == 1) hipLaunchKernelGGL((k_infer<6, 512, 1, 1>), dim3(blocks), dim3(512), lds, s, d_in, d_out, n, img);
    else if (abl == 2) hipLaunchKernelGGL((k_infer<6, 512, 1, 2>), dim3(blocks), dim3(512), lds, s, d_in, d_out, n, img);
    else hipLaunchKernelGGL((k_infer<6, 512, 1, 3>), dim3(blocks), dim3(512), lds, s, d_in, d_out, n, img);
    NRC_HIP(hipGetLastError());
}
#endif

// the row-block tasks of k_wgrad2: one per (layer, 32-row block of its delta, group of up to four 32-column blocks of its input)
void Mlp::build_wgrad_tasks()
{
    if (d_tasks_) return;
    // Which weight-gradient kernel (and, in backward(), which generic fwd/bwd kernel): round 4's k_wgrad2 / k_train_gen2 are the faster
    // ones alone at every width (8x128: 138 -> 64 us per step) and what the 128-wide frame needs (configs[4] 6 850 against 5 750 Msamples/s
    // with round 3's pair); beside gen_rays the 64-wide models run better with round 3's k_wgrad / k_train_gen -- fewer, lighter workgroups
    // (default preset + 0.7 %, HashGrid + 3.2 %, TriangleWave 64 + 2.5 %, A/B on one box).  The renderer's frame is what the library is
    // for: round 3's kernels up to 64 neurons, round 4's for 128.  NRC_DEBUG=wgrad_old / NRC_DEBUG=train_gen_old = 0 | 1 override.
    wgrad_old_ = debug_value("wgrad_old", kw_ <= 64 ? 1 : 0) != 0;
    std::vector<WgradTask> tasks;
    const uint32_t D = depth_;
    for (uint32_t l = 0; l <= D; l++) {
        const MlpLayer& L = layers_[l];
        const uint32_t a_rows = l == D ? D * kw_ : l * kw_;                       // delta_l rows (kernel width: 32 for width 16)
        const uint32_t b_rows = l == 0 ? 0 : enc_dims_ + (l - 1) * kw_;           // a_{l-1} rows (enc for l = 0)
        for (uint32_t mt = 0; mt * 32 < L.out; mt++)
            for (uint32_t n0 = 0; n0 * 32 < L.in; n0 += WGRAD2_NTL) {
                WgradTask T;
                const uint32_t tiles_left = (L.in - 32 * n0 + 31) / 32;
                T.n_tiles = tiles_left < (uint32_t)WGRAD2_NTL ? tiles_left : (uint32_t)WGRAD2_NTL;
                T.a_row0 = a_rows + 32 * mt;
                T.b_row0 = b_rows + 32 * n0;
                T.m_valid = L.out - 32 * mt < 32 ? L.out - 32 * mt : 32;
                const uint32_t last0 = 32 * (n0 + T.n_tiles - 1);
                T.n_valid_last = L.in - last0 < 32 ? L.in - last0 : 32;
                T.param_off = L.off + 32 * mt * L.in + 32 * n0;
                T.in_dim = L.in;
                T.pad = 0;
                tasks.push_back(T);
            }
    }
    n_wgrad_tasks_ = (int)tasks.size();
    dev_alloc(&d_tasks_, tasks.size() * sizeof(WgradTask), "d_tasks_");
    NRC_HIP(hipMemcpy(d_tasks_, tasks.data(), tasks.size() * sizeof(WgradTask), hipMemcpyHostToDevice));
}
// samples per K-chunk of the weight-gradient launch: ~2 workgroups per CU, a multiple of 32 samples, at least 64
uint32_t Mlp::wgrad_chunk(uint32_t n)
{
    if (wgrad_old_) return WGRAD_CHUNK;
    const uint32_t wg_per_chunk = ceil_div((uint32_t)n_wgrad_tasks_, (uint32_t)WGRAD2_WAVES);
    uint32_t chunks = std::max(1u, 2u * (uint32_t)num_cus() / std::max(1u, wg_per_chunk));
    uint32_t c = ceil_div(ceil_div(n, chunks), 32u) * 32u;
    return std::max(c, 64u);
}

void Mlp::ensure_train_workspace(uint32_t n)
{
    if (n <= ws_n_) return;
    if (d_acts_) dev_free(d_acts_);
    if (d_deltas_) dev_free(d_deltas_);
    if (d_slabs_) dev_free(d_slabs_);
    if (d_loss_part_) dev_free(d_loss_part_);
    d_acts_ = d_deltas_ = nullptr;
    d_slabs_ = d_loss_part_ = nullptr;
    const size_t rows_a = enc_dims_ + (size_t)depth_ * kw_, rows_d = (size_t)depth_ * kw_ + 8;
    dev_alloc(&d_acts_, rows_a * n * 2, "d_acts_");
    dev_alloc(&d_deltas_, rows_d * n * 2, "d_deltas_");
    NRC_HIP(hipMemset(d_deltas_, 0, rows_d * n * 2));
    build_wgrad_tasks();
    // (k_wgrad2: a smaller batch has a smaller chunk, never more chunks than the launch's target count)
    const uint32_t max_chunks = wgrad_old_ ? ceil_div(n, WGRAD_CHUNK)
                                           : std::max(ceil_div(n, wgrad_chunk(n)), std::max(1u, 2u * (uint32_t)num_cus() / std::max(1u, ceil_div((uint32_t)n_wgrad_tasks_, (uint32_t)WGRAD2_WAVES))));
    dev_alloc(&d_slabs_, (size_t)max_chunks * n_mlp_ * 4, "d_slabs_");
    if (hash_) {
        if (d_denc_) dev_free(d_denc_);
        d_denc_ = nullptr;
        dev_alloc(&d_denc_, (size_t)n * 32 * 2, "d_denc_");
        // bin lists of the table gradient (k_grid_scatter): levels of at least GB_MIN_BINS bins of GB_BIN entries; a level's lists hold twice
        // what the level sends a bin on average (n * 8 / bins pairs); the rest goes into the fixed-point shadow
        if (d_grid_lists_) dev_free(d_grid_lists_);
        d_grid_lists_ = nullptr;
        if (d_grid_bin_entry0_) dev_free(d_grid_bin_entry0_);
        d_grid_bin_entry0_ = nullptr;
        if (d_grid_counters_) dev_free(d_grid_counters_);
        d_grid_counters_ = nullptr;
        grid_bins_total_ = 0;
        const bool no_bins = debug_switch("grid_no_bins");      // tests: every pair through the fixed-point shadow
        std::vector<uint32_t> info;      // uint4 per bin: {first entry, pair offset of the list, capacity, 0}
        size_t pairs = 0;
        for (uint32_t l = 0; l < HG_LEVELS; l++) {
            const uint32_t cnt = hg_off_[l + 1] - hg_off_[l];
            const uint32_t bins = (cnt % GB_BIN == 0u) ? cnt / GB_BIN : 0u;
            grid_bin_first_[l] = grid_bins_total_;
            grid_bin_count_[l] = (bins >= GB_MIN_BINS && bins <= GB_MAX_BINS && !no_bins) ? bins : 0u;
            grid_bin_cap_[l] = grid_bin_count_[l] ? std::max(1024u, (uint32_t)std::min<uint64_t>(2ull * n * 8ull / grid_bin_count_[l], 1u << 24)) : 0u;
            grid_list0_[l] = (uint32_t)pairs;
            for (uint32_t b = 0; b < grid_bin_count_[l]; b++) {
                info.insert(info.end(), {hg_off_[l] + b * GB_BIN, (uint32_t)(pairs + (size_t)b * grid_bin_cap_[l]), grid_bin_cap_[l], 0u});
            }
            pairs += (size_t)grid_bin_count_[l] * grid_bin_cap_[l];
            grid_bins_total_ += grid_bin_count_[l];
        }
        if (pairs > 0xffffffffull) fail("train batch too large for the table gradient's bin lists");
        if (grid_bins_total_ != 0u) {
            dev_alloc(&d_grid_lists_, pairs * 8, "d_grid_lists_");
            dev_alloc(&d_grid_counters_, (size_t)grid_bins_total_ * 4, "d_grid_counters_");
            NRC_HIP(hipMemset(d_grid_counters_, 0, (size_t)grid_bins_total_ * 4));
            dev_alloc(&d_grid_bin_entry0_, info.size() * 4, "d_grid_bin_info_");
            NRC_HIP(hipMemcpy(d_grid_bin_entry0_, info.data(), info.size() * 4, hipMemcpyHostToDevice));
        }
        // the fixed-point shadow of the table: two int64 per entry, all zero between two steps (k_grid_gather clears what it folds in)
        if (!d_grid_fix_) {
            dev_alloc(&d_grid_fix_, (size_t)n_grid_entries_ * 16, "d_grid_fix_");
            NRC_HIP(hipMemset(d_grid_fix_, 0, (size_t)n_grid_entries_ * 16));
        }
    }
    dev_alloc(&d_loss_part_, (size_t)(n / 32) * 4, "d_loss_part_");
    ws_n_ = n;
    if (!d_tiles_) {
        std::vector<WgradTile> tiles;
        const uint32_t D = depth_;
        for (uint32_t l = 0; l <= D; l++) {
            const MlpLayer& L = layers_[l];
            const uint32_t a_rows = l == D ? D * kw_ : l * kw_;                       // delta_l rows (kernel width: 32 for width 16)
            const uint32_t b_rows = l == 0 ? 0 : enc_dims_ + (l - 1) * kw_;           // a_{l-1} rows (enc for l = 0)
            for (uint32_t mt = 0; mt * 32 < L.out; mt++)
                for (uint32_t nt = 0; nt * 32 < L.in; nt++) {
                    WgradTile T;
                    T.a_row0 = a_rows + 32 * mt;
                    T.b_row0 = b_rows + 32 * nt;
                    T.m_valid = L.out - 32 * mt < 32 ? L.out - 32 * mt : 32;
                    T.n_valid = L.in - 32 * nt < 32 ? L.in - 32 * nt : 32;
                    T.param_off = L.off + 32 * mt * L.in + 32 * nt;
                    T.in_dim = L.in;
                    tiles.push_back(T);
                }
        }
        n_wgrad_tiles_ = (int)tiles.size();
        dev_alloc(&d_tiles_, tiles.size() * sizeof(WgradTile), "d_tiles_");
        NRC_HIP(hipMemcpy(d_tiles_, tiles.data(), tiles.size() * sizeof(WgradTile), hipMemcpyHostToDevice));
    }
}

const void* Mlp::pre_encode(const float* d_in, uint32_t n, int parity, hipStream_t s)
{
    if (!can_pre_encode()) fail("pre_encode: the model's encoding is trainable or computed inside its kernels");
    const int slot = 2 + (parity & 1);
    launch_features(d_in, n, false, slot, s, false);
    return d_feat_[slot];
}

void Mlp::backward(const float* d_in, const float* d_target, uint32_t n, uint32_t n_norm, hipStream_t s, bool widen_grid_grad, const void* features_ready)
{
    if (n == 0 || n % 32 != 0) fail("training batch must be a non-zero multiple of 32 samples");
    ensure_train_workspace(n);
    const uint32_t n_tiles = n / 32;
    constexpr int THREADS = 256;
    uint32_t blocks = ceil_div(n_tiles, THREADS / 64);
    if (fused_) {
        TrainArgs a;
        a.in = d_in;
        a.target = d_target;
        a.n = n;
        a.inv_n_total = (float)(1.0 / (3.0 * (double)n_norm));      // 3 * n_norm can exceed 32 bits
        a.loss_id = loss_id_;
        a.acts = (half_t*)d_acts_;
        a.deltas = (half_t*)d_deltas_;
        a.loss_part = d_loss_part_;
        if (blocks > (uint32_t)num_cus()) blocks = (uint32_t)num_cus();
        const size_t lds = (size_t)std::max(n_frag_fwd_, n_frag_bwd_) * 1024;      // one image at a time
        auto kernel = k_train_fwd_bwd_light<6, THREADS>;
        if (!attr_train_set_) {
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_train_set_ = true;
        }
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), lds, s, a, (const uint4*)d_pk_fwd_, (const uint4*)d_pk_bwd_);
    } else {
        if (features_ready == nullptr) launch_features(d_in, n, false, 1, s, false);
        if (hash_ && !grad16_clean_) NRC_HIP(hipMemsetAsync(d_grad16_, 0, (size_t)n_grid_entries_ * 4, s));      // (k_grid_opt leaves it clean)
        grad16_clean_ = false;
        TrainArgsGen a;
        a.feat = features_ready != nullptr ? (const half_t*)features_ready : (const half_t*)d_feat_[1];
        a.target = d_target;
        a.n = n;
        a.inv_n_total = (float)(1.0 / (3.0 * (double)n_norm));      // 3 * n_norm can exceed 32 bits
        a.loss_id = loss_id_;
        a.acts = (half_t*)d_acts_;
        a.deltas = (half_t*)d_deltas_;
        a.loss_part = d_loss_part_;
        a.depth = (int)depth_;
        a.ks0 = (int)enc_dims_ / 16;
        a.d_enc = hash_ ? (half_t*)d_denc_ : nullptr;
        // a training batch is a few hundred 32-sample tiles (16 384 rays = 512): four-wave workgroups (128 of them) stream each
        // layer once per four tiles; large batches use eight waves, two workgroups per CU
        const bool small = n_tiles <= (uint32_t)num_cus() * 8u;
        const uint32_t waves = small ? 4u : 8u;
        blocks = ceil_div(n_tiles, waves);
        const uint32_t cap = (uint32_t)num_cus() * 2u;
        if (blocks > cap) blocks = cap;
        const int mtg = (int)kw_ / 32, ksg = (int)kw_ / 16;
        // two weight stages + per wave: the k-group transposition scratch and one 64-bit ReLU mask per lane and layer
        const size_t lds = (size_t)2 * mtg * (ksg > 5 ? ksg : 5) * 1024 + waves * (KG_SCRATCH + (size_t)depth_ * 512);
        if (lds > 160 * 1024) fail("network too deep for the training kernel's LDS budget");
        if (!attr_train_set_) {      // per instance = per device: the attribute belongs to the device's code object
            const int cap_lds = 160 * 1024;
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<32, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen<128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
            attr_train_set_ = true;
        }
        const uint4 *fw = (const uint4*)d_pk_fwd_, *bw = (const uint4*)d_pk_bwd_;
        // round 4: rows split over the waves (k_train_gen2); NRC_DEBUG=train_gen_old=0|1 picks k_train_gen2 / k_train_gen, NRC_DEBUG=train_gen_nt|2 sets the tiles
        // per sample group
        // (the environment is read per call: a test compares the kernels inside one process)
        const bool gen_old = debug_value("train_gen_old", kw_ <= 64 ? 1 : 0) != 0;      // (see build_wgrad_tasks)
        const int gen_nt_env = (int)debug_value("train_gen_nt", 0);
        if (!gen_old) {
            const uint32_t sgn = 4u / (uint32_t)mtg;                 // sample groups per workgroup
            // (a 16 384-ray batch of an 8x128 net, stand-alone: 29.5 us with one tile per sample group, 39.8 with two; two halve the weight
            // traffic from L2 -- 33 KB per stage and sample group -- and take over where one tile would put more than four workgroups on a CU)
            const int nt = gen_nt_env == 1 || gen_nt_env == 2 ? gen_nt_env : (n_tiles > 4u * sgn * (uint32_t)num_cus() ? 2 : 1);
            uint32_t groups = ceil_div(n_tiles, sgn * (uint32_t)nt);
            if (groups > cap) groups = cap;
            const size_t lds2 = (size_t)2 * sgn * nt * ksg * 1024 + 4 * KG_SCRATCH + (size_t)4 * depth_ * nt * 256;
            if (lds2 > 160 * 1024) fail("network too deep for the training kernel's LDS budget");
            if (!attr_train2_set_) {
                const int cap_lds = 160 * 1024;
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_gen2<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap_lds));
                attr_train2_set_ = true;
            }
            if (kw_ == 32) {
                if (nt == 1) hipLaunchKernelGGL((k_train_gen2<32, 1>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
                else hipLaunchKernelGGL((k_train_gen2<32, 2>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
            } else if (kw_ == 64) {
                if (nt == 1) hipLaunchKernelGGL((k_train_gen2<64, 1>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
                else hipLaunchKernelGGL((k_train_gen2<64, 2>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
            } else {
                if (nt == 1) hipLaunchKernelGGL((k_train_gen2<128, 1>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
                else hipLaunchKernelGGL((k_train_gen2<128, 2>), dim3(groups), dim3(256), lds2, s, a, fw, bw);
            }
        } else if (kw_ == 32) {
            if (small) hipLaunchKernelGGL((k_train_gen<32, 4>), dim3(blocks), dim3(256), lds, s, a, fw, bw);
            else hipLaunchKernelGGL((k_train_gen<32, 8>), dim3(blocks), dim3(512), lds, s, a, fw, bw);
        } else if (kw_ == 64) {
            if (small) hipLaunchKernelGGL((k_train_gen<64, 4>), dim3(blocks), dim3(256), lds, s, a, fw, bw);
            else hipLaunchKernelGGL((k_train_gen<64, 8>), dim3(blocks), dim3(512), lds, s, a, fw, bw);
        } else {
            if (small) hipLaunchKernelGGL((k_train_gen<128, 4>), dim3(blocks), dim3(256), lds, s, a, fw, bw);
            else hipLaunchKernelGGL((k_train_gen<128, 8>), dim3(blocks), dim3(512), lds, s, a, fw, bw);
        }
        if (hash_) {
            HashLevels lv;
            for (uint32_t l = 0; l <= HG_LEVELS; l++) lv.off[l] = hg_off_[l];
            {
                // bin lists + exact LDS sums for the large levels, the fixed-point shadow for the rest (k_grid_scatter / k_grid_gather)
                GridBins gb;
                LooseRanges loose;
                loose.n = 0;
                uint32_t loose_chunks = 0;
                for (uint32_t l = 0; l < HG_LEVELS; l++) {
                    gb.first[l] = grid_bin_first_[l]; gb.count[l] = grid_bin_count_[l]; gb.cap[l] = grid_bin_cap_[l]; gb.list0[l] = grid_list0_[l];
                    if (grid_bin_count_[l] == 0u) {
                        loose.first[loose.n] = hg_off_[l];
                        loose.count[loose.n] = hg_off_[l + 1] - hg_off_[l];
                        loose_chunks += ceil_div(loose.count[loose.n], GB_BIN);
                        loose.n++;
                    }
                }
                for (uint32_t r = loose.n; r < HG_LEVELS; r++) loose.first[r] = loose.count[r] = 0u;
                if (!attr_gather_set_) {
                    NRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_gather), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GB_BIN * 16u)));
                    attr_gather_set_ = true;
                }
                const bool level_major = !xcd8_;
                hipLaunchKernelGGL(k_grid_scatter, level_major ? dim3(ceil_div(n, 256), HG_LEVELS) : dim3(ceil_div(n, 256) * HG_LEVELS, 1), dim3(256), 0, s, d_in, (const half_t*)d_denc_, (long long*)d_grid_fix_, n, lv,
                                   gb, (uint32_t*)d_grid_counters_, (uint2*)d_grid_lists_);
                hipLaunchKernelGGL(k_grid_gather, dim3(grid_bins_total_ + loose_chunks), dim3(NRC_GB_GATHER_THREADS), GB_BIN * 16u, s, (uint32_t*)d_grad16_,
                                   (const uint4*)d_grid_bin_entry0_, grid_bins_total_, (uint32_t*)d_grid_counters_, (const uint2*)d_grid_lists_, (long long*)d_grid_fix_, loose);
            }
            // the fp32 copy in the gradient vector is for whoever reads the vector (exchange, hook, debug read-back): the
            // optimizer takes the table gradient from grad16 itself (k_grid_opt)
            grid16_valid_ = fused_opt_;
            if (widen_grid_grad || !fused_opt_)
                hipLaunchKernelGGL(k_grid_grad_f32, dim3(ceil_div(n_grid_entries_, 256)), dim3(256), 0, s, (const uint32_t*)d_grad16_,
                                   d_grad_ + n_mlp_, n_grid_entries_);
        }
    }
    NRC_HIP(hipGetLastError());
    const uint32_t chunk = wgrad_chunk(n);
    const uint32_t n_chunks = ceil_div(n, chunk);
    if (wgrad_old_)
        hipLaunchKernelGGL(k_wgrad, dim3(n_chunks), dim3(WGRAD_WAVES * 64), 0, s, (const half_t*)d_deltas_, (const half_t*)d_acts_, n,
                           depth_ * kw_ + 8, enc_dims_ + depth_ * kw_, (const WgradTile*)d_tiles_, n_wgrad_tiles_, d_slabs_,
                           n_mlp_);
    else
        hipLaunchKernelGGL(k_wgrad2, dim3(n_chunks, ceil_div((uint32_t)n_wgrad_tasks_, (uint32_t)WGRAD2_WAVES)), dim3(WGRAD2_WAVES * 64), 0, s,
                           (const half_t*)d_deltas_, (const half_t*)d_acts_, n, chunk, depth_ * kw_ + 8, enc_dims_ + depth_ * kw_,
                           (const WgradTask*)d_tasks_, n_wgrad_tasks_, d_slabs_, n_mlp_);
    NRC_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_reduce_grads, dim3(ceil_div(n_mlp_, 64)), dim3(256), 0, s, d_slabs_, n_chunks, n_mlp_,
                       d_grad_, d_loss_part_, n_tiles, d_loss_);
    NRC_HIP(hipGetLastError());
}

uint32_t Mlp::grid_list_capacity(uint32_t n) const
{
    const unsigned long long touched = (unsigned long long)n * HG_LEVELS * 8ull;      // one entry per (sample, level, corner) at most
    return (uint32_t)std::min<unsigned long long>(touched, n_grid_entries_);
}

// the table gradient of the last backward() as a list (see k_grid_pack); d_list: grid_list_words(cap) words
void Mlp::grid_grad_pack(uint32_t* d_list, uint32_t cap, hipStream_t s)
{
    if (!hash_) fail("grid_grad_pack: this model has no trainable encoding");
    NRC_HIP(hipMemsetAsync(d_list, 0xff, grid_list_words(cap) * 4, s));        // entry 0xffffffff: padding
    NRC_HIP(hipMemsetAsync(d_list, 0, 8, s));
    hipLaunchKernelGGL(k_grid_pack, dim3(ceil_div(n_grid_entries_, 4096)), dim3(256), 0, s, (const uint32_t*)d_grad16_,
                       n_grid_entries_, d_list, cap);
    NRC_HIP(hipGetLastError());
}

// the gradient vector's table part := sum of n_lists lists (grid_list_words(cap) words apart), added in list order
void Mlp::grid_grad_apply(const uint32_t* d_lists, uint32_t n_lists, uint32_t cap, hipStream_t s)
{
    if (!hash_) fail("grid_grad_apply: this model has no trainable encoding");
    grid16_valid_ = false;      // the optimizer reads the summed fp32 table gradient
    NRC_HIP(hipMemsetAsync(d_grad_ + n_mlp_, 0, (size_t)n_grid_entries_ * 8, s));
    for (uint32_t r = 0; r < n_lists; r++)
        hipLaunchKernelGGL(k_grid_apply, dim3(ceil_div(cap, 256)), dim3(256), 0, s, d_lists + (size_t)r * grid_list_words(cap), cap,
                           d_grad_ + n_mlp_, n_grid_entries_);
    NRC_HIP(hipGetLastError());
}

bool Mlp::optimizer_step(hipStream_t s, uint32_t loss_seq, unsigned long long* loss_cell)
{
    step += 1;
    const double b1 = 0.9, b2 = 0.999;
    const double t = (double)step;
    const double d = (double)cfg_.ema_decay;
    AdamArgs a;
    a.lr_t = cfg_.learning_rate * (float)(std::sqrt(1.0 - std::pow(b2, t)) / (1.0 - std::pow(b1, t)));
    a.inv_loss_scale = 1.0f / kLossScale;
    a.ema_old = (float)(d * (1.0 - std::pow(d, t - 1.0)));
    a.ema_new = (float)(1.0 - d);
    a.ema_div = (float)(1.0 - std::pow(d, t));
    if (fused_opt_) {
        const int next = infer_set_ ^ 1;
        const PackDst d{d_dst_, d_dst_ + n_mlp_, d_dst_ + 2 * (size_t)n_mlp_, (half_t*)d_pk_fwd_, (half_t*)d_pk_infer_[next], (half_t*)d_pk_bwd_};
        // (the step's last launch takes the armed event along -- launch_last, nrc_common.hpp: k_opt_pack, or the table's optimizer behind it)
        if (hash_) {
            if (sgd_)
                hipLaunchKernelGGL(k_opt_pack<true>, dim3(ceil_div(n_mlp_, 256)), dim3(256), 0, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, n_mlp_,
                                   cfg_.learning_rate, a, d, (const float*)d_loss_, loss_seq, loss_cell);
            else
                hipLaunchKernelGGL(k_opt_pack<false>, dim3(ceil_div(n_mlp_, 256)), dim3(256), 0, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, n_mlp_,
                                   cfg_.learning_rate, a, d, (const float*)d_loss_, loss_seq, loss_cell);
        } else if (sgd_)
            launch_last(k_opt_pack<true>, dim3(ceil_div(n_mlp_, 256)), dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, n_mlp_,
                        cfg_.learning_rate, a, d, (const float*)d_loss_, loss_seq, loss_cell);
        else
            launch_last(k_opt_pack<false>, dim3(ceil_div(n_mlp_, 256)), dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, n_mlp_,
                        cfg_.learning_rate, a, d, (const float*)d_loss_, loss_seq, loss_cell);
        if (hash_) {
            const dim3 g(ceil_div(n_grid_entries_, 256));
            uint32_t *tt = (uint32_t*)d_t16_train_, *te = (uint32_t*)d_t16_ema_[next];
            uint32_t* g16 = (uint32_t*)d_grad16_;
            if (grid16_valid_) grad16_clean_ = true;      // k_grid_opt<., true> clears the entries it reads
            if (n_mlp_ % 4u == 0u && n_grid_entries_ % 2u == 0u) {      // (two entries per thread, 16-byte accesses; else round 3's kernel)
                const uint32_t np = n_grid_entries_ / 2u;
                const dim3 g2(ceil_div(np, 256));
                if (sgd_ && grid16_valid_)
                    launch_last(k_grid_opt2<true, true>, g2, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, np, cfg_.learning_rate, a, tt, te);
                else if (sgd_)
                    launch_last(k_grid_opt2<true, false>, g2, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, np, cfg_.learning_rate, a, tt, te);
                else if (grid16_valid_)
                    launch_last(k_grid_opt2<false, true>, g2, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, np, cfg_.learning_rate, a, tt, te);
                else
                    launch_last(k_grid_opt2<false, false>, g2, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, np, cfg_.learning_rate, a, tt, te);
            } else
            if (sgd_ && grid16_valid_)
                launch_last(k_grid_opt<true, true>, g, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, n_grid_entries_, cfg_.learning_rate, a, tt, te);
            else if (sgd_)
                launch_last(k_grid_opt<true, false>, g, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, n_grid_entries_, cfg_.learning_rate, a, tt, te);
            else if (grid16_valid_)
                launch_last(k_grid_opt<false, true>, g, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, n_grid_entries_, cfg_.learning_rate, a, tt, te);
            else
                launch_last(k_grid_opt<false, false>, g, dim3(256), 0u, s, d_w_, d_ema_, d_m_, d_v_, d_grad_, g16, n_mlp_, n_grid_entries_, cfg_.learning_rate, a, tt, te);
        }
        NRC_HIP(hipGetLastError());
        infer_set_ = next;
        return loss_cell != nullptr;
    }
    if (sgd_)
        hipLaunchKernelGGL(k_sgd_ema, dim3(ceil_div(n_params_, 256)), dim3(256), 0, s, d_w_, d_ema_, d_grad_, n_params_,
                           cfg_.learning_rate, a);
    else
        hipLaunchKernelGGL(k_adam_ema, dim3(ceil_div(n_params_, 256)), dim3(256), 0, s, d_w_, d_ema_, d_m_, d_v_, d_grad_,
                           n_params_, n_mlp_, a);
    NRC_HIP(hipGetLastError());
    repack(s);
    return false;
}

}  // namespace nrc
