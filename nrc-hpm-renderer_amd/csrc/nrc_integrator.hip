// nrc_integrator.hip -- headless HIP path integrator for gfx950: one lane per pixel path, density grid as R8 in HBM
// (served from L2 / Infinity Cache), HDR framebuffer out.  Compiled with -ffp-contract=off and with every fused
// multiply-add written explicitly (nrc_fmaf_): together with nrc_math.h this makes every per-pixel branch decision
// reproducible against the CPU oracle.
//
// Restates (paths relative to the reference checkout):
//   data/shader/include/random.glsl      hash RNG                         -> Rng
//   data/shader/include/volume.glsl      box SDF, entry/exit, density     -> sky_sdf, find_entry_exit, get_density
//   data/shader/include/dir_gen.glsl     HG phase + direction sampling    -> hg_phase, new_ray_dir
//   data/shader/include/path_trace.glsl  ratio/delta tracking, lights     -> ratio_track, delta_track, trace_scene
//   data/shader/nrc/gen_rays.comp + prep_infer_rays.comp (fused)          -> k_gen_rays
//   data/shader/mc/render.comp                                            -> k_mc_render
//   data/shader/nrc/clear.comp + prep_train_rays.comp                     -> k_train_scan + k_prep_train
//   data/shader/nrc/render.comp                                           -> k_composite
//   data/shader/ref/{cmp1,norm,cmp2}.comp                                 -> k_compare_*
#include "nrc_integrator.hpp"

#include <cstdlib>
#include <string>

#include "nrc_math.h"

// 1: the tracking loops carry their predicates as uniform 64-bit lane masks (round 4); 0: as per-lane bools (round 3)
// 1: every look-up tests the LDS occupancy bits unconditionally; 0 (product): behind a wave-uniform test for the table, as in round 3.
// Round 4 measured both: without the test k_gen_rays is 1 % faster alone (0.2092 against 0.2117 ms: one branch and four mask merges
// fewer per trip) and 12 % faster inside the configs[4] frame (0.283 against 0.318 ms) -- and the FRAMES are slower, 7 730 against 7 850
// Msamples/s on the default preset and 4 160 against 4 810 on configs[4]: a camera kernel that issues more densely leaves the
// inference / training kernels beside it fewer issue slots (train stage 0.18 -> 0.23 ms, configs[4] train rays 0.53 -> 0.98 ms), and
// those set the frame rate wherever they are the longer chain.  (tools/ab_config.sh, same box, lib / lib_o0 / lib_m0o0 / round 3's.)
#ifndef NRC_OCC_ALWAYS
#define NRC_OCC_ALWAYS 0
#endif
// 1 (product): software-pipelined, predicated tracking loops; 0 (diagnostic A/B build): the plain two-collision loops
// waves per SIMD the camera kernels are register-allocated for
#ifndef NRC_CAMERA_WAVES_PER_SIMD
#define NRC_CAMERA_WAVES_PER_SIMD 5
#endif

namespace nrc {
namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return nrc_fmaf_(a.z, b.z, nrc_fmaf_(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a)
{
    float inv = 1.0f / length(a);
    return v3(a.x * inv, a.y * inv, a.z * inv);
}
// b ? x : y per component (a ternary on the struct makes hipcc select between two stack copies)
__device__ __forceinline__ V3 sel(bool b, V3 x, V3 y) { return v3(b ? x.x : y.x, b ? x.y : y.y, b ? x.z : y.z); }
__device__ __forceinline__ V3 madd(V3 d, float t, V3 o) { return v3(nrc_fmaf_(d.x, t, o.x), nrc_fmaf_(d.y, t, o.y), nrc_fmaf_(d.z, t, o.z)); }   // o + d*t

// ---- include/random.glsl:24-70
__device__ __forceinline__ uint32_t hash1(uint32_t x)
{
    x += (x << 10);
    x ^= (x >> 6);
    x += (x << 3);
    x ^= (x >> 11);
    x += (x << 15);
    return x;
}
__device__ __forceinline__ float float_construct(uint32_t m) { return nrc_u2f((m & 0x007fffffu) | 0x3f800000u) - 1.0f; }
__device__ __forceinline__ float random1(float x) { return float_construct(hash1(nrc_f2u(x))); }
__device__ __forceinline__ float random2(float x, float y) { return float_construct(hash1(nrc_f2u(x) ^ hash1(nrc_f2u(y)))); }
__device__ __forceinline__ float random4(const float* v)
{
    return float_construct(hash1(nrc_f2u(v[0]) ^ hash1(nrc_f2u(v[1])) ^ hash1(nrc_f2u(v[2])) ^ hash1(nrc_f2u(v[3]))));
}

// two fp32 lanes per VGPR pair: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 perform the same IEEE operation on each half,
// so the packed forms below are bit-identical to their scalar statements (nrc_math.h) at half the VALU issue cost
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// ---- the free-flight log (nrc_math.h: nrc_logf).  Its 128-bin table lives in LDS: one instance per kernel that uses it (a file-scope
// __shared__ object), filled by every wave for itself -- identical values to identical addresses, no workgroup barrier, like
// load_occupancy_per_wave -- from the constant copy in device memory.
__device__ const NrcLogBin g_log_tab[128] = {
#include "nrc_log_table.inc"
};
__shared__ NrcLogBin s_log_tab[128];
__device__ __forceinline__ void fill_log_table()
{
    const uint4* src = reinterpret_cast<const uint4*>(g_log_tab);
    uint4* dst = reinterpret_cast<uint4*>(s_log_tab);
    dst[threadIdx.x & 63u] = src[threadIdx.x & 63u];      // 64 lanes x 16 bytes = the 1 KB table
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// nrc_logf on two arguments: the table reads and the integer part per argument, the arithmetic packed (bit-identical, see above)
__device__ __forceinline__ f2 logf2(f2 x)
{
    const uint32_t i0 = nrc_f2u(x.x), i1 = nrc_f2u(x.y);
    const f2 fe = f2{(float)__builtin_amdgcn_frexp_expf(x.x), (float)__builtin_amdgcn_frexp_expf(x.y)};     // == ((i >> 23) & 255) - 126 for normal x
    const f2 m = f2{__builtin_amdgcn_frexp_mantf(x.x), __builtin_amdgcn_frexp_mantf(x.y)};                 // == (i & 0x7fffff) | 0x3f000000
    const f2 t0 = *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(s_log_tab) + ((i0 >> 13) & 0x3f8u));      // {inv_c, log_c}
    const f2 t1 = *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(s_log_tab) + ((i1 >> 13) & 0x3f8u));
    // (the two table entries arrive in two register pairs: the steps that take them are scalar, or each would cost a move to re-pair)
    const f2 r = f2{nrc_fmaf_(m.x, t0.x, -1.0f), nrc_fmaf_(m.y, t1.x, -1.0f)};
    f2 q = fma2(r, splat(NRC_THIRD), splat(-0.5f));
    q = fma2(q, r, splat(1.0f));
    const f2 s = f2{nrc_fmaf_(fe.x, NRC_LN2, t0.y), nrc_fmaf_(fe.y, NRC_LN2, t1.y)};
    return fma2(q, r, s);
}

// correctly rounded sqrt for x == 0 or normal x: v_sqrt_f32 (1 ulp) + the one-ulp fix-up hipcc's own sqrtf lowering uses,
// without its denormal pre-scaling and class test.  Only sky_sdf calls it: its argument is a sum of squares of
// |p| - half_size terms, each 0 or >= half an ulp of a scene-sized coordinate, never denormal.
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    float y = __builtin_amdgcn_sqrtf(x);
    const float ym = nrc_u2f(nrc_f2u(y) - 1u), yp = nrc_u2f(nrc_f2u(y) + 1u);
    const float rm = nrc_fmaf_(-ym, y, x), rp = nrc_fmaf_(-yp, y, x);
    y = (rm <= 0.0f) ? ym : y;
    y = (rp > 0.0f) ? yp : y;
    return y;
}

// tools/loop_profile.py builds a second library with -DNRC_LOOP_PROFILE: per loop kind, iterations summed over lanes
// ("useful") and 64 x iterations issued by the wave ("issued"); never defined in the product build
#ifdef NRC_LOOP_PROFILE
// wave-level trips of the tracking loops by the number of walks alive: [0] ratio, 64-lane trips; [1] ratio, pair trips; [2] delta;
// classes 1, 2, 3-4, 5-8, 9-16, 17-32, 33-64
__device__ unsigned long long g_live_hist[3][8];
__device__ unsigned long long g_loop_prof[16];
__device__ unsigned long long g_wave_times[4 * 65536];
#endif
// -DNRC_NO_LOOP_COUNTERS keeps only the per-wave time stamps (the counters slow the kernel several-fold)
#if defined(NRC_LOOP_PROFILE) && !defined(NRC_NO_LOOP_COUNTERS)
#define NRC_PROF(c, k)                                                                    \
    do {                                                                                  \
        (c).useful[k]++;                                                                  \
        const unsigned long long m_ = __ballot(1);                                        \
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1) {                     \
            (c).issued[k] += 64u;                                                         \
            if ((k) == 2 || (k) == 3) {      /* tracking loops: trips issued with <= 32 / <= 16 lanes active (kinds 6, 7) */ \
                if (__popcll(m_) <= 32) (c).issued[6] += 64u;                             \
                if (__popcll(m_) <= 16) (c).issued[7] += 64u;                             \
            }                                                                             \
        }                                                                                 \
    } while (0)
#define NRC_PROF_LIVE(which, mask)                                                           \
    do {                                                                                     \
        const int n_ = __popcll(mask);                                                       \
        if (n_ > 0 && (threadIdx.x & 63u) == 0u) {                                           \
            const int cls_ = n_ <= 1 ? 0 : n_ <= 2 ? 1 : n_ <= 4 ? 2 : n_ <= 8 ? 3 : n_ <= 16 ? 4 : n_ <= 32 ? 5 : 6; \
            atomicAdd(&g_live_hist[which][cls_], 1ull);                                      \
        }                                                                                    \
    } while (0)
#else
#define NRC_PROF(c, k) do { } while (0)
#define NRC_PROF_LIVE(which, mask) do { } while (0)
#endif

// COUNT: the density look-ups are counted (nrc_renderer_count_fetches: a measurement launch); in the product launch the counter,
// its register and its adds do not exist
template <bool COUNT>
struct CtxT {
    const DevScene& sc;
    float rng;              // randomState
    uint32_t fetches;
    // raw buffer view of the volume: offsets >= the voxel count read 0, which is the sampler's black border
    __amdgpu_buffer_rsrc_t vol = __builtin_amdgcn_make_buffer_rsrc((void*)sc.density, 0, (int)(sc.nx * sc.ny * sc.nz), 0x00020000);
    const uint32_t* occ = nullptr;      // LDS copy of DevScene::occ_bits (load_occupancy), or nullptr
#ifdef NRC_LOOP_PROFILE
    uint32_t useful[8] = {0, 0, 0, 0, 0, 0, 0, 0}, issued[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t fee_kind = 1;
#endif
#ifdef NRC_DIAG_CUT_TAIL2
    uint32_t cut2 = 0;      // DIAGNOSTIC (wrong frames): the second vertex's delta walk stops when at most this many lanes still walk
#endif
    __device__ __forceinline__ float rand(float max_val)
    {
        rng = random1(rng);
        return rng * max_val;
    }
    __device__ __forceinline__ void count(uint32_t n)
    {
        if constexpr (COUNT) fetches += n;
    }
};

// The integrator is bound by the rate at which a CU gets scattered bytes out of the L2 (tools/gather_rate.hip: 2.3 cycles per
// distinct line per CU from L2, 7.6 from the Infinity Cache, against 0.1 from LDS), not by arithmetic.  Most look-ups of a
// shadow or sky ray fall into empty space around the cloud, where the answer is known to be 0: every kernel that samples the
// volume keeps the scene's occupancy bits (<= 8 KB) in LDS, and a look-up whose cell holds no non-zero voxel is answered from
// there -- it goes to the raw buffer with the out-of-range offset, which returns 0 without a memory access.  Exact by
// construction: the bit says that the byte the gather would have fetched is 0.
__device__ __forceinline__ const uint32_t* load_occupancy(const DevScene& sc, uint32_t* s_occ)
{
    fill_log_table();      // (every kernel that walks the volume draws free flights; the __syncthreads below covers it)
    if (sc.occ_bits == nullptr) return nullptr;
    for (uint32_t i = threadIdx.x; i < sc.occ_words; i += blockDim.x) s_occ[i] = sc.occ_bits[i];
    __syncthreads();
    return s_occ;
}
// The same for a kernel whose waves leave the workgroup at different times (k_gen_rays: a wave whose tile misses the medium is
// gone before this point): every wave copies the whole table for itself -- identical values to identical addresses -- and there is
// no workgroup barrier.  occ_words is a multiple of 4 (Scene::build_occupancy_bits).
__device__ __forceinline__ const uint32_t* load_occupancy_per_wave(const DevScene& sc, uint32_t* s_occ)
{
    fill_log_table();
    if (sc.occ_bits == nullptr) return nullptr;
    const uint4* src = reinterpret_cast<const uint4*>(sc.occ_bits);
    uint4* dst = reinterpret_cast<uint4*>(s_occ);
    for (uint32_t i = threadIdx.x & 63u; i < (sc.occ_words >> 2); i += 64u) dst[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return s_occ;
}

// wave-uniform 32-bit loads through the scalar cache (the compiler takes the vector path for a pointer it cannot prove unwritten)
__device__ __forceinline__ uint32_t scalar_load(const uint32_t* p)
{
    uint32_t r;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ void scalar_load2(const uint32_t* p, const uint32_t* q, uint32_t* a, uint32_t* b)
{
    uint32_t r0, r1;
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(p), "s"(q) : "memory");
    *a = r0;
    *b = r1;
}

template <class C>
__device__ __forceinline__ void init_random(C& c, float u, float v, const float* frame_random)
{
    c.rng = random2(random2(u, v), random4(frame_random));
}

// ---- include/volume.glsl
__device__ __forceinline__ float sky_sdf(const DevScene& s, V3 p)
{
    V3 d = v3(fabsf(p.x) - s.half_size[0], fabsf(p.y) - s.half_size[1], fabsf(p.z) - s.half_size[2]);
    V3 dm = v3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
    return sqrt_rn_normal(dot(dm, dm)) + fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f);
}

// sky_sdf inside the sphere-trace loops.  When at most one of the three per-axis distances is positive (the point is inside
// the box or in front of one face: med3 <= 0) the box distance is max3 itself -- length((0,0,x)) = sqrt(fl(x*x)) = x under
// correct rounding, and 0 + min(max3,0) = max3 -- so the square root is skipped whenever every active lane of the wave is
// in that case (4 of 5 iterations on the bench view); "+ 0.0f" keeps the -0 -> +0 of the general formula.
__device__ __forceinline__ float sky_sdf_step(const DevScene& s, V3 p)
{
    const float dx = fabsf(p.x) - s.half_size[0], dy = fabsf(p.y) - s.half_size[1], dz = fabsf(p.z) - s.half_size[2];
    const float m3 = fmaxf(dx, fmaxf(dy, dz));
    const bool general = !(__builtin_amdgcn_fmed3f(dx, dy, dz) <= 0.0f);
    if (__ballot(general) == 0ull) return m3 + 0.0f;
    const V3 dm = v3(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), fmaxf(dz, 0.0f));
    return sqrt_rn_normal(dot(dm, dm)) + fminf(m3, 0.0f);
}

template <class C>
__device__ __forceinline__ void find_entry_exit(C& c, V3 ro, V3 rd, V3* entry, V3* exit_)
{
    const DevScene& s = c.sc;
    float dist;
    do {
        NRC_PROF(c, c.fee_kind);
        dist = sky_sdf_step(s, ro);
        ro = madd(rd, dist, ro);
    } while (dist > 0.125f && dist < 100000.0f);
    *entry = ro;
    ro = madd(rd, s.len2size, ro);
    rd = neg(rd);
    do {
        NRC_PROF(c, c.fee_kind);
        dist = sky_sdf_step(s, ro);
        ro = madd(rd, dist, ro);
    } while (dist > 0.125f && dist < 100000.0f);
    *exit_ = ro;
}

// (z * ny + y) * nx + x on the full-rate 24-bit multiplier (hipcc turns __umul24 of an unbounded operand back into a quarter-rate
// 32-bit multiply); exact when every factor is below 2^24 and the result fits 32 bits -- guaranteed by the size checks at scene
// upload.  The two multiply-adds of an index are ONE asm statement that opens with two wait states: ny and nx are scalar operands,
// the compiler may have just reloaded them from an SGPR spill lane with v_readlane_b32, and a vector instruction that reads an SGPR
// another vector instruction wrote needs two wait states on this target -- which the compiler's hazard recognizer inserts for its
// own instructions but cannot see inside an asm string.
__device__ __forceinline__ uint32_t index24(uint32_t z, uint32_t ny, uint32_t y, uint32_t nx, uint32_t x)
{
    uint32_t r;
    asm("s_nop 1\n\tv_mad_u32_u24 %0, %1, %2, %3\n\tv_mad_u32_u24 %0, %0, %4, %5" : "=&v"(r) : "v"(z), "s"(ny), "v"(y), "s"(nx), "v"(x));
    return r;
}

// volume.glsl:31-39; sampler: R8 UNORM, NEAREST, CLAMP_TO_BORDER black (src/Texture3D.cpp:79-81,221).
// One byte per voxel, fetched with a 32-bit voxel index (24-bit mads); the per-axis border test is branch-free and
// out-of-range lanes read voxel 0 and discard it.
template <class C>
__device__ __forceinline__ float get_density(C& c, V3 p)
{
    const DevScene& s = c.sc;
    float u = nrc_fmaf_(p.x, s.inv_size[0], 0.5f);
    float v = nrc_fmaf_(p.y, s.inv_size[1], 0.5f);
    float w = nrc_fmaf_(p.z, s.inv_size[2], 0.5f);
    float fx = u * s.fnx, fy = v * s.fny, fz = w * s.fnz;
    c.count(1u);
    const bool inb = (fx >= 0.0f) & (fx < s.fnx) & (fy >= 0.0f) & (fy < s.fny) & (fz >= 0.0f) & (fz < s.fnz);
    const uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy, iz = (uint32_t)fz;
    uint32_t idx = index24(iz, s.ny, iy, s.nx, ix);
    idx = inb ? idx : 0u;
    const uint8_t t = s.density[idx];
    const float d = s.density_factor * ((float)t * (1.0f / 255.0f));
    return inb ? d : 0.0f;
}

// Two density look-ups, at start + dir*t1 and start + dir*t2, split into address, load and use so that the tracking loops can
// locate the NEXT pair of collisions while the gathers of the current pair are in flight (software pipelining: the free-flight
// chain does not depend on the fetched densities).  fetch2_addr computes both voxel indices (slots that are masked off or
// outside the volume get offset 2^31: the raw buffer returns the sampler's black border without a memory access), fetch2_load
// issues the two 1-byte gathers, fetch2_density turns the bytes into getDensity()'s value, volume.glsl:31-39.
struct Addr2 {
    uint32_t i0, i1;
};
struct Fetch2 {
    uint32_t b0, b1;
};
template <class C>
__device__ __forceinline__ Addr2 fetch2_addr(const C& c, V3 dir, V3 start, float t1, float t2, bool first, bool second)
{
    const DevScene& s = c.sc;
    const f2 t = f2{t1, t2};
    const f2 px = fma2(splat(dir.x), t, splat(start.x));
    const f2 py = fma2(splat(dir.y), t, splat(start.y));
    const f2 pz = fma2(splat(dir.z), t, splat(start.z));
    const f2 u = fma2(px, splat(s.inv_size[0]), splat(0.5f));
    const f2 v = fma2(py, splat(s.inv_size[1]), splat(0.5f));
    const f2 w = fma2(pz, splat(s.inv_size[2]), splat(0.5f));
    const f2 fx = u * splat(s.fnx), fy = v * splat(s.fny), fz = w * splat(s.fnz);
    // 0 <= f < n  <=>  0 <= u < 1 (n >= 1; u*n never rounds up to n; u is never -0)  <=>  bits(u) < bits(1.0f)
    bool in0 = (max(max(nrc_f2u(u.x), nrc_f2u(v.x)), nrc_f2u(w.x)) < 0x3f800000u) & first;
    bool in1 = (max(max(nrc_f2u(u.y), nrc_f2u(v.y)), nrc_f2u(w.y)) < 0x3f800000u) & second;
    const uint32_t x0 = (uint32_t)fx.x, y0 = (uint32_t)fy.x, z0 = (uint32_t)fz.x;
    const uint32_t x1 = (uint32_t)fx.y, y1 = (uint32_t)fy.y, z1 = (uint32_t)fz.y;
    const uint32_t idx0 = index24(z0, s.ny, y0, s.nx, x0);
    const uint32_t idx1 = index24(z1, s.ny, y1, s.nx, x1);
#if !NRC_OCC_ALWAYS
    if (c.occ != nullptr)
#endif
    {      // occupancy bit of the voxel's cell from LDS: an empty cell's byte is 0 without asking memory.  (The table always exists --
           // Scene::build_occupancy_bits; all ones under NRC_NO_OCCUPANCY.)
        const uint32_t sh = s.occ_shift;
        uint32_t c0 = index24(z0 >> sh, s.occ_gy, y0 >> sh, s.occ_gx, x0 >> sh);
        uint32_t c1 = index24(z1 >> sh, s.occ_gy, y1 >> sh, s.occ_gx, x1 >> sh);
        c0 = in0 ? c0 : 0u;
        c1 = in1 ? c1 : 0u;
        in0 &= ((c.occ[c0 >> 5] >> (c0 & 31u)) & 1u) != 0u;
        in1 &= ((c.occ[c1 >> 5] >> (c1 & 31u)) & 1u) != 0u;
    }
    return Addr2{in0 ? idx0 : 0x80000000u, in1 ? idx1 : 0x80000000u};
}
template <class C>
__device__ __forceinline__ Fetch2 fetch2_load(C& c, const Addr2& a)
{
    Fetch2 f;
    f.b0 = __builtin_amdgcn_raw_buffer_load_b8(c.vol, (int)a.i0, 0, 0);
    f.b1 = __builtin_amdgcn_raw_buffer_load_b8(c.vol, (int)a.i1, 0, 0);
    return f;
}
__device__ __forceinline__ f2 fetch2_density(const DevScene& s, const Fetch2& f)
{
    return splat(s.density_factor) * (f2{(float)f.b0, (float)f.b1} * splat(1.0f / 255.0f));
}

// ---- include/dir_gen.glsl
__device__ __forceinline__ float hg_phase(const DevScene& s, float cos_theta)
{
    float g = s.g;
    float g2 = g * g;
    float x = nrc_fmaf_(-(2.0f * g), cos_theta, 1.0f + g2);
    return (0.5f * (1.0f - g2)) / (x * sqrtf(x));
}

template <bool SECOND = false>
__device__ __forceinline__ V3 rotate(V3 axis, float angle, V3 v)
{
    axis = normalize(axis);
    float s, co;
    nrc_sincosf(angle, &s, &co);
    float oc = 1.0f - co;
    const float ox = oc * axis.x, oy = oc * axis.y, oz = oc * axis.z;
    V3 c0 = v3(nrc_fmaf_(ox, axis.x, co), nrc_fmaf_(ox, axis.y, -(axis.z * s)), nrc_fmaf_(oz, axis.x, axis.y * s));
    V3 c1 = v3(nrc_fmaf_(ox, axis.y, axis.z * s), nrc_fmaf_(oy, axis.y, co), nrc_fmaf_(oy, axis.z, -(axis.x * s)));
    V3 c2 = v3(nrc_fmaf_(oz, axis.x, -(axis.y * s)), nrc_fmaf_(oy, axis.z, axis.x * s), nrc_fmaf_(oz, axis.z, co));
    const V3 r = v3(nrc_fmaf_(c2.x, v.z, nrc_fmaf_(c1.x, v.y, c0.x * v.x)),
                    nrc_fmaf_(c2.y, v.z, nrc_fmaf_(c1.y, v.y, c0.y * v.x)),
                    nrc_fmaf_(c2.z, v.z, nrc_fmaf_(c1.z, v.y, c0.z * v.x)));
    return r;
}

// (The diagnostic builds of DESIGN.md section 7.1 -- -DNRC_DIAG_LASTDIR, -DNRC_DIAG_BISECT=<mask>: code regions of new_ray_dir bracketed with
// s_setprio -- did their work in round 3 and left the product source in round 6; tools/stress_lastdir.sh, bisect_build.sh and
// asm_patch_experiment.sh name the commit that still builds them.)
template <class C>
__device__ __forceinline__ V3 new_ray_dir(C& c, V3 old_dir, bool phase_sampling)
{
    NRC_PROF(c, 4);
    old_dir = normalize(old_dir);
    V3 ortho = old_dir.z < old_dir.x ? v3(old_dir.y, -old_dir.x, 0.0f) : v3(0.0f, -old_dir.z, old_dir.y);
    if (ortho.x == 0.0f && ortho.y == 0.0f && ortho.z == 0.0f) ortho = v3(0.0f, 1.0f, 0.0f);   // robustness
    ortho = normalize(ortho);
    float angle;
    if (phase_sampling) {
        float g = c.sc.g;
        float cos_theta;
        if (fabsf(g) < 0.001f) {
            cos_theta = 1.0f - 2.0f * c.rand(1.0f);
        } else {
            float sqr_term = (1.0f - g * g) / nrc_fmaf_(2.0f * g, c.rand(1.0f), 1.0f - g);
            cos_theta = nrc_fmaf_(-sqr_term, sqr_term, 1.0f + g * g) / (2.0f * g);
        }
        angle = nrc_acosf_clamped(cos_theta);
    } else {
        angle = c.rand(NRC_PI);
    }
    V3 nd = rotate(ortho, angle, old_dir);
    angle = c.rand(NRC_TWO_PI);
    nd = rotate<true>(old_dir, angle, nd);
    nd = normalize(nd);
    return nd;
}

// ---- include/path_trace.glsl
// RatioTrack, path_trace.glsl:24-43.  Two collisions per trip: nothing in a step depends on the previous fetch, so both free-
// flight logs share packed math.  The loop is software-pipelined: a trip's two gathers are issued at the top of the loop body,
// the NEXT trip is located (hash chain, logs, positions, voxel indices -- about 80 instructions that do not depend on any
// density) while they are in flight, and only then are the densities used.  A lane that ends on a collision keeps the RNG state
// of that draw, exactly as the one-step loop would; the trip located ahead of it is dropped.
struct RatioTrip {
    float s1, s2, t1, t2;
    bool live1, second;       // collision 1 / collision 2 lie inside the segment
};
__device__ __forceinline__ RatioTrip ratio_trip(float rng, float t, float t_max, float inv)
{
    RatioTrip r;
    r.s1 = random1(rng);
    r.s2 = random1(r.s1);
    const f2 l = logf2(f2{1.0f - r.s1, 1.0f - r.s2});
    r.t1 = nrc_fmaf_(-l.x, inv, t);
    r.t2 = nrc_fmaf_(-l.y, inv, r.t1);
    r.live1 = !(r.t1 >= t_max);
    r.second = !(r.t2 >= t_max);
    return r;
}
// ---- thin trips: two lanes per surviving walk ---------------------------------------------------------------------------------
// A tracking loop runs until its last walk ends: the trips issued with at most 32 of the 64 lanes still walking cost 28 % of the
// launch (NRC_DIAG_CUT_TAIL).  When a loop is down to 32 walks or fewer they are handed to lane PAIRS (lane p and p + 32, state pushed
// there with ds_permute): an iteration then advances a walk by FOUR collisions -- the hash chain is drawn by both lanes (it is the
// only serial part, a quarter of a trip's instructions), the two free-flight logs, positions, look-ups and transmittance factors
// of collisions 1, 2 are lane p's work and those of 3, 4 lane p + 32's, exchanged with v_permlane32_swap, and the free-flight sums
// and transmittance products are then formed by both lanes in the sequential order -- bit for bit the walk of the one-lane loop.
// (value of the lower-half lane, value of the upper-half lane) of a pair, on both of its lanes
__device__ __forceinline__ void pair_both(float x, float* lo, float* hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(nrc_f2u(x), nrc_f2u(x), false, false);
    *lo = nrc_u2f(r[0]);
    *hi = nrc_u2f(r[1]);
}

// (Two further tails were built bit-exact, measured and removed -- docs/MEASUREMENT_LOG.md: `ratio_wide`, 32 lanes for each of a wave's last one
// or two ratio walks: the walks it applies to are too few and its registers cost the kernel 0.038 ms; `delta_pairs`, the pair tail for
// delta tracking: 144 B of scratch under the 96-register cap, slower at five waves per SIMD.  Their sources: git history, round 5's tree.)
// the surviving walks of a ratio_track loop, two lanes per walk.  On entry: `alive` lanes have a located trip whose base state is
// (bs, bt) = (chain value before its first draw, free-flight position before it), n collisions done.  On exit: tr / rng of the alive
// lanes are the finished walks' results.
template <class C>
__device__ __forceinline__ void ratio_pairs(C& c, unsigned long long am, bool alive, V3 start, V3 dir, float t_max, float inv, float bs,
                                            float bt, uint32_t n, float& tr, float& rng)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
    const uint32_t k = (uint32_t)__popcll(am);
    const bool is_b = (lane & 32u) != 0u;
    // walk state -> lanes rank and rank + 32 (rank = the lane's position among the surviving ones)
    float y[11];
    {
        const float x[11] = {bs, bt, t_max, tr, nrc_u2f(n), start.x, start.y, start.z, dir.x, dir.y, dir.z};
#pragma unroll
        for (int q = 0; q < 11; q++) {
            // every lane pushes (a lane that is masked off neither sends nor receives): the lanes without a walk push to lanes 31 and
            // 63, which are never part of a pair (at most 31 walks survive)
            const int lo = __builtin_amdgcn_ds_permute((int)((alive ? rank : 31u) * 4u), (int)nrc_f2u(x[q]));
            const int hi = __builtin_amdgcn_ds_permute((int)((alive ? rank + 32u : 63u) * 4u), (int)nrc_f2u(x[q]));
            y[q] = nrc_u2f((uint32_t)(is_b ? hi : lo));
        }
    }
    float R0 = y[0], t0 = y[1], T = y[3];
    const float tm = y[2];
    uint32_t nn = nrc_f2u(y[4]);
    const V3 st = v3(y[5], y[6], y[7]), dr = v3(y[8], y[9], y[10]);
    bool act = (lane & 31u) < k;
    float rf = R0;
    for (;;) {
        const unsigned long long actm = __ballot(act);
        if (actm == 0ull) break;
        NRC_PROF_LIVE(1, actm & 0xffffffffull);
        const float R1 = random1(R0), R2 = random1(R1), R3 = random1(R2), R4 = random1(R3);
        const f2 l = logf2(f2{1.0f - (is_b ? R3 : R1), 1.0f - (is_b ? R4 : R2)});
        float l1, l2, l3, l4;
        pair_both(l.x, &l1, &l3);
        pair_both(l.y, &l2, &l4);
        const float t1 = nrc_fmaf_(-l1, inv, t0), t2 = nrc_fmaf_(-l2, inv, t1), t3 = nrc_fmaf_(-l3, inv, t2), t4 = nrc_fmaf_(-l4, inv, t3);
        // the sequential walk: draw k is made unless 128 collisions are done; the walk ends on a draw that lies beyond the segment
        bool go = act;
        const bool d1 = go & (nn + 1u <= 128u); rf = d1 ? R1 : rf; const bool p1 = d1 & !(t1 >= tm); go = p1;
        const bool d2 = go & (nn + 2u <= 128u); rf = d2 ? R2 : rf; const bool p2 = d2 & !(t2 >= tm); go = p2;
        const bool d3 = go & (nn + 3u <= 128u); rf = d3 ? R3 : rf; const bool p3 = d3 & !(t3 >= tm); go = p3;
        const bool d4 = go & (nn + 4u <= 128u); rf = d4 ? R4 : rf; const bool p4 = d4 & !(t4 >= tm); go = p4;
        // look-ups: collisions 1, 2 on the lower lane, 3, 4 on the upper one
        const bool ma = is_b ? p3 : p1, mb = is_b ? p4 : p2;
        const Addr2 ad = fetch2_addr(c, dr, st, is_b ? t3 : t1, is_b ? t4 : t2, ma, mb);
        const Fetch2 fa = fetch2_load(c, ad);
        const f2 dens = fetch2_density(c.sc, fa);
        c.count((ma ? 1u : 0u) + (mb ? 1u : 0u));
        float f1, f2_, f3, f4;
        pair_both(nrc_fmaf_(-dens.x, inv, 1.0f), &f1, &f3);
        pair_both(nrc_fmaf_(-dens.y, inv, 1.0f), &f2_, &f4);
        T = p1 ? T * f1 : T;
        T = p2 ? T * f2_ : T;
        T = p3 ? T * f3 : T;
        T = p4 ? T * f4 : T;
        act = p4 & (nn + 4u < 128u);
        nn += 4u;
        R0 = R4;
        t0 = t4;
    }
    // results back to the walks' own lanes (from the lower lane of their pair)
    const float tr_new = nrc_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank * 4u), (int)nrc_f2u(T)));
    const float rng_new = nrc_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank * 4u), (int)nrc_f2u(rf)));
    tr = alive ? tr_new : tr;
    rng = alive ? rng_new : rng;
}

// The loop is wave-uniform (it runs while any lane still walks) and its body is predicated with selects instead of per-lane
// branches: with `break`s the compiler sinks the look-ahead into the continue path, i.e. behind the wait for the gathers, and
// the overlap is gone.  Lanes that have finished keep their results and issue no gathers (offset 2^31).
// UNI: the call sits in wave-uniform control flow (every lane of the wave executes it, `valid` says which lanes have a walk), so
// the lanes without a walk can help with the thin trips at the end (ratio_pairs)
// Round 4: the predicates the loop carries from trip to trip (who still walks, whether the located trip's collisions lie inside the
// segment) are 64-bit LANE MASKS in scalar registers, not per-lane bools.  A loop-carried bool is a divergent i1 phi, which the
// compiler merges with three scalar instructions per predicate, per loop header and per exit (s_andn2 / s_and / s_or with exec:
// 30 of them in front of every delta trip) -- and a scalar instruction costs a SIMD as much issue time as a vector one here
// (tools/issue_mix.hip: at five waves per SIMD a v_fma_f32 1.6 clocks, a v_fma_f32 + s_and_b64 pair 3.6).  A uniform mask is an
// ordinary 64-bit value: its phi is a copy.  lane_bool() hands it to v_cndmask as the select mask (no instruction).
__device__ __forceinline__ bool lane_bool(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ unsigned long long lane_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// Round 5 measured the next step of that idea and took it out again (commit 4f700df has the code; DESIGN.md section 4 "Lane groups"): every
// surviving walk on a GROUP of L = 2^floor(log2(64 / walks alive)) lanes, chunks of 2 Lc collisions per iteration, the walk's serial
// recurrences (hash chain, free-flight sums, transmittance product) as DPP wave_shr:1 scans along the group (tools/scan_probe.hip,
// tools/dpp_probe.hip).  Bit-exact at the first run of every oracle comparison -- and slower than the pair tail on every preset
// (k_gen_rays alone 0.2111 -> 0.2138 ... 0.2181 ms by entry threshold, frame - 5 %, configs[4] - 4.7 %, Monte-Carlo renderer - 1.5 %).
template <bool UNI = false, class C>
__device__ __forceinline__ float ratio_track(C& c, V3 start, V3 end, bool valid = true)
{
    V3 d = sub(end, start);
    const V3 dir = normalize(d);
    const float t_max = length(d);
    const float inv = c.sc.inv_max_density;
    float tr = 1.0f;
    float rng = c.rng;
    unsigned long long alive_m = lane_mask(valid);
    RatioTrip a = ratio_trip(rng, 0.0f, t_max, inv);
    unsigned long long live1_m = lane_mask(a.live1), second_m = lane_mask(a.second);
    Addr2 ia = fetch2_addr(c, dir, start, a.t1, a.t2, a.live1, a.live1 & a.second);
    float bs = rng, bt = 0.0f;                 // base of the located trip `a`: chain value and position before its first draw
    for (uint32_t i = 0;; i += 2) {            // i counts collisions: at most 128 (path_trace.glsl:34)
        rng = lane_bool(alive_m & ~live1_m) ? a.s1 : rng;             // collision 1 beyond the segment: the walk ends on this draw
        alive_m &= live1_m;
#ifdef NRC_DIAG_CUT_TAIL
        if (__popcll(alive_m) <= NRC_DIAG_CUT_TAIL) break;      // DIAGNOSTIC (wrong frames): what the trips with few live lanes cost
#else
        if (alive_m == 0ull) break;
#endif
        if constexpr (UNI) {
            if (__popcll(alive_m) <= 31) {        // few walks left: two lanes each (ratio_pairs; lanes 31 and 63 stay free as push targets)
                ratio_pairs(c, alive_m, lane_bool(alive_m), start, dir, t_max, inv, bs, bt, i, tr, rng);
                break;
            }
        }
        const bool alive = lane_bool(alive_m);
        if (alive) NRC_PROF(c, 3);
        NRC_PROF_LIVE(0, alive_m);
        const Fetch2 fa = fetch2_load(c, ia);                        // this trip's gathers ...
        __builtin_amdgcn_sched_barrier(0);
        const bool last = i + 2 >= 128;
        const unsigned long long two_m = alive_m & second_m;
        const unsigned long long more_m = last ? 0ull : two_m;
        const bool more = lane_bool(more_m);
        const RatioTrip b = ratio_trip(a.s2, a.t2, t_max, inv);      // ... fly while the next trip is located
        const Addr2 ib = fetch2_addr(c, dir, start, b.t1, b.t2, more & b.live1, more & b.live1 & b.second);
        __builtin_amdgcn_sched_barrier(0);
        const f2 dens = fetch2_density(c.sc, fa);
        c.count(alive ? (lane_bool(second_m) ? 2u : 1u) : 0u);
        tr = alive ? tr * nrc_fmaf_(-dens.x, inv, 1.0f) : tr;
        tr = lane_bool(two_m) ? tr * nrc_fmaf_(-dens.y, inv, 1.0f) : tr;
        rng = lane_bool(alive_m & ~more_m) ? a.s2 : rng;              // ends after collision 1 / after the 128th collision
        alive_m = more_m;
        live1_m = lane_mask(b.live1);
        second_m = lane_mask(b.second);
        bs = a.s2;
        bt = a.t2;
        a.s1 = b.s1; a.s2 = b.s2; a.t1 = b.t1; a.t2 = b.t2;
        ia = ib;
    }
    c.rng = rng;
    return tr;
}

template <bool UNI = false, class C>
__device__ __forceinline__ V3 trace_dir_light(C& c, V3 pos, V3 dir, bool valid = true)
{
    const DevScene& s = c.sc;
    if (s.dir_light_strength == 0.0f) return v3(0, 0, 0);
    V3 ld = v3(s.dir_light_dir[0], s.dir_light_dir[1], s.dir_light_dir[2]);
    V3 en, ex;
    if constexpr (UNI) pos = sel(valid, pos, v3(0.0f, 0.0f, 0.0f));
    find_entry_exit(c, pos, neg(normalize(ld)), &en, &ex);
    float tr = ratio_track<UNI>(c, pos, ex, valid);
    float phase = hg_phase(s, dot(ld, neg(dir)));
    float l = (1.0f * tr) * s.dir_light_strength * phase;
    return v3(l, l, l);
}

template <bool UNI = false, class C>
__device__ __forceinline__ V3 trace_point_light(C& c, V3 pos, V3 dir, bool valid = true)
{
    const DevScene& s = c.sc;
    if (s.point_light_strength == 0.0f) return v3(0, 0, 0);
    V3 lp = v3(s.point_light_pos[0], s.point_light_pos[1], s.point_light_pos[2]);
    float tr = ratio_track<UNI>(c, lp, pos, valid);
    float phase = hg_phase(s, dot(normalize(sub(lp, pos)), neg(dir)));
    return v3(((s.point_light_color[0] * s.point_light_strength) * tr) * phase,
              ((s.point_light_color[1] * s.point_light_strength) * tr) * phase,
              ((s.point_light_color[2] * s.point_light_strength) * tr) * phase);
}

__device__ __forceinline__ V3 env_lookup(const DevScene& s, float u, float v)
{
    if (s.env == nullptr || s.env_w == 0) return v3(0, 0, 0);
    float fx = u * (float)s.env_w - 0.5f, fy = v * (float)s.env_h - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float wx = fx - flx, wy = fy - fly;
    int x0 = (int)flx, y0 = (int)fly, x1 = x0 + 1, y1 = y0 + 1;
    int mw = (int)s.env_w - 1, mh = (int)s.env_h - 1;
    x0 = min(max(x0, 0), mw);
    x1 = min(max(x1, 0), mw);
    y0 = min(max(y0, 0), mh);
    y1 = min(max(y1, 0), mh);
    const float* p00 = s.env + 4 * ((size_t)y0 * s.env_w + x0);
    const float* p10 = s.env + 4 * ((size_t)y0 * s.env_w + x1);
    const float* p01 = s.env + 4 * ((size_t)y1 * s.env_w + x0);
    const float* p11 = s.env + 4 * ((size_t)y1 * s.env_w + x1);
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = nrc_fmaf_(wx, p10[k] - p00[k], p00[k]);
        float b = nrc_fmaf_(wx, p11[k] - p01[k], p01[k]);
        r[k] = nrc_fmaf_(wy, b - a, a) * s.env_strength;
    }
    return v3(r[0], r[1], r[2]);
}

__device__ __forceinline__ V3 sample_env_dir(const DevScene& s, V3 dir)
{
    float phi = nrc_atan2f(dir.z, dir.x);
    float theta = nrc_asinf(dir.y);
    return env_lookup(s, nrc_fmaf_(phi, 0.1591f, 0.5f), nrc_fmaf_(theta, 0.3183f, 0.5f));
}

template <bool UNI = false, class C>
__device__ __forceinline__ V3 sample_env(C& c, V3 pos, V3 dir, bool valid = true)
{
    if (c.sc.env_strength == 0.0f) return v3(0, 0, 0);
    V3 rdir = v3(0.0f, 0.0f, 1.0f);
    if constexpr (UNI) {
        if (valid) rdir = new_ray_dir(c, dir, false);      // lanes without a walk draw nothing
        pos = sel(valid, pos, v3(0.0f, 0.0f, 0.0f));
    } else {
        rdir = new_ray_dir(c, dir, false);
    }
    float phase = hg_phase(c.sc, dot(rdir, neg(dir)));
    V3 en, ex;
    find_entry_exit(c, pos, rdir, &en, &ex);
    float tr = ratio_track<UNI>(c, pos, ex, valid);
    V3 e = sample_env_dir(c.sc, rdir);
    return v3((e.x * phase) * tr, (e.y * phase) * tr, (e.z * phase) * tr);
}

template <bool UNI = false, class C>
__device__ __forceinline__ V3 trace_scene(C& c, V3 pos, V3 dir, bool valid = true)
{
    NRC_PROF(c, 5);
    V3 a = trace_dir_light<UNI>(c, pos, dir, valid);
    V3 b = trace_point_light<UNI>(c, pos, dir, valid);
    V3 e = sample_env<UNI>(c, pos, dir, valid);
    return add(add(a, b), e);
}

// DeltaTrack, path_trace.glsl:150-174, two collisions per trip, software-pipelined like ratio_track: the second collision's free
// flight and fetch are issued beside the first's and the NEXT trip's two collisions are located while the gathers are in flight
// (the RNG chain does not depend on the density); whichever event comes first in sequence order -- exit, accept 1, exit,
// accept 2 -- ends the walk with the RNG state the one-step loop would have, and the trip located ahead is dropped
struct DeltaTrip {
    float s1, a1, s2, a2, t1, t2;
    bool live1, second;
};
__device__ __forceinline__ DeltaTrip delta_trip(float rng, float t, float t_max, float inv)
{
    DeltaTrip r;
    r.s1 = random1(rng);
    r.a1 = random1(r.s1);
    r.s2 = random1(r.a1);
    r.a2 = random1(r.s2);
    const f2 l = logf2(f2{1.0f - r.s1, 1.0f - r.s2});
    r.t1 = nrc_fmaf_(-l.x, inv, t);
    r.t2 = nrc_fmaf_(-l.y, inv, r.t1);
    r.live1 = !(r.t1 >= t_max);
    r.second = !(r.t2 >= t_max);
    return r;
}
template <bool UNI = false, class C>
__device__ __forceinline__ V3 delta_track(C& c, V3 ro, V3 rd, bool* volume_exit, bool valid = true)
{
    V3 en, ex;
    if constexpr (UNI) {      // lanes without a walk march a harmless ray (from the centre along +z): the march must end for them too
        find_entry_exit(c, sel(valid, ro, v3(0.0f, 0.0f, 0.0f)), sel(valid, rd, v3(0.0f, 0.0f, 1.0f)), &en, &ex);
    } else {
        find_entry_exit(c, ro, rd, &en, &ex);
    }
    const float t_max = length(sub(ex, ro));
    const float inv = c.sc.inv_max_density;
    float rng = c.rng;
    // loop-carried predicates as uniform lane masks (see ratio_track)
    unsigned long long alive_m = lane_mask(valid), hit_m = 0ull, vexit_m = 0ull;
    float t_hit = 0.0f;
    DeltaTrip a = delta_trip(rng, 0.0f, t_max, inv);
    unsigned long long live1_m = lane_mask(a.live1), second_m = lane_mask(a.second);
    Addr2 ia = fetch2_addr(c, rd, ro, a.t1, a.t2, a.live1, a.live1 & a.second);
    float bs = rng, bt = 0.0f;                 // base of the located trip `a`
    for (uint32_t i = 0;; i += 2) {            // i counts collisions: at most 128 (path_trace.glsl:161); predicated like ratio_track
        const unsigned long long out1_m = alive_m & ~live1_m;        // collision 1 beyond the exit point
        rng = lane_bool(out1_m) ? a.s1 : rng;
        vexit_m |= out1_m;
        alive_m &= live1_m;
#ifdef NRC_DIAG_CUT_TAIL
        if (__popcll(alive_m) <= NRC_DIAG_CUT_TAIL) break;      // DIAGNOSTIC (wrong frames): what the trips with few live lanes cost
#else
        if (alive_m == 0ull) break;
#endif
#ifdef NRC_DIAG_CUT_TAIL2
        if ((uint32_t)__popcll(alive_m) <= c.cut2) break;      // DIAGNOSTIC (wrong frames): the same for the second vertex's delta walk alone
#endif
        const bool alive = lane_bool(alive_m);
        if (alive) NRC_PROF(c, 2);
        NRC_PROF_LIVE(2, alive_m);
        const Fetch2 fa = fetch2_load(c, ia);
        __builtin_amdgcn_sched_barrier(0);
        const bool last = i + 2 >= 128;
        const DeltaTrip b = delta_trip(a.a2, a.t2, t_max, inv);      // located ahead; used only if this trip accepts nothing
        const bool maybe = lane_bool(last ? 0ull : (alive_m & second_m));
        const Addr2 ib = fetch2_addr(c, rd, ro, b.t1, b.t2, maybe & b.live1, maybe & b.live1 & b.second);
        __builtin_amdgcn_sched_barrier(0);
        const f2 dens = fetch2_density(c.sc, fa) * splat(inv);
        c.count(alive ? 1u : 0u);
        const unsigned long long acc1_m = alive_m & lane_mask(dens.x > a.a1);
        const unsigned long long alive2_m = alive_m & ~acc1_m;
        const unsigned long long out2_m = alive2_m & ~second_m;      // collision 2 beyond the exit point
        const unsigned long long alive3_m = alive2_m & second_m;
        c.count(lane_bool(alive3_m) ? 1u : 0u);
        const unsigned long long acc2_m = alive3_m & lane_mask(dens.y > a.a2);
        hit_m |= acc1_m | acc2_m;
        t_hit = lane_bool(acc1_m) ? a.t1 : (lane_bool(acc2_m) ? a.t2 : t_hit);
        vexit_m |= out2_m;
        // RNG state of the event that ended the walk: accept 1 -> a1, exit 2 -> s2, accept 2 or the 128-collision cap -> a2
        rng = lane_bool(acc1_m) ? a.a1 : rng;
        rng = lane_bool(out2_m) ? a.s2 : rng;
        rng = lane_bool(last ? alive3_m : acc2_m) ? a.a2 : rng;
        alive_m = last ? 0ull : (alive3_m & ~acc2_m);
        live1_m = lane_mask(b.live1);
        second_m = lane_mask(b.second);
        bs = a.a2;
        bt = a.t2;
        a.s1 = b.s1; a.a1 = b.a1; a.s2 = b.s2; a.a2 = b.a2; a.t1 = b.t1; a.t2 = b.t2;
        const bool go = lane_bool(alive_m);
        ia.i0 = go ? ib.i0 : 0x80000000u;                            // a lane that has just finished fetches nothing next trip
        ia.i1 = go ? ib.i1 : 0x80000000u;
    }
    c.rng = rng;
    const bool hit = lane_bool(hit_m);
    *volume_exit = lane_bool(vexit_m);
    if (hit) return madd(rd, t_hit, ro);
    if constexpr (UNI) {
        if (!valid) return ro;          // no walk: no draw
    }
    return madd(rd, c.rand(t_max), ro);
}

// camera ray: mc/render.comp:42-60, nrc/gen_rays.comp:53-72 (no half-pixel offset, no y flip)
__device__ __forceinline__ void camera_ray(const DevCamera& cam, float u, float v, V3* ro, V3* rd)
{
    float sx = nrc_fmaf_(u, 2.0f, -1.0f), sy = nrc_fmaf_(v, 2.0f, -1.0f);
    const float* m = cam.m;
    float wx = nrc_fmaf_(m[4], sy, nrc_fmaf_(m[0], sx, m[12]));
    float wy = nrc_fmaf_(m[5], sy, nrc_fmaf_(m[1], sx, m[13]));
    float wz = nrc_fmaf_(m[6], sy, nrc_fmaf_(m[2], sx, m[14]));
    float ww = nrc_fmaf_(m[7], sy, nrc_fmaf_(m[3], sx, m[15]));
    V3 p = v3(wx / ww, wy / ww, wz / ww);
    *ro = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    *rd = normalize(sub(p, *ro));
}

// StoreNrcInferInput / StoreNrcTrainData normalisation (quirks Q3-Q5 kept)
__device__ __forceinline__ void nrc_query(const DevScene& s, V3 pos, V3 dir, float* q)
{
    q[0] = pos.x / s.size[0] + s.size[0] / 2.0f;
    q[1] = pos.y / s.size[1] + s.size[1] / 2.0f;
    q[2] = pos.z / s.size[2] + s.size[2] / 2.0f;
    float theta = nrc_atan2f(dir.z, dir.x);
    q[3] = theta / NRC_PI + 0.5f;
    float lxz = sqrtf(dir.x * dir.x + dir.z * dir.z);
    float phi = nrc_acosf(dir.y / lxz);
    q[4] = phi / NRC_PI;
}

// global column of local column lx (nrc_tile: strips of 2^x_block_log2 columns, every x_stride-th strip)
__device__ __forceinline__ uint32_t global_x(const DevFrame& fr, uint32_t lx)
{
    const uint32_t b = fr.x_block_log2;
    return ((fr.x_offset + (lx >> b) * fr.x_stride) << b) + (lx & ((1u << b) - 1u));
}

// 16x16 pixel tile per 256-thread workgroup, 8x8 per wave (coherent paths inside a wave)
__device__ __forceinline__ bool pixel_of_thread(const DevFrame& fr, uint32_t* lx, uint32_t* y)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    *lx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    *y = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    return *lx < fr.w && *y < fr.h;
}

// Camera kernels: each wave renders one 8x8 pixel tile (coherent paths inside a wave), four tiles of a row per workgroup.
// Tile rows are issued centre-out, so the expensive middle of the image starts first and the cheap rim fills the tail of the
// launch.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD (b + c) mod 8, MI355X_MICROARCH.md) and an
// XCD never helps another out, so the MAPPING must spread the expensive tiles evenly: with 60 workgroups per tile row (1920
// pixels) the XCD of a column alternates between only two values from row to row, every XCD ends up with its own comb of
// columns, and the per-XCD work differed by 1.9x between the lightest and the heaviest (tools/loop_profile.py) -- the launch
// ended with three of eight XCDs busy.  Rows are therefore padded to an ODD number of workgroups (the padding workgroup exits):
// the XCD of a column then walks through all eight values over eight consecutive rows.
// On top of that mapping the tiles are launched costliest first (DevFrame::tile_order, k_tile_order below): slot d of the grid
// renders tile order[d].  (Measured and rejected, tools/loop_profile.py: one-wave workgroups.)
#ifndef NRC_CAMERA_WAVES_PER_BLOCK
#define NRC_CAMERA_WAVES_PER_BLOCK 4
#endif
constexpr uint32_t CAMERA_WAVES_PER_BLOCK = NRC_CAMERA_WAVES_PER_BLOCK;
// (wave priority of the camera kernels: NRC_RAISE_WAVE_PRIORITY in nrc_common.hpp)
__device__ __forceinline__ void camera_wave_priority(const DevFrame& fr)
{
    if (!((NRC_DIAG_LOWPRIO) & 8) && fr.raise_priority != 0u) __builtin_amdgcn_s_setprio(NRC_WAVE_PRIORITY);
}
__host__ __device__ inline uint32_t camera_row_blocks(uint32_t w)
{
    const uint32_t blocks_x = (((w + 7u) >> 3) + CAMERA_WAVES_PER_BLOCK - 1u) / CAMERA_WAVES_PER_BLOCK;
    return blocks_x | 1u;
}
// *slot: the wave's default slot (index into DevFrame::tile_cost)
// The renderer's query / radiance buffers are tile-major INSIDE the renderer: query ((ty * tiles_x + tx) * 64 + (y & 7) * 8 + (x & 7)) is
// pixel (x, y) of the camera kernels' 8x8 tile (tx, ty) -- a gen_rays wave stores its 64 queries as one 1 280-byte run, and the 32
// queries of an inference tile are four rows of eight neighbouring pixels, so the compositing epilogue of the inference kernel reads
// and writes whole 128-byte row segments of the images.  Tiles beyond the image's edge keep all-zero queries (never computed).  The
// reference's x * H + y order (nrc/prep_infer_rays.comp:31) is what nrc_renderer_buffer hands out (k_query_layout), and what the
// cache's own API speaks.
__host__ __device__ inline size_t query_index(uint32_t w, uint32_t lx, uint32_t y)
{
    const uint32_t tiles_x = (w + 7u) >> 3;
    return ((size_t)((y >> 3) * tiles_x + (lx >> 3)) << 6) + ((y & 7u) << 3) + (lx & 7u);
}
// launch slot (workgroup * 4 + wave) -> the wave's pixel
__device__ __forceinline__ bool pixel_of_launch_slot(const DevFrame& fr, uint32_t d, uint32_t* lx, uint32_t* y, uint32_t* slot = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_x = (fr.w + 7u) >> 3, tiles_y = (fr.h + 7u) >> 3;
    const uint32_t row_blocks = camera_row_blocks(fr.w);
    if (fr.tile_order != nullptr) d = scalar_load(fr.tile_order + d);      // costliest tiles first
    if (slot) *slot = d;
    const uint32_t bd = d / CAMERA_WAVES_PER_BLOCK;
    const uint32_t k = bd / row_blocks, jb = bd - k * row_blocks;
    const uint32_t tx = jb * CAMERA_WAVES_PER_BLOCK + (d - bd * CAMERA_WAVES_PER_BLOCK);
    if (k >= tiles_y || tx >= tiles_x) return false;
    const uint32_t mid = tiles_y >> 1;
    const uint32_t ty = (k & 1u) ? mid - ((k + 1u) >> 1) : mid + (k >> 1);      // mid, mid-1, mid+1, ...: a bijection
    *lx = tx * 8u + (lane & 7u);
    *y = ty * 8u + (lane >> 3);
    return *lx < fr.w && *y < fr.h;
}
__device__ __forceinline__ bool pixel_of_wave_tile(const DevFrame& fr, uint32_t* lx, uint32_t* y, uint32_t* slot = nullptr)
{
    return pixel_of_launch_slot(fr, __builtin_amdgcn_readfirstlane(blockIdx.x * CAMERA_WAVES_PER_BLOCK + (threadIdx.x >> 6)), lx, y, slot);
}

// Empty-space early-out, exact: the tile mask (k_tile_mask) clears the bit of an 8x8 tile only when no camera ray of the tile can
// come within a voxel of non-empty density.  Every density such a ray's delta tracking fetches is 0, so every tentative collision
// is rejected and the walk can end in two ways only: it leaves the volume (the pixel is env(rd), didScatter = 0, an all-zero
// query, and the RNG state behind the walk is never used), or it runs into DeltaTrack's cap of 128 collisions
// (path_trace.glsl:161-173) and "scatters" at a random point of the segment.  Which of the two happens is a function of the
// pixel's RNG state alone: the hash chain has only 2^23 states, and k_flight_table computes, for every state, the optical distance
// the 128 free flights drawn from it cover.  A state whose distance exceeds the largest optical depth any ray of the scene can
// have (with a margin for the rounding of the walk's own sums) provably leaves the volume before the cap; the others ("capped":
// k_flight_select) reach the kernels as a short list -- for the bench scene the chain's fixed point 0 alone (hash(0) = 0: every
// flight has length 0) -- or, for denser media, as one bit per state.  A tile's waves skip the walk only when no pixel of the
// tile is in a capped state.
__device__ __forceinline__ bool tile_mask_clear(const DevFrame& fr, uint32_t lx, uint32_t y)
{
    if (fr.tile_mask == nullptr || fr.flight_mode == 0u) return false;
    const uint32_t tiles_x = (fr.w + 7u) >> 3, tiles_y = (fr.h + 7u) >> 3;
    const uint32_t id = __builtin_amdgcn_readfirstlane((y >> 3) * tiles_x + (lx >> 3));      // wave-uniform: one tile per wave
    const uint32_t n_words = (tiles_x * tiles_y + 31u) >> 5;
    uint32_t word, ignore;
    scalar_load2(fr.tile_mask + (id >> 5), fr.tile_mask + n_words, &word, &ignore);
    return ignore == 0u && ((word >> (id & 31u)) & 1u) == 0u;
}
// is a pixel of the wave's tile in a capped RNG state?  (wave-uniform)
__device__ __forceinline__ bool tile_has_capped_state(const DevFrame& fr, bool inside, float rng0)
{
    // state -> mantissa: the states are the multiples of 2^-23 in [0, 1), so rng0 + 1 is exact
    const uint32_t m = nrc_f2u(rng0 + 1.0f) & 0x007fffffu;
    bool capped = false;
    if (fr.flight_mode == 1u) {      // a handful of states: compared one by one (no memory access)
#pragma unroll
        for (uint32_t k = 0; k < kFlightListMax; k++) capped |= (k < fr.flight_n) & (m == fr.flight_list[k]);
    } else {
        capped = ((fr.flight_bits[m >> 5] >> (m & 31u)) & 1u) != 0u;
    }
    return __ballot(inside & capped) != 0ull;
}
__device__ __forceinline__ bool tile_is_empty(const DevFrame& fr, uint32_t lx, uint32_t y, bool inside, float rng0)
{
    return tile_mask_clear(fr, lx, y) && !tile_has_capped_state(fr, inside, rng0);
}

// optical distance (in units of 1 / sigma_max) covered by the 128 free flights of a delta walk that starts in RNG state
// float_construct(m) and rejects every tentative collision: draws alternate flight, acceptance (path_trace.glsl:163-170)
__global__ __launch_bounds__(256) void k_flight_table(float* __restrict__ table)
{
    fill_log_table();
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    float r = float_construct(m);
    float d = 0.0f;
    for (int k = 0; k < 128; k++) {
        const float s = random1(r);
        d = d - nrc_logf(1.0f - s, s_log_tab);
        r = random1(s);
    }
    table[m] = d;
}

__global__ __launch_bounds__(256) void k_flight_select(const float* __restrict__ table, float lambda, uint32_t* __restrict__ out, uint32_t* __restrict__ bits)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    const bool capped = !(table[m] > lambda);
    const unsigned long long b = __ballot(capped);
    if ((threadIdx.x & 63u) == 0u) {
        bits[m >> 5] = (uint32_t)b;
        bits[(m >> 5) + 1u] = (uint32_t)(b >> 32);
    }
    if (capped) {
        const uint32_t k = atomicAdd(&out[0], 1u);
        if (k < kFlightListMax) out[1u + k] = m;
    }
}

// the tiles with a pixel in a capped RNG state (list mode), for DevFrame::hot_tiles: hot[kHotTilesMax] counts them (zeroed by the
// caller), the first kHotTilesMax are kept in hot[0..].  Used in front of a camera kernel whose list is not there yet -- k_gen_rays
// builds the next frame's list itself (DevFrame::hot_next); this kernel serves its first frame, pinned random numbers and
// k_mc_render.  One-wave workgroups of 10 VGPRs: they find room beside the camera kernels of other renderers, whose five waves
// per SIMD leave 32 of a lane's 512 VGPRs (1024-thread workgroups waited for such a kernel to thin out: 99 us instead of 3).
// Eight pixels per thread.
__global__ __launch_bounds__(64) void k_hot_tiles(DevFrame fr, uint32_t* __restrict__ hot)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    const uint32_t n = fr.w * fr.h;
    for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
        const uint32_t y = i / fr.w, lx = i - y * fr.w;
        const float u = (float)global_x(fr, lx) * fr.inv_gw, v = (float)y * fr.inv_gh;
        const float rng0 = random2(random2(u, v), random4(fr.random));      // init_random
        const uint32_t m = nrc_f2u(rng0 + 1.0f) & 0x007fffffu;
        bool capped = false;
#pragma unroll
        for (uint32_t k = 0; k < kFlightListMax; k++) capped |= (k < fr.flight_n) & (m == fr.flight_list[k]);
        if (capped) {
            const uint32_t k = atomicAdd(&hot[kHotTilesMax], 1u);
            if (k < kHotTilesMax) hot[k] = ((y >> 3) << 16) | (lx >> 3);
        }
    }
}

// The tile of a camera kernel's wave.  With a hot-tile list (DevFrame::hot_tiles) the launch has kHotTilesMax waves in front of the
// ordered ones: wave k traces hot tile k, and the wave the order gives that tile to leaves.  false: the wave has nothing to do.
__device__ __forceinline__ bool camera_wave_tile(const DevFrame& fr, uint32_t* lx_, uint32_t* y_, uint32_t* slot_, bool* inside_, bool* hot_wave_)
{
    uint32_t lx = 0, y = 0, slot = 0;
    bool inside;
    bool hot_wave = false;
    {
        uint32_t d = __builtin_amdgcn_readfirstlane(blockIdx.x * CAMERA_WAVES_PER_BLOCK + (threadIdx.x >> 6));
        if (fr.hot_tiles != nullptr) {
            // the launch has kHotTilesMax waves in front of the ordered ones: wave k traces hot tile k, and the wave the order
            // gives that tile to leaves
            typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
            uint32_t hot_n;
            u32x8 hv;
            asm volatile("s_load_dword %0, %2, 0x20\n\ts_load_dwordx8 %1, %2, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(hot_n), "=&s"(hv) : "s"(fr.hot_tiles) : "memory");
            hot_n = min(hot_n, kHotTilesMax);
            const uint32_t hot[kHotTilesMax] = {hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]};
            if (d < kHotTilesMax) {
                if (d == 0u && fr.hot_reset != nullptr && (threadIdx.x & 63u) == 0u) *fr.hot_reset = 0u;
                if (d >= hot_n) return false;
                uint32_t t = 0;
#pragma unroll
                for (uint32_t k = 0; k < kHotTilesMax; k++) t = d == k ? hot[k] : t;
                // k_hot_tiles appends one entry per capped PIXEL: two such pixels in one tile list the tile twice, and the tile must
                // still be traced exactly once (k_mc_render blends in place) -- the later duplicate leaves
                bool dup = false;
#pragma unroll
                for (uint32_t k = 0; k + 1u < kHotTilesMax; k++) dup |= (k < d) & (hot[k] == t);
                if (dup) return false;
                const uint32_t lane = threadIdx.x & 63u;
                lx = (t & 0xffffu) * 8u + (lane & 7u);
                y = (t >> 16) * 8u + (lane >> 3);
                inside = lx < fr.w && y < fr.h;
                hot_wave = true;
            } else {
                inside = pixel_of_launch_slot(fr, d - kHotTilesMax, &lx, &y, &slot);
                const uint32_t t = __builtin_amdgcn_readfirstlane(((y >> 3) << 16) | (lx >> 3));
                bool is_hot = false;
#pragma unroll
                for (uint32_t k = 0; k < kHotTilesMax; k++) is_hot |= (k < hot_n) & (t == hot[k]);
                if (is_hot) return false;
            }
        } else {
            inside = pixel_of_launch_slot(fr, d, &lx, &y, &slot);
        }
    }
    *lx_ = lx; *y_ = y; *slot_ = slot; *inside_ = inside; *hot_wave_ = hot_wave;
    return true;
}

// what a tile cost in this launch, kept as a decaying maximum over the sampled launches (DevFrame::tile_cost_keep)
__device__ __forceinline__ void store_tile_cost(const DevFrame& fr, uint32_t slot, unsigned long long cycles)
{
    uint32_t c = (uint32_t)min(cycles, 0xffffffffull);
    if (fr.tile_cost_keep != 0u) {
        const uint32_t old = fr.tile_cost[slot];
        c = max(c, old - (old >> fr.tile_cost_keep));
    }
    fr.tile_cost[slot] = c;
}

__device__ __forceinline__ void count_fetches(unsigned long long* counter, uint32_t n)
{
    if (counter == nullptr) return;
    unsigned long long v = n;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(counter, v);
}

// The stores of k_gen_rays' frame outputs (primary colour, scatter flag, query: 40 B per pixel = 83 MB per 1080p launch) are plain
// write-back stores; write-through (sc0 sc1) and non-temporal stores were measured and lost (docs/MEASUREMENT_LOG.md).
typedef float out_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void out_store(float4* p, float4 v)
{
    *p = v;
}
__device__ __forceinline__ void out_store(float* p, float v)
{
    *p = v;
}

// ------------------------------------------------------------------------------------------------ nrc/gen_rays.comp + prep_infer_rays.comp
#ifndef NRC_GEN_WAVES_PER_SIMD
#define NRC_GEN_WAVES_PER_SIMD NRC_CAMERA_WAVES_PER_SIMD
#endif
template <bool COUNT>
__global__ __launch_bounds__(64 * CAMERA_WAVES_PER_BLOCK, NRC_GEN_WAVES_PER_SIMD) void k_gen_rays(DevScene sc, DevCamera cam, DevFrame fr, uint32_t primary_ray_length,
                                                 float primary_ray_prob, float4* __restrict__ primary,
                                                 float* __restrict__ info, float4* __restrict__ origin,
                                                 float4* __restrict__ dirs, float* __restrict__ infer_in,
                                                 unsigned long long* fetch_counter, TrainGrid tg, int full_vertex_images)
{
    camera_wave_priority(fr);
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    __shared__ uint32_t s_occ[kOccMaxWords];
    uint32_t lx = 0, y = 0, slot = 0;
    bool inside, hot_wave;
    if (!camera_wave_tile(fr, &lx, &y, &slot, &inside, &hot_wave)) return;
#ifdef NRC_LOOP_PROFILE
    const uint32_t wave_id = blockIdx.x * CAMERA_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0 && wave_id < 65536u) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_wave_times[4 * wave_id] = wall_clock64();
        g_wave_times[4 * wave_id + 2] = xcc;
        g_wave_times[4 * wave_id + 3] = hw;
    }
#endif
    if (__ballot(inside) == 0ull) return;      // a padding workgroup's wave (the grid is rounded up to whole rows of workgroups)
    CtxT<COUNT> c{sc, 0.0f, 0u};
    const uint32_t gx = global_x(fr, lx);
    const float u = (float)gx * fr.inv_gw, v = (float)y * fr.inv_gh;
    const float seed_uv = random2(u, v);      // the pixel's part of init_random's seed
    if (fr.hot_next != nullptr) {             // the next frame's hot tiles (DevFrame::hot_next): is a pixel of this tile in a capped state then?
        const uint32_t m = nrc_f2u(random2(seed_uv, random4(fr.random_next)) + 1.0f) & 0x007fffffu;
        bool capped = false;
#pragma unroll
        for (uint32_t k = 0; k < kFlightListMax; k++) capped |= (k < fr.flight_n) & (m == fr.flight_list[k]);
        if (__ballot(inside & capped) != 0ull && (threadIdx.x & 63u) == 0u) {
            const uint32_t k = atomicAdd(&fr.hot_next[kHotTilesMax], 1u);
            if (k < kHotTilesMax) fr.hot_next[k] = __builtin_amdgcn_readfirstlane(((y >> 3) << 16) | (lx >> 3));
        }
    }
    // Two of three tiles of the bench view miss the medium.  Their waves are the tail of the launch (k_tile_order starts the
    // costliest tiles first): the tile's bit comes through the scalar cache, and the wave is gone before the occupancy table is
    // copied (there is no workgroup barrier in this kernel: load_occupancy_per_wave).  What is left of them is dispatch: ~1000
    // waves per microsecond start at the end of the launch (tools/loop_profile.py), and all they do costs 0.004 ms of 0.215.
    if (tile_mask_clear(fr, lx, y)) {
        c.rng = random2(seed_uv, random4(fr.random));      // init_random
        if (!tile_has_capped_state(fr, inside, c.rng)) {
            if (inside) {
                V3 ro, rd;
                camera_ray(cam, u, v, &ro, &rd);
                const V3 e = sample_env_dir(sc, rd);
                const size_t pix = (size_t)y * fr.w + lx;
                out_store(&primary[pix], make_float4(e.x, e.y, e.z, 1.0f));
                out_store(&info[pix], 0.0f);
                if (fr.skip_dead_queries == 0u) {
                    float* qo = infer_in + query_index(fr.w, lx, y) * 5u;
#pragma unroll
                    for (int k = 0; k < 5; k++) out_store(&qo[k], 0.0f);
                }
            }
            if (fr.tile_cost != nullptr && !hot_wave && (threadIdx.x & 63u) == 0) store_tile_cost(fr, slot, __builtin_amdgcn_s_memtime() - t_start);
#ifdef NRC_LOOP_PROFILE
            if (inside && full_vertex_images != 0) reinterpret_cast<float*>(origin)[4 * ((size_t)y * fr.w + lx) + 3] = 0.0f;
            if ((threadIdx.x & 63u) == 0 && wave_id < 65536u) g_wave_times[4 * wave_id + 1] = wall_clock64();
#endif
            return;
        }
    }
    c.occ = load_occupancy_per_wave(sc, s_occ);
    // Wave-uniform control flow with per-lane predicates from here to the stores: a lane whose path has ended (or that has none)
    // stays in the instruction stream, so that the tracking loops can hand the last walks to lane pairs (ratio_pairs).
    V3 ro, rd;
    camera_ray(cam, u, v, &ro, &rd);
    c.rng = random2(seed_uv, random4(fr.random));      // init_random
    const bool enter = inside;
    V3 entry = ro, ex;
    {
#ifdef NRC_LOOP_PROFILE
        c.fee_kind = 0;
#endif
        find_entry_exit(c, sel(enter, ro, v3(0.0f, 0.0f, 0.0f)), sel(enter, rd, v3(0.0f, 0.0f, 1.0f)), &entry, &ex);
#ifdef NRC_LOOP_PROFILE
        c.fee_kind = 1;
#endif
    }
    const bool entered = enter && !(sky_sdf(sc, entry) > 100000.0f);
    // The colour of a pixel that does not scatter -- the environment along its camera ray, throughput 1 (gen_rays.comp:86-95) -- is
    // stored NOW, before the walk; a pixel that scatters overwrites it at the end (same lane, same address: the stores are
    // performed in order).  The camera ray direction is therefore dead during the walk; kept live for the end, two of its
    // components were the one value this kernel spilled to scratch (tests/test_abi.py checks that there is none).
    if (inside) {
        const V3 e = sample_env_dir(sc, rd);
        out_store(&primary[(size_t)y * fr.w + lx], make_float4(e.x, e.y, e.z, 1.0f));
    }
    V3 light = v3(0, 0, 0);
    V3 cur = entry, dir = rd;      // TracePath recomputes the same entry (gen_rays.comp:11)
    float factor = 1.0f;
    bool did_scatter = false, walking = entered;
    for (int i = 0;; i++) {
        if (__ballot(walking) == 0ull) break;
        bool vexit = false;
#ifdef NRC_DIAG_CUT_TAIL2
        c.cut2 = i >= 1 ? (uint32_t)(NRC_DIAG_CUT_TAIL2) : 0u;
#endif
        const V3 nc = delta_track<true>(c, cur, dir, &vexit, walking);
        cur = sel(walking, nc, cur);
        walking &= !vexit;
        did_scatter |= walking;
        factor = walking ? factor * 0.5f : factor;
        if (__ballot(walking) == 0ull) break;
        const V3 ts = trace_scene<true>(c, cur, dir, walking);
        if (walking) {
            light = add(light, mul(ts, factor));
            dir = new_ray_dir(c, dir, true);
            if ((uint32_t)i >= primary_ray_length) {
                if (c.rand(1.0f) >= primary_ray_prob || i == 128) walking = false;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)y * fr.w + lx;
        float q[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (entered) {
            // the NRC vertex images (nrcRayOrigin / nrcRayDir of gen_rays.comp:97-100) have one reader besides the query packing
            // fused in here: prep_train_rays, at the pixels (tx * xDist, ty * yDist) of the train grid.  Only those are stored
            // (32 B for 16 384 of 2 073 600 pixels instead of 66 MB per frame) unless the caller asks for the whole images.
            bool on_grid = full_vertex_images != 0;
            if (!on_grid) {
                const uint32_t qx = tg.x_dist ? lx / tg.x_dist : 0u, qy = tg.y_dist ? y / tg.y_dist : 0u;
                on_grid = (qx * tg.x_dist == lx) & (qy * tg.y_dist == y) & (qx < tg.tw) & (qy < tg.th);
            }
            if (on_grid) {
                origin[pix] = make_float4(cur.x, cur.y, cur.z, 0.0f);
                dirs[pix] = make_float4(dir.x, dir.y, dir.z, 0.0f);
            }
            if (did_scatter) {
                nrc_query(sc, cur, dir, q);
                out_store(&primary[pix], make_float4(light.x, light.y, light.z, factor));      // replaces the environment colour stored above
            }
        }
        out_store(&info[pix], did_scatter ? 1.0f : 0.0f);
        // the reference zero-fills the query buffer each frame (vkCmdFillBuffer, NrcHpmRenderer.cu:1996) and prep_infer_rays writes only
        // scattered pixels.  Here: no memset; without the live-query list every slot is written (zeros for the others: the list-free
        // inference recognises them), with it only the scattered pixels' (DevFrame::skip_dead_queries: nobody reads the rest)
        if (fr.skip_dead_queries == 0u || (entered && did_scatter)) {
            float* qo = infer_in + query_index(fr.w, lx, y) * 5u;
#pragma unroll
            for (int k = 0; k < 5; k++) out_store(&qo[k], q[k]);
        }
    }
    if (fr.live_list != nullptr) {      // the frame's live queries, which the renderer's inference walks (DevFrame::live_list): one atomic per wave
        const bool live = inside && entered && did_scatter;
        const unsigned long long lm = __ballot(live);
        if (lm != 0ull) {
            uint32_t base = 0;
            if ((threadIdx.x & 63u) == 0u) base = atomicAdd(fr.live_count, (uint32_t)__popcll(lm));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
            if (live) fr.live_list[base + rank] = (uint32_t)query_index(fr.w, lx, y);
        }
    }
#ifdef NRC_LOOP_PROFILE
    // per-pixel look-up count in the w component of the origin image (tools/lane_model.py)
    if (inside && full_vertex_images != 0) reinterpret_cast<float*>(origin)[4 * ((size_t)y * fr.w + lx) + 3] = (float)c.fetches;
#endif
    if constexpr (COUNT) count_fetches(fetch_counter, c.fetches);
    // what this tile cost (shader cycles): next frames launch the costliest tiles first (k_tile_order).  (Not for a hot wave: what
    // it measured is this frame's one pixel in a capped state, not the tile.)
    if (fr.tile_cost != nullptr && !hot_wave && (threadIdx.x & 63u) == 0) store_tile_cost(fr, slot, __builtin_amdgcn_s_memtime() - t_start);
#if defined(NRC_LOOP_PROFILE) && !defined(NRC_NO_LOOP_COUNTERS)
    for (int k = 0; k < 8; k++) { count_fetches(&g_loop_prof[k], c.useful[k]); count_fetches(&g_loop_prof[8 + k], c.issued[k]); }
#endif
#ifdef NRC_LOOP_PROFILE
    if ((threadIdx.x & 63u) == 0 && wave_id < 65536u) g_wave_times[4 * wave_id + 1] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------ mc/render.comp
template <bool COUNT>
__global__ __launch_bounds__(64 * CAMERA_WAVES_PER_BLOCK, NRC_CAMERA_WAVES_PER_SIMD) void k_mc_render(DevScene sc, DevCamera cam, DevFrame fr, uint32_t path_length,
                                                  float blend_factor, float4* __restrict__ out_rgba,
                                                  float* __restrict__ info, unsigned long long* fetch_counter)
{
    camera_wave_priority(fr);
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    __shared__ uint32_t s_occ[kOccMaxWords];
    const uint32_t* occ = load_occupancy(sc, s_occ);
    uint32_t lx = 0, y = 0, slot = 0;
    bool inside, hot_wave;
    if (!camera_wave_tile(fr, &lx, &y, &slot, &inside, &hot_wave)) return;
    CtxT<COUNT> c{sc, 0.0f, 0u};
    c.occ = occ;
    // wave-uniform control flow with per-lane predicates, as in k_gen_rays (the thin trips of the 32 x 3 tracking loops go to lane pairs)
    const uint32_t gx = global_x(fr, lx);
    const float u = (float)gx * fr.inv_gw, v = (float)y * fr.inv_gh;
    V3 ro, rd;
    camera_ray(cam, u, v, &ro, &rd);
    init_random(c, u, v, fr.random);
    const bool empty = tile_is_empty(fr, lx, y, inside, c.rng);      // wave-uniform, see tile_is_empty
    const bool enter = inside & !empty;
    V3 entry = ro, ex;
    if (__ballot(enter) != 0ull) find_entry_exit(c, sel(enter, ro, v3(0.0f, 0.0f, 0.0f)), sel(enter, rd, v3(0.0f, 0.0f, 1.0f)), &entry, &ex);
    const bool entered = enter && !(sky_sdf(sc, entry) > 100000.0f);
    V3 light = v3(0, 0, 0);
    V3 cur = entry, dir = rd;
    float factor = 1.0f;
    bool did_scatter = false, walking = entered;
    for (uint32_t i = 0; i < path_length; i++) {
        if (__ballot(walking) == 0ull) break;
        bool vexit = false;
        const V3 nc = delta_track<true>(c, cur, dir, &vexit, walking);
        cur = sel(walking, nc, cur);
        walking &= !vexit;
        did_scatter |= walking;
        factor = walking ? factor * 0.5f : factor;
        if (__ballot(walking) == 0ull) break;
        const V3 ts = trace_scene<true>(c, cur, dir, walking);
        if (walking) {
            light = add(light, mul(ts, factor));
            dir = new_ray_dir(c, dir, true);
        }
    }
    if (inside) {
        V3 col = light;
        if (!did_scatter) col = sample_env_dir(sc, rd);
        const float a = did_scatter ? 1.0f : 0.0f;
        const size_t pix = (size_t)y * fr.w + lx;
        const float4 prev = out_rgba[pix];
        const float ib = 1.0f - blend_factor;
        out_rgba[pix] = make_float4(blend_factor * col.x + ib * prev.x, blend_factor * col.y + ib * prev.y,
                                    blend_factor * col.z + ib * prev.z, blend_factor * a + ib * prev.w);
        if (info) info[pix] = a;
    }
    if constexpr (COUNT) count_fetches(fetch_counter, c.fetches);
    if (fr.tile_cost != nullptr && !hot_wave && (threadIdx.x & 63u) == 0)      // see k_gen_rays / k_tile_order (a longer walk: 8 192-cycle classes)
        store_tile_cost(fr, slot, (__builtin_amdgcn_s_memtime() - t_start) >> 4);
}

// ------------------------------------------------------------------------------------------------ costliest-first launch order
// A tile costs between 2 us (provably empty) and 160 us (cloud interior), a launch holds ~4 200 of its 32 400 waves at a time, and
// the hardware starts workgroups in index order: in the default centre-out order the last expensive tiles start at 60 % of the
// launch and a third of its duration is a thinning tail (tools/loop_profile.py).  Tile costs repeat from frame to frame
// (correlation 0.95), so the waves are launched in order of decreasing cost of an earlier frame: a counting sort over 1024
// cost classes of 512 cycles, one workgroup, order within a class arbitrary -- any permutation gives the same frame.
// (Measured and taken out again, rounds 2-4: ranking a tile by the maximum over its four neighbours; listing the costliest tiles as two
// half tiles -- DESIGN.md section 4.)
__global__ __launch_bounds__(1024) void k_tile_order(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ order)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    __shared__ uint32_t hist[1024];
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0u;
    __syncthreads();
    auto key_of = [=](uint32_t i) { return 1023u - min(cost[i] >> 9, 1023u); };      // descending cost
    for (uint32_t i = tid; i < n; i += 1024u) atomicAdd(&hist[key_of(i)], 1u);
    __syncthreads();
    // exclusive prefix sum over the 1024 classes (wave scans + wave totals)
    const uint32_t v = hist[tid];
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)(tid & 63u) >= off) incl += t;
    }
    __shared__ uint32_t wsum[16];
    if ((tid & 63u) == 63u) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) woff += wsum[w];
    __syncthreads();
    hist[tid] = woff + incl - v;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024u) order[atomicAdd(&hist[key_of(i)], 1u)] = i;
}

// XCD-aware finish of the order (round 4; xcd_row_len = slots per tile row).  The hardware deals workgroups to the eight XCDs round-robin, so
// the 32 M consecutive ranks of workgroups 8 M j .. 8 M (j + 1) - 1 -- tiles of about the same cost -- start together, 4 M on each XCD.  Within
// such a window the tiles are handed out by SCREEN ROW: the 4 M topmost to the first XCD, the next to the second, ...  Every XCD then works
// on a band of rows -- in whatever class of cost the launch is at -- and the volume's [z][y][x] lines its camera rays touch are its own: each
// 4 MB L2 sees a part of the volume instead of all of it.  The order stays a permutation, costliest first.  wg_off = the workgroups in
// front of the ordered ones (the hot tiles' two).  A kernel of its own because k_tile_order is ONE workgroup.
__global__ __launch_bounds__(1024) void k_tile_order_xcd(uint32_t* __restrict__ order, uint32_t n, uint32_t xcd_row_len, uint32_t wg_off)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    // one workgroup per window of blockDim.x = 32 M ranks (M workgroups per XCD); a window that is not whole (the first one behind the hot
    // tiles' workgroups, the last one) keeps its order
    __shared__ uint32_t keys[1024];
    const uint32_t S = blockDim.x, m4 = S >> 3;                // ranks per XCD and window
    const uint32_t tid = threadIdx.x;
    const uint32_t tiles_y = n / xcd_row_len, mid = tiles_y >> 1;
    const long long r0 = (long long)blockIdx.x * S - (long long)wg_off * 4;      // rank of the window's first slot
    if (r0 < 0 || (unsigned long long)r0 + S > n) return;      // (workgroup-uniform)
    const uint32_t t = order[(uint32_t)r0 + tid];
    const uint32_t k = t / xcd_row_len, j = t - k * xcd_row_len;
    const uint32_t ty = (k & 1u) ? mid - ((k + 1u) >> 1) : mid + (k >> 1);      // pixel_of_launch_slot's row bijection
    const uint32_t key = (ty << 16) | j;                       // unique per tile
    keys[tid] = key;
    __syncthreads();
    uint32_t pos = 0u;
    for (uint32_t q = 0; q < S; q++) pos += keys[q] < key ? 1u : 0u;
    // position pos of the window's tiles by screen row -> XCD pos / m4, its (pos % m4)-th slot: workgroup xcd + 8 * (slot / 4) of the window
    const uint32_t xcd = pos / m4, q = pos - xcd * m4;
    order[(uint32_t)r0 + ((xcd + 8u * (q >> 2)) << 2) + (q & 3u)] = t;
}

// ------------------------------------------------------------------------------------------------ empty-space tile mask
// One thread per occupancy box (world-space AABB around a run of non-empty 8^3-voxel cells, grown by one voxel): its eight
// corners are projected with the camera's forward transform; the screen rectangle around them, grown by a pixel, covers every
// pixel whose camera ray can pass through the box (a pixel's ray consists of the points that project onto the pixel).  The 8x8
// tiles the rectangle touches are marked.  A box with a corner at or behind the eye plane switches the mask off for this camera.
__global__ __launch_bounds__(256) void k_tile_mask(const float* __restrict__ boxes, uint32_t n_boxes, DevProjView pv, DevFrame fr,
                                                  uint32_t* __restrict__ mask)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_boxes) return;
    const uint32_t tiles_x = (fr.w + 7u) >> 3, tiles_y = (fr.h + 7u) >> 3;
    const uint32_t n_words = (tiles_x * tiles_y + 31u) >> 5;
    const float* b = boxes + 6u * (size_t)i;
    float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
    bool behind = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x = b[(k & 1) ? 3 : 0], y = b[(k & 2) ? 4 : 1], z = b[(k & 4) ? 5 : 2];
        const float cx = pv.m[0] * x + pv.m[4] * y + pv.m[8] * z + pv.m[12];
        const float cy = pv.m[1] * x + pv.m[5] * y + pv.m[9] * z + pv.m[13];
        const float cw = pv.m[3] * x + pv.m[7] * y + pv.m[11] * z + pv.m[15];
        if (!(cw > 1.0e-3f)) { behind = true; continue; }
        const float nx = cx / cw, ny = cy / cw;
        xmin = fminf(xmin, nx); xmax = fmaxf(xmax, nx);
        ymin = fminf(ymin, ny); ymax = fmaxf(ymax, ny);
    }
    if (behind || !(xmin <= xmax) || !(ymin <= ymax)) {      // NaN-safe
        atomicOr(&mask[n_words], 1u);
        return;
    }
    // pixel (gx, y) looks along ndc = (2 gx / gw - 1, 2 y / gh - 1): gx = (ndc.x + 1) / 2 * gw
    const float gw = 1.0f / fr.inv_gw, gh = 1.0f / fr.inv_gh;
    float gx0 = floorf((xmin + 1.0f) * 0.5f * gw) - 1.0f, gx1 = ceilf((xmax + 1.0f) * 0.5f * gw) + 1.0f;
    float gy0 = floorf((ymin + 1.0f) * 0.5f * gh) - 1.0f, gy1 = ceilf((ymax + 1.0f) * 0.5f * gh) + 1.0f;
    // local columns: strip s = gx >> b of the global frame is the local strip (s - x_offset) / x_stride (global_x); every local
    // column of the local strips the rectangle can touch is taken (conservative)
    const float blk = (float)(1u << fr.x_block_log2);
    float lx0 = floorf((floorf(gx0 / blk) - (float)fr.x_offset) / (float)fr.x_stride) * blk;
    float lx1 = (ceilf((floorf(gx1 / blk) - (float)fr.x_offset) / (float)fr.x_stride) + 1.0f) * blk - 1.0f;
    lx0 = fmaxf(lx0, 0.0f); gy0 = fmaxf(gy0, 0.0f);
    lx1 = fminf(lx1, (float)(fr.w - 1u)); gy1 = fminf(gy1, (float)(fr.h - 1u));
    if (!(lx0 <= lx1) || !(gy0 <= gy1)) return;             // off screen
    const uint32_t tx0 = (uint32_t)lx0 >> 3, tx1 = (uint32_t)lx1 >> 3, ty0 = (uint32_t)gy0 >> 3, ty1 = (uint32_t)gy1 >> 3;
    for (uint32_t ty = ty0; ty <= ty1; ty++)
        for (uint32_t tx = tx0; tx <= tx1; tx++) {
            const uint32_t id = ty * tiles_x + tx;
            const uint32_t bit = 1u << (id & 31u);
            if ((mask[id >> 5] & bit) == 0u) atomicOr(&mask[id >> 5], bit);
        }
}

// ------------------------------------------------------------------------------------------------ nrc/clear.comp + ring ordering
// scratch layout: [0..T) scatter flag, [T..2T) exclusive rank among its kind (push rank if scattered, pop rank otherwise),
// [2T] n_push, [2T+1] n_pop, [2T+2] head (wrapped), [2T+3] tail (wrapped).
// Deterministic replacement of the two atomic counters of prep_train_rays.comp:7-31: ranks in linear train-index order.
__global__ __launch_bounds__(1024) void k_train_scan(DevFrame fr, TrainGrid tg, const float* __restrict__ info,
                                                    uint32_t* __restrict__ ring, uint32_t* __restrict__ scratch)
{
    NRC_RAISE_WAVE_PRIORITY(4);
    __shared__ uint32_t wsum[16];
    const uint32_t T = tg.tw * tg.th;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // each thread owns a contiguous run of train indices: flags -> local count -> block-wide exclusive scan -> ranks.
    // Runs of <= 64 indices (T <= 65536, the reference's default 4 x 2^14 included) keep their flags in a register mask, so
    // the strided gathers from the info image are independent and all in flight together; longer runs re-read the flags.
    const uint32_t per = (T + 1023u) / 1024u;
    const uint32_t i0 = min(tid * per, T), i1 = min(i0 + per, T);
    const bool in_regs = per <= 64u;
    auto flag_of = [&](uint32_t i) -> uint32_t {
        const uint32_t tx = i % tg.tw, ty = i / tg.tw;
        const uint32_t rx = tx * tg.x_dist, ry = ty * tg.y_dist;
        return (rx < fr.w && ry < fr.h) ? (info[(size_t)ry * fr.w + rx] == 1.0f ? 1u : 0u) : 0u;   // OOB imageLoad -> 0 (Q1)
    };
    unsigned long long flags = 0ull;
    uint32_t cnt = 0;
    if (in_regs) {
#pragma unroll 8
        for (uint32_t j = 0; j < i1 - i0; j++) flags |= (unsigned long long)flag_of(i0 + j) << j;
        cnt = (uint32_t)__popcll(flags);
    } else {
        for (uint32_t i = i0; i < i1; i++) cnt += flag_of(i);
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (uint32_t w = 0; w < 16u; w++) {
        if (w < wave) woff += wsum[w];
        total += wsum[w];
    }
    uint32_t push_rank = woff + incl - cnt;        // scattered entries before i0
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t f = in_regs ? (uint32_t)(flags >> (i - i0)) & 1u : flag_of(i);
        scratch[i] = f;
        scratch[T + i] = f ? push_rank : (i - push_rank);
        push_rank += f;
    }
    if (tid == 0) {
        uint32_t head = ring[0], tail = ring[1];
        if (tg.ring_size > 0) {          // clear.comp:5-9
            head %= tg.ring_size;
            tail %= tg.ring_size;
        }
        scratch[2 * T] = total;
        scratch[2 * T + 1] = T - total;
        scratch[2 * T + 2] = head;
        scratch[2 * T + 3] = tail;
    }
}

// ------------------------------------------------------------------------------------------------ nrc/prep_train_rays.comp
// Phase A (k_prep_train): pop / trace / write train data; pushes are deferred into `pending` so that every pop sees the
// ring as it was at frame start.  Phase B (k_ring_push): apply pushes in linear order, advance head/tail.
// MODE 0: the whole of prep_train_rays.comp in one launch (pop / trace / write).  Long train paths (quirk Q2 fixed: up to 32 vertices) make
// that launch the longest of the frame -- a few hundred latency-bound waves, ~0.8 ms beside the camera kernels -- and its place in the
// frame graph (behind gen_rays(N), in front of ring_push(N), on ONE stream) makes the frames queue up behind it.  The frame graph then
// splits it: MODE 1 (k_train_start's work: tiny) takes the rays' start vertices from the images or the ring, writes them to `start`
// ([T][6] floats) and the training INPUT they determine; MODE 2 traces from `start` and writes the TARGET -- it touches neither the
// images nor the ring, so frame N's trace can run beside frame N + 1's on another stream.  Same arithmetic, same results, bit for bit.
template <int MODE>
__global__ __launch_bounds__(256) void k_prep_train(DevScene sc, DevFrame fr, TrainGrid tg, const float4* __restrict__ origin,
                                                   const float4* __restrict__ dirs, const uint32_t* __restrict__ ring,
                                                   const uint32_t* __restrict__ scratch, float* __restrict__ train_in,
                                                   float* __restrict__ train_target, uint32_t rays_per_wave, float* __restrict__ start)
{
    NRC_RAISE_WAVE_PRIORITY(4);
    __shared__ uint32_t s_occ[kOccMaxWords];
    const uint32_t* occ = load_occupancy(sc, s_occ);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t T = tg.tw * tg.th;
    uint32_t tx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    uint32_t ty = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    // wave-uniform control flow with per-lane predicates, as in k_gen_rays: lanes beyond the train grid or done with their path stay in
    // the instruction stream and help with the last walks of every tracking loop (ratio_pairs)
    bool in_grid = tx < tg.tw && ty < tg.th;
    // Long train paths (quirk Q2 fixed: up to TRAIN_RAY_LENGTH = 32 vertices, three walks each): the launch is a few hundred waves whose
    // duration is the longest of their 64 paths, on a chip with 5 120 wave slots.  rays_per_wave < 64 gives a wave only that many rays (its
    // first lanes; the others help with the pair tails): the maximum runs over fewer paths, every ratio walk starts on lane pairs, and four
    // to eight times as many waves share the chip -- the same rays, the same targets, bit for bit (a ray's result does not depend on its lane).
    if (rays_per_wave < 64u) {
        const uint32_t i_lin = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + wave) * rays_per_wave + lane;
        in_grid = lane < rays_per_wave && i_lin < T;
        tx = in_grid ? i_lin % tg.tw : 0u;
        ty = in_grid ? i_lin / tg.tw : 0u;
    }
    const uint32_t i = in_grid ? ty * tg.tw + tx : 0u;
    CtxT<false> c{sc, 0.0f, 0u};
    c.occ = occ;
    // seed from TRAIN coordinates over the render size (quirk Q6, prep_train_rays.comp:108); sharded: global column
    const uint32_t gx = global_x(fr, tx);
    init_random(c, (float)gx * fr.inv_gw, (float)ty * fr.inv_gh, fr.random);
    V3 ro = v3(0, 0, 0);
    V3 rdir = normalize(v3(1.0f, 1.0f, 1.0f));
    if constexpr (MODE == 2) {
        if (in_grid) {
            const float* r = start + 6 * (size_t)i;
            ro = v3(r[0], r[1], r[2]);
            rdir = v3(r[3], r[4], r[5]);
        }
    } else {
        const bool scat = in_grid && scratch[i] != 0u;
        if (scat) {
            const size_t p = (size_t)(ty * tg.y_dist) * fr.w + tx * tg.x_dist;
            const float4 o = origin[p], d = dirs[p];
            ro = v3(o.x, o.y, o.z);
            rdir = v3(d.x, d.y, d.z);
        } else if (in_grid && tg.ring_size > 0) {
            const uint32_t tail = scratch[2 * T + 3];
            const float* r = reinterpret_cast<const float*>(ring + 2) + 6 * (size_t)((tail + scratch[T + i]) % tg.ring_size);
            ro = v3(r[0], r[1], r[2]);
            rdir = v3(r[3], r[4], r[5]);
        }
    }
    if constexpr (MODE == 1) {      // the start vertices and the input they determine; the trace is another launch's
        if (in_grid) {
            float* r = start + 6 * (size_t)i;
            r[0] = ro.x; r[1] = ro.y; r[2] = ro.z; r[3] = rdir.x; r[4] = rdir.y; r[5] = rdir.z;
            if (tg.ring_size > 0) {
                float q[5];
                nrc_query(sc, ro, rdir, q);
#pragma unroll
                for (int k = 0; k < 5; k++) train_in[5 * (size_t)i + k] = q[k];
            }
        }
        return;
    }
    V3 target = v3(0, 0, 0);
    for (uint32_t s = 0; s < tg.spp; s++) {
        V3 light = v3(0, 0, 0);
        V3 en, ex;
        find_entry_exit(c, ro, rdir, &en, &ex);      // (lanes beyond the grid march the default ray from the centre)
        V3 cur = en, dir = rdir;
        float factor = 1.0f;
        bool walking = in_grid;
        for (uint32_t k = 0; k < tg.ray_length; k++) {
            if (__ballot(walking) == 0ull) break;
            bool vexit = false;
            const V3 nc = delta_track<true>(c, cur, dir, &vexit, walking);
            cur = sel(walking, nc, cur);
            walking &= !vexit;
            factor = walking ? factor * 0.5f : factor;
            if (__ballot(walking) == 0ull) break;
            const V3 ts = trace_scene<true>(c, cur, dir, walking);
            if (walking) {
                light = add(light, mul(ts, factor));
                dir = new_ray_dir(c, dir, true);
            }
        }
        target = add(target, light);
    }
    const float fs = (float)tg.spp;
    target = v3(target.x / fs, target.y / fs, target.z / fs);
    if (in_grid && tg.ring_size > 0) {
        if constexpr (MODE == 0) {
            float q[5];
            nrc_query(sc, ro, rdir, q);
#pragma unroll
            for (int k = 0; k < 5; k++) train_in[5 * (size_t)i + k] = q[k];
        }
        train_target[3 * (size_t)i + 0] = fminf(8.0f, target.x);
        train_target[3 * (size_t)i + 1] = fminf(8.0f, target.y);
        train_target[3 * (size_t)i + 2] = fminf(8.0f, target.z);
    }
}

__global__ void k_ring_push(DevFrame fr, TrainGrid tg, const float4* __restrict__ origin, const float4* __restrict__ dirs,
                            uint32_t* __restrict__ ring, const uint32_t* __restrict__ scratch)
{
    NRC_RAISE_WAVE_PRIORITY(4);
    const uint32_t T = tg.tw * tg.th;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (tg.ring_size == 0) return;
    const uint32_t head = scratch[2 * T + 2], n_push = scratch[2 * T];
    // pushes are applied in linear train-index order; when a frame pushes more entries than the ring holds, a slot is written
    // several times and the last push in that order must win: only the final ring_size pushes are written at all
    if (i < T && scratch[i] != 0u && scratch[T + i] + tg.ring_size >= n_push) {
        const uint32_t tx = i % tg.tw, ty = i / tg.tw;
        const size_t p = (size_t)(ty * tg.y_dist) * fr.w + tx * tg.x_dist;
        const float4 o = origin[p], d = dirs[p];
        float* r = reinterpret_cast<float*>(ring + 2) + 6 * (size_t)((head + scratch[T + i]) % tg.ring_size);
        r[0] = o.x; r[1] = o.y; r[2] = o.z;
        r[3] = d.x; r[4] = d.y; r[5] = d.z;
    }
    if (i == 0) {
        ring[0] = head + scratch[2 * T];         // un-wrapped, as atomicAdd leaves them; wrapped next frame (clear.comp)
        ring[1] = scratch[2 * T + 3] + scratch[2 * T + 1];
    }
}

// ------------------------------------------------------------------------------------------------ nrc/render.comp
__global__ __launch_bounds__(256) void k_composite(DevFrame fr, uint32_t show_nrc, float blend_factor,
                                                  const float4* __restrict__ primary, const float* __restrict__ info,
                                                  const float* __restrict__ infer_out, float4* __restrict__ out_rgba,
                                                  uint32_t* __restrict__ live_count_reset)
{
    NRC_RAISE_WAVE_PRIORITY(2);
    // the frame's live-query list (DevFrame::live_list) has had its last reader -- this launch is ordered behind the frame's inference --
    // and the set's next gen_rays waits for this launch: the count is reset here instead of by a memset on the render stream
    if (live_count_reset != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *live_count_reset = 0u;
    uint32_t lx, y;
    if (!pixel_of_thread(fr, &lx, &y)) return;
    const size_t pix = (size_t)y * fr.w + lx, lin = query_index(fr.w, lx, y);
    const float4 p = primary[pix];
    float cr = p.x, cg = p.y, cb = p.z;
    if (show_nrc == 1u && info[pix] == 1.0f) {
        cr += fmaxf(0.0f, infer_out[3 * lin + 0]) * p.w;
        cg += fmaxf(0.0f, infer_out[3 * lin + 1]) * p.w;
        cb += fmaxf(0.0f, infer_out[3 * lin + 2]) * p.w;
    }
    const float4 prev = out_rgba[pix];
    const float ib = 1.0f - blend_factor;
    out_rgba[pix] = make_float4(blend_factor * cr + ib * prev.x, blend_factor * cg + ib * prev.y,
                                blend_factor * cb + ib * prev.z, blend_factor * 1.0f + ib * prev.w);
}

// tile-major (query_index) -> the reference's x * H + y order, C floats per query: what nrc_renderer_buffer hands out
// info (the frame's didScatter image, or nullptr): the entries of pixels that did not scatter are handed out as zeros -- the reference's
// zero-filled query buffer (NrcHpmRenderer.cu:1996); inside the renderer those slots are not written at all (DevFrame::skip_dead_queries,
// the live-query list)
__global__ __launch_bounds__(256) void k_query_layout(DevFrame fr, uint32_t C, const float* __restrict__ tiled, float* __restrict__ linear,
                                                     const float* __restrict__ info)
{
    uint32_t lx, y;
    if (!pixel_of_thread(fr, &lx, &y)) return;
    const size_t q = query_index(fr.w, lx, y), lin = (size_t)lx * fr.h + y;
    const bool dead = info != nullptr && info[(size_t)y * fr.w + lx] != 1.0f;
    for (uint32_t c = 0; c < C; c++) linear[lin * C + c] = dead ? 0.0f : tiled[q * C + c];
}

// ------------------------------------------------------------------------------------------------ ref/cmp1, norm, cmp2
// deterministic two-level reductions in fp64 instead of float atomics
constexpr int CMP_BLOCKS = 256;

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    double r = sh[0];
    __syncthreads();
    return r;
}

// pass 1: per-block partials of {sq err, ref sum, own sum, valid count}; scratch[8 + 4*b + k]
__global__ __launch_bounds__(256) void k_compare_1(const float4* __restrict__ ref, const float4* __restrict__ own, uint32_t n,
                                                  double* __restrict__ scratch)
{
    __shared__ double sh[256];
    double se = 0, rs = 0, os = 0, cnt = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += CMP_BLOCKS * 256u) {
        const float4 r = ref[i], o = own[i];
        if (r.w == 0.0f) continue;            // cmp1.comp:32
        cnt += 1.0;
        const double dx = (double)o.x - r.x, dy = (double)o.y - r.y, dz = (double)o.z - r.z;
        se += dx * dx + dy * dy + dz * dz;
        rs += (double)r.x + r.y + r.z;
        os += (double)o.x + o.y + o.z;
    }
    se = block_sum(se, sh); rs = block_sum(rs, sh); os = block_sum(os, sh); cnt = block_sum(cnt, sh);
    if (threadIdx.x == 0) {
        double* p = scratch + 8 + 4 * blockIdx.x;
        p[0] = se; p[1] = rs; p[2] = os; p[3] = cnt;
    }
}
// norm.comp: fold partials -> scratch[0..3] = {mse, refMean, ownMean, count}
__global__ void k_compare_norm(double* __restrict__ scratch)
{
    if (threadIdx.x != 0) return;
    double se = 0, rs = 0, os = 0, cnt = 0;
    for (int b = 0; b < CMP_BLOCKS; b++) {
        const double* p = scratch + 8 + 4 * b;
        se += p[0]; rs += p[1]; os += p[2]; cnt += p[3];
    }
    const double inv = cnt > 0 ? 1.0 / (cnt * 3.0) : 0.0;
    scratch[0] = se * inv; scratch[1] = rs * inv; scratch[2] = os * inv; scratch[3] = cnt;
}
// cmp2.comp: variance of own around ownMean
__global__ __launch_bounds__(256) void k_compare_2(const float4* __restrict__ ref, const float4* __restrict__ own, uint32_t n,
                                                  double* __restrict__ scratch)
{
    __shared__ double sh[256];
    const double mean = scratch[2];
    double var = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += CMP_BLOCKS * 256u) {
        const float4 r = ref[i], o = own[i];
        if (r.w == 0.0f) continue;
        const double dx = o.x - mean, dy = o.y - mean, dz = o.z - mean;
        var += dx * dx + dy * dy + dz * dz;
    }
    var = block_sum(var, sh);
    if (threadIdx.x == 0) scratch[8 + 4 * CMP_BLOCKS + blockIdx.x] = var;
}
__global__ void k_compare_final(const double* __restrict__ scratch, float* __restrict__ result5)
{
    if (threadIdx.x != 0) return;
    double var = 0;
    for (int b = 0; b < CMP_BLOCKS; b++) var += scratch[8 + 4 * CMP_BLOCKS + b];
    const double cnt = scratch[3];
    const double inv = cnt > 0 ? 1.0 / (cnt * 3.0) : 0.0;
    result5[0] = (float)scratch[0];
    result5[1] = (float)scratch[1];
    result5[2] = (float)scratch[2];
    result5[3] = (float)(var * inv);
    result5[4] = (float)cnt;
}

// ---- the same metrics of a frame sharded over ranks ("reduce only the metrics"): pass 1 leaves the rank's RAW sums
// {sq err, ref sum, own sum, valid count} in scratch[0..3], the caller all-reduces them, k_compare_scale turns the global sums into
// {mse, refMean, ownMean, count}; pass 2 (k_compare_2 with the GLOBAL ownMean) leaves the rank's raw variance sum in scratch[4], the
// caller all-reduces it, k_compare_final_sharded writes the Result.
__global__ void k_compare_fold(double* __restrict__ scratch)
{
    if (threadIdx.x != 0) return;
    double se = 0, rs = 0, os = 0, cnt = 0;
    for (int b = 0; b < CMP_BLOCKS; b++) {
        const double* p = scratch + 8 + 4 * b;
        se += p[0]; rs += p[1]; os += p[2]; cnt += p[3];
    }
    scratch[0] = se; scratch[1] = rs; scratch[2] = os; scratch[3] = cnt;
}
__global__ void k_compare_scale(double* __restrict__ scratch)
{
    if (threadIdx.x != 0) return;
    const double cnt = scratch[3];
    const double inv = cnt > 0 ? 1.0 / (cnt * 3.0) : 0.0;
    scratch[0] *= inv; scratch[1] *= inv; scratch[2] *= inv;
}
__global__ void k_compare_fold_var(double* __restrict__ scratch)
{
    if (threadIdx.x != 0) return;
    double var = 0;
    for (int b = 0; b < CMP_BLOCKS; b++) var += scratch[8 + 4 * CMP_BLOCKS + b];
    scratch[4] = var;
}
__global__ void k_compare_final_sharded(const double* __restrict__ scratch, float* __restrict__ result5)
{
    if (threadIdx.x != 0) return;
    const double cnt = scratch[3];
    const double inv = cnt > 0 ? 1.0 / (cnt * 3.0) : 0.0;
    result5[0] = (float)scratch[0];
    result5[1] = (float)scratch[1];
    result5[2] = (float)scratch[2];
    result5[3] = (float)(scratch[4] * inv);
    result5[4] = (float)cnt;
}

// ---- a sharded frame put together again (nrc_tile: strips of 2^block_log2 columns dealt round-robin over `world` ranks):
// gathered = [world][h][max_lw] RGBA32F, rank r's local image padded to max_lw columns; out = [h][gw]
__global__ __launch_bounds__(256) void k_assemble_columns(const float4* __restrict__ gathered, uint32_t world, uint32_t block_log2, uint32_t gw, uint32_t h,
                                                         uint32_t max_lw, float4* __restrict__ out)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    const uint32_t gx = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
    if (gx >= gw) return;
    const uint32_t strip = gx >> block_log2, rank = strip % world, lstrip = strip / world;
    const uint32_t lx = (lstrip << block_log2) + (gx & ((1u << block_log2) - 1u));
    out[(size_t)y * gw + gx] = gathered[((size_t)rank * h + y) * max_lw + lx];
}
// local image [h][lw] -> [h][max_lw] (the all-gather moves equal blocks; a rank's last strip may be short or missing)
__global__ __launch_bounds__(256) void k_pad_columns(const float4* __restrict__ local, uint32_t lw, uint32_t h, uint32_t max_lw, float4* __restrict__ padded)
{
    NRC_RAISE_WAVE_PRIORITY(16);
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
    if (x >= max_lw) return;
    padded[(size_t)y * max_lw + x] = x < lw ? local[(size_t)y * lw + x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------ test hooks
__global__ void k_test_math(int fn, const float* __restrict__ a, const float* __restrict__ b, uint32_t n,
                            float* __restrict__ out, float* __restrict__ out2)
{
    fill_log_table();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f, c = 0.0f;
    switch (fn) {
    case 0: s = nrc_logf(a[i], s_log_tab); break;
    case 1: nrc_sincosf(a[i], &s, &c); break;
    case 2: s = nrc_acosf(a[i]); break;
    case 3: s = nrc_asinf(a[i]); break;
    case 4: s = nrc_atan2f(a[i], b[i]); break;
    case 5: s = nrc_acosf_clamped(a[i]); break;
    case 6: s = a[i] / b[i]; break;
    case 7: s = sqrtf(a[i]); break;
    case 8: s = (float)(_Float16)a[i]; break;
    case 9: { const f2 r = logf2(f2{a[i], b[i]}); s = r.x; c = r.y; break; }
    case 10: s = sqrt_rn_normal(a[i]); break;
    default: break;
    }
    out[i] = s;
    if (out2) out2[i] = c;
}

__global__ void k_test_rng(float u, float v, float r0, float r1, float r2, float r3, uint32_t n, float* __restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float fr[4] = {r0, r1, r2, r3};
    float state = random2(random2(u, v), random4(fr));
    out[0] = state;
    for (uint32_t i = 0; i < n; i++) {
        state = random1(state);
        out[1 + i] = state * 1.0f;
    }
}

}  // namespace

// ================================================================================================ launchers
static dim3 pixel_grid(uint32_t w, uint32_t h) { return dim3(ceil_div(w, 16), ceil_div(h, 16)); }
// k_gen_rays / k_mc_render: one 8x8 pixel tile per wave, see pixel_of_wave_tile
static dim3 wave_tile_grid(uint32_t w, uint32_t h) { return dim3(camera_row_blocks(w) * ceil_div(h, 8)); }

static int g_host_raise_wave_priority = 1;      // (integrator_set_wave_priority_raise)

void launch_gen_rays(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, uint32_t primary_ray_length,
                     float primary_ray_prob, float* primary, float* info, float* origin, float* dir, float* infer_in,
                     unsigned long long* fetch_counter, const TrainGrid& tg, bool full_vertex_images, hipStream_t s)
{
    auto kernel = fetch_counter ? k_gen_rays<true> : k_gen_rays<false>;      // the look-up counter exists only in measurement launches
#ifdef NRC_LOOP_PROFILE
    kernel = k_gen_rays<true>;      // (the profiling build's per-pixel look-up counts, tools/lane_model.py)
#endif
    dim3 grid = wave_tile_grid(fr.w, fr.h);
    if (fr.hot_tiles != nullptr) grid.x += kHotTilesMax / CAMERA_WAVES_PER_BLOCK;
    DevFrame fa = fr;
    fa.raise_priority = (g_host_raise_wave_priority != 0 && fr.camera_priority_low == 0u) ? 1u : 0u;
    // (launch_last: the frame's start / "gen_rays done" events ride on the launch when the frame graph armed them -- nrc_common.hpp)
    launch_last(kernel, grid, dim3(64 * CAMERA_WAVES_PER_BLOCK), 0u, s, sc, cam, fa,
                primary_ray_length, primary_ray_prob, (float4*)primary, info, (float4*)origin, (float4*)dir, infer_in,
                fetch_counter, tg, full_vertex_images ? 1 : 0);
    NRC_HIP(hipGetLastError());
}

void launch_hot_tiles(const DevFrame& fr, uint32_t* hot, hipStream_t s)
{
    hipLaunchKernelGGL(k_hot_tiles, dim3(4096), dim3(64), 0, s, fr, hot);
    NRC_HIP(hipGetLastError());
}

uint32_t camera_slots(uint32_t w, uint32_t h) { return camera_row_blocks(w) * CAMERA_WAVES_PER_BLOCK * ceil_div(h, 8); }

void launch_tile_order(const uint32_t* cost, uint32_t n_slots, uint32_t* order, uint32_t w, hipStream_t s, uint32_t xcd_window, uint32_t workgroups_in_front)
{
    const uint32_t row_len = camera_row_blocks(w) * CAMERA_WAVES_PER_BLOCK;
    // (the sort's last launch carries the "order ready" event when the frame graph armed one -- nrc_common.hpp, LaunchTail)
    const bool xcd = xcd_window != 0u && CAMERA_WAVES_PER_BLOCK == 4u && n_slots % row_len == 0u;
    if (!xcd) launch_last(k_tile_order, dim3(1), dim3(1024), 0u, s, cost, n_slots, order);
    else {
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, cost, n_slots, order);
        const uint32_t S = 32u * std::min(32u, xcd_window);
        launch_last(k_tile_order_xcd, dim3((n_slots + workgroups_in_front * 4u + S - 1u) / S), dim3(S), 0u, s, order, n_slots, row_len, workgroups_in_front);
    }
    NRC_HIP(hipGetLastError());
}

uint32_t tile_mask_words(uint32_t w, uint32_t h) { return (ceil_div(w, 8) * ceil_div(h, 8) + 31u) / 32u + 1u; }

void launch_tile_mask(const float* boxes, uint32_t n_boxes, const DevProjView& pv, const DevFrame& fr, uint32_t* mask, hipStream_t s)
{
    NRC_HIP(hipMemsetAsync(mask, 0, (size_t)tile_mask_words(fr.w, fr.h) * 4, s));
    if (n_boxes == 0) return;
    hipLaunchKernelGGL(k_tile_mask, dim3(ceil_div(n_boxes, 256)), dim3(256), 0, s, boxes, n_boxes, pv, fr, mask);
    NRC_HIP(hipGetLastError());
}

void launch_flight_table(float* table, hipStream_t s)
{
    hipLaunchKernelGGL(k_flight_table, dim3(kFlightStates / 256u), dim3(256), 0, s, table);
    NRC_HIP(hipGetLastError());
}

void launch_flight_select(const float* table, float lambda, uint32_t* count_and_list, uint32_t* bits, hipStream_t s)
{
    NRC_HIP(hipMemsetAsync(count_and_list, 0, (1 + kFlightListMax) * 4, s));
    hipLaunchKernelGGL(k_flight_select, dim3(kFlightStates / 256u), dim3(256), 0, s, table, lambda, count_and_list, bits);
    NRC_HIP(hipGetLastError());
}

void launch_mc_render(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, uint32_t path_length,
                      float blend_factor, float* out_rgba, float* info, unsigned long long* fetch_counter, hipStream_t s)
{
    dim3 grid = wave_tile_grid(fr.w, fr.h);
    if (fr.hot_tiles != nullptr) grid.x += kHotTilesMax / CAMERA_WAVES_PER_BLOCK;
    DevFrame fa = fr;
    fa.raise_priority = (g_host_raise_wave_priority != 0 && fr.camera_priority_low == 0u) ? 1u : 0u;
    hipLaunchKernelGGL(fetch_counter ? k_mc_render<true> : k_mc_render<false>, grid, dim3(64 * CAMERA_WAVES_PER_BLOCK), 0, s, sc,
                       cam, fa, path_length, blend_factor, (float4*)out_rgba, info, fetch_counter);
    NRC_HIP(hipGetLastError());
}

void launch_prep_train(const DevScene& sc, const DevFrame& fr, const TrainGrid& tg, const float* info, const float* origin,
                       const float* dir, uint32_t* ring, uint32_t* scratch, float* train_in, float* train_target,
                       hipStream_t s, float* start)
{
    const uint32_t T = tg.tw * tg.th;
    hipLaunchKernelGGL(k_train_scan, dim3(1), dim3(1024), 0, s, fr, tg, info, ring, scratch);
    NRC_HIP(hipGetLastError());
    // (one 8x8 block of the train grid per wave for the reference's single-vertex targets; 32 rays per wave for long paths: k_prep_train)
    // (measured on the bench frame with train ray length 32: 64 / 32 / 16 / 8 rays per wave -> 2 309 / 2 412 / 2 367 / 2 347 Msamples/s)
    const uint32_t rpw = train_paths_are_long(tg) ? 32u : 64u;
    const dim3 grid = rpw >= 64u ? pixel_grid(tg.tw, tg.th) : dim3(ceil_div(T, 4u * rpw), 1);
    if (start == nullptr)
        hipLaunchKernelGGL(k_prep_train<0>, grid, dim3(256), 0, s, sc, fr, tg, (const float4*)origin, (const float4*)dir, (const uint32_t*)ring,
                           (const uint32_t*)scratch, train_in, train_target, rpw, (float*)nullptr);
    else      // the split frame graph: start vertices + inputs here, the trace by launch_train_trace on a stream of its own
        hipLaunchKernelGGL(k_prep_train<1>, pixel_grid(tg.tw, tg.th), dim3(256), 0, s, sc, fr, tg, (const float4*)origin, (const float4*)dir,
                           (const uint32_t*)ring, (const uint32_t*)scratch, train_in, train_target, 64u, start);
    NRC_HIP(hipGetLastError());
    launch_last(k_ring_push, dim3(ceil_div(T, 256)), dim3(256), 0u, s, fr, tg, (const float4*)origin,
                (const float4*)dir, ring, (const uint32_t*)scratch);
    NRC_HIP(hipGetLastError());
}

bool train_paths_are_long(const TrainGrid& tg) { return tg.ray_length * tg.spp >= 4u; }

// the trace of the split frame graph (k_prep_train<2>): from the start vertices launch_prep_train(..., start) left, to the training targets
void launch_train_trace(const DevScene& sc, const DevFrame& fr, const TrainGrid& tg, const float* start, float* train_target, hipStream_t s)
{
    const uint32_t T = tg.tw * tg.th;
    const uint32_t rpw = 32u;
    launch_last(k_prep_train<2>, dim3(ceil_div(T, 4u * rpw), 1), dim3(256), 0u, s, sc, fr, tg, (const float4*)nullptr, (const float4*)nullptr,
                (const uint32_t*)nullptr, (const uint32_t*)nullptr, (float*)nullptr, train_target, rpw, const_cast<float*>(start));
    NRC_HIP(hipGetLastError());
}

// this translation unit's copy of the run-time priority switch (nrc_common.hpp), and the host-side value the camera kernels' launchers
// hand over as a kernel argument (DevFrame::raise_priority)
void integrator_set_wave_priority_raise(int on)
{
    NRC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_raise_wave_priority), &on, sizeof(int)));
    g_host_raise_wave_priority = on;
}

void launch_composite(const DevFrame& fr, uint32_t show_nrc, float blend_factor, const float* primary, const float* info,
                      const float* infer_out, float* out_rgba, hipStream_t s, uint32_t* live_count_reset)
{
    launch_last(k_composite, pixel_grid(fr.w, fr.h), dim3(256), 0u, s, fr, show_nrc, blend_factor,
                (const float4*)primary, info, infer_out, (float4*)out_rgba, live_count_reset);
    NRC_HIP(hipGetLastError());
}

void launch_query_layout(const DevFrame& fr, uint32_t floats_per_query, const float* tiled, float* linear, hipStream_t s, const float* info)
{
    hipLaunchKernelGGL(k_query_layout, pixel_grid(fr.w, fr.h), dim3(256), 0, s, fr, floats_per_query, tiled, linear, info);
    NRC_HIP(hipGetLastError());
}
uint32_t query_count(uint32_t w, uint32_t h) { return ceil_div(w, 8) * ceil_div(h, 8) * 64u; }

void launch_compare(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, float* d_result5,
                    hipStream_t s)
{
    hipLaunchKernelGGL(k_compare_1, dim3(CMP_BLOCKS), dim3(256), 0, s, (const float4*)ref_rgba, (const float4*)own_rgba,
                       n_pixels, d_scratch);
    hipLaunchKernelGGL(k_compare_norm, dim3(1), dim3(64), 0, s, d_scratch);
    hipLaunchKernelGGL(k_compare_2, dim3(CMP_BLOCKS), dim3(256), 0, s, (const float4*)ref_rgba, (const float4*)own_rgba,
                       n_pixels, d_scratch);
    hipLaunchKernelGGL(k_compare_final, dim3(1), dim3(64), 0, s, (const double*)d_scratch, d_result5);
    NRC_HIP(hipGetLastError());
}

void launch_compare_sharded_1(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, hipStream_t s)
{
    hipLaunchKernelGGL(k_compare_1, dim3(CMP_BLOCKS), dim3(256), 0, s, (const float4*)ref_rgba, (const float4*)own_rgba, n_pixels, d_scratch);
    hipLaunchKernelGGL(k_compare_fold, dim3(1), dim3(64), 0, s, d_scratch);
    NRC_HIP(hipGetLastError());
}
void launch_compare_sharded_2(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, hipStream_t s)
{
    hipLaunchKernelGGL(k_compare_scale, dim3(1), dim3(64), 0, s, d_scratch);
    hipLaunchKernelGGL(k_compare_2, dim3(CMP_BLOCKS), dim3(256), 0, s, (const float4*)ref_rgba, (const float4*)own_rgba, n_pixels, d_scratch);
    hipLaunchKernelGGL(k_compare_fold_var, dim3(1), dim3(64), 0, s, d_scratch);
    NRC_HIP(hipGetLastError());
}
void launch_compare_sharded_3(double* d_scratch, float* d_result5, hipStream_t s)
{
    hipLaunchKernelGGL(k_compare_final_sharded, dim3(1), dim3(64), 0, s, (const double*)d_scratch, d_result5);
    NRC_HIP(hipGetLastError());
}
void launch_pad_columns(const float* local, uint32_t lw, uint32_t h, uint32_t max_lw, float* padded, hipStream_t s)
{
    hipLaunchKernelGGL(k_pad_columns, dim3(ceil_div(max_lw, 256), h), dim3(256), 0, s, (const float4*)local, lw, h, max_lw, (float4*)padded);
    NRC_HIP(hipGetLastError());
}
void launch_assemble_columns(const float* gathered, uint32_t world, uint32_t block_log2, uint32_t gw, uint32_t h, uint32_t max_lw, float* out,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_assemble_columns, dim3(ceil_div(gw, 256), h), dim3(256), 0, s, (const float4*)gathered, world, block_log2, gw, h, max_lw,
                       (float4*)out);
    NRC_HIP(hipGetLastError());
}

void launch_test_math(int fn, const float* a, const float* b, uint32_t n, float* out, float* out2, hipStream_t s)
{
    hipLaunchKernelGGL(k_test_math, dim3(ceil_div(n, 256)), dim3(256), 0, s, fn, a, b, n, out, out2);
    NRC_HIP(hipGetLastError());
}

void launch_test_rng(float u, float v, const float* fr, uint32_t n, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_test_rng, dim3(1), dim3(64), 0, s, u, v, fr[0], fr[1], fr[2], fr[3], n, out);
    NRC_HIP(hipGetLastError());
}

}  // namespace nrc

#ifdef NRC_LOOP_PROFILE
// profiling build only (tools/loop_profile.py): read / reset the loop counters
extern "C" int nrc_debug_wave_times(unsigned long long* out, unsigned n_waves)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nrc::g_wave_times), (size_t)n_waves * 32) == hipSuccess ? 0 : 1;
}

extern "C" int nrc_debug_loop_profile(unsigned long long* out16, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(nrc::g_loop_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nrc::g_loop_prof), z, 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(nrc::g_live_hist), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
// trips of the tracking loops by the number of walks alive (g_live_hist)
extern "C" int nrc_debug_live_hist(unsigned long long* out24)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out24, HIP_SYMBOL(nrc::g_live_hist), 24 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
