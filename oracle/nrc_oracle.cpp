/*
 * oracle/nrc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see nrc_oracle.h for the parity-pinning statement).
 *
 * CPU restatement of the reference hot path.  Every function cites the reference lines it follows;
 * paths are relative to the reference checkout (data/shader/... , src/...).
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC  (oracle/Makefile)
 */
#include "nrc_oracle.h"
#include "orc_math.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct V3 { float x, y, z; };
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 mul(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline float dot(V3 a, V3 b) { return orc_fmaf_(a.z, b.z, orc_fmaf_(a.y, b.y, a.x * b.x)); }
static inline float length(V3 a) { return sqrtf(dot(a, a)); }
static inline V3 normalize(V3 a) { float inv = 1.0f / length(a); return v3(a.x * inv, a.y * inv, a.z * inv); }
static inline V3 madd(V3 d, float t, V3 o) { return v3(orc_fmaf_(d.x, t, o.x), orc_fmaf_(d.y, t, o.y), orc_fmaf_(d.z, t, o.z)); }   /* o + d*t */

/* ------------------------------------------------------------------ RNG: include/random.glsl */
static inline uint32_t hash1(uint32_t x)               /* random.glsl:24-32 */
{
    x += (x << 10); x ^= (x >> 6); x += (x << 3); x ^= (x >> 11); x += (x << 15);
    return x;
}
static inline float float_construct(uint32_t m)        /* random.glsl:41-51 */
{
    return orc_u2f((m & 0x007fffffu) | 0x3f800000u) - 1.0f;
}
static inline float random1(float x) { return float_construct(hash1(orc_f2u(x))); }                   /* :54 */
static inline float random2(float x, float y) { return float_construct(hash1(orc_f2u(x) ^ hash1(orc_f2u(y)))); } /* :35,55 */
static inline float random4(const float* v)                                                            /* :37,57 */
{
    return float_construct(hash1(orc_f2u(v[0]) ^ hash1(orc_f2u(v[1])) ^ hash1(orc_f2u(v[2])) ^ hash1(orc_f2u(v[3]))));
}

struct Ctx {
    const orc_scene* sc;
    V3 size, half_size, inv_size;
    float len2size;          /* length(2*skySize), volume.glsl:19 */
    float inv_max_density;   /* 1/VOLUME_DENSITY_FACTOR, path_trace.glsl:26,154 */
    float fnx, fny, fnz;
    float rng;               /* randomState, random.glsl:59 */
    uint64_t fetches;
    uint16_t* trace = nullptr;   /* tests/walk_model.py: every tracking walk appends the number of free flights it drew */
    uint32_t trace_n = 0, trace_max = 0;
    void walk_done(uint32_t flights) { if (trace && trace_n < trace_max) trace[trace_n++] = (uint16_t)flights; }
    float rand(float max_val) { rng = random1(rng); return rng * max_val; }   /* random.glsl:66-70 */
};

static void ctx_init(Ctx& c, const orc_scene* sc)
{
    c.sc = sc;
    c.size = v3(sc->size[0], sc->size[1], sc->size[2]);
    c.half_size = mul(c.size, 0.5f);
    c.inv_size = v3(1.0f / c.size.x, 1.0f / c.size.y, 1.0f / c.size.z);
    c.len2size = length(mul(c.size, 2.0f));
    c.inv_max_density = 1.0f / sc->density_factor;
    c.fnx = (float)sc->nx; c.fny = (float)sc->ny; c.fnz = (float)sc->nz;
    c.rng = 0.0f;
    c.fetches = 0;
}

static inline void init_random(Ctx& c, float u, float v, const float* frame_random)   /* random.glsl:61-64 */
{
    c.rng = random2(random2(u, v), random4(frame_random));
}

/* ------------------------------------------------------------------ volume: include/volume.glsl */
static inline float sky_sdf(const Ctx& c, V3 p)          /* volume.glsl:1-5 (skyPos = 0) */
{
    V3 d = v3(fabsf(p.x) - c.half_size.x, fabsf(p.y) - c.half_size.y, fabsf(p.z) - c.half_size.z);
    V3 dm = v3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
    return length(dm) + fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f);
}

static inline void find_entry_exit(const Ctx& c, V3 ro, V3 rd, V3* entry, V3* exit_)   /* volume.glsl:7-29 */
{
    float dist;
    do {
        dist = sky_sdf(c, ro);
        ro = madd(rd, dist, ro);
    } while (dist > 0.125f && dist < 100000.0f);
    *entry = ro;
    ro = madd(rd, c.len2size, ro);
    rd = neg(rd);
    do {
        dist = sky_sdf(c, ro);
        ro = madd(rd, dist, ro);
    } while (dist > 0.125f && dist < 100000.0f);
    *exit_ = ro;
}

/* volume.glsl:31-39 + sampler of src/Texture3D.cpp:79-81,106,221: R8 UNORM, NEAREST, border 0.
 * uvw = pos/size + 0.5 is evaluated as pos*(1/size) + 0.5 (DESIGN.md "math spec"). */
static inline float get_density(Ctx& c, V3 p)
{
    float u = orc_fmaf_(p.x, c.inv_size.x, 0.5f);
    float v = orc_fmaf_(p.y, c.inv_size.y, 0.5f);
    float w = orc_fmaf_(p.z, c.inv_size.z, 0.5f);
    float fx = u * c.fnx, fy = v * c.fny, fz = w * c.fnz;
    c.fetches++;
    if (!(fx >= 0.0f && fx < c.fnx && fy >= 0.0f && fy < c.fny && fz >= 0.0f && fz < c.fnz)) return 0.0f;
    uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy, iz = (uint32_t)fz;
    uint8_t t = c.sc->density[(size_t)ix + (size_t)c.sc->nx * ((size_t)iy + (size_t)c.sc->ny * (size_t)iz)];
    return c.sc->density_factor * ((float)t * (1.0f / 255.0f));
}

/* ------------------------------------------------------------------ dir_gen.glsl */
static inline float hg_phase(const Ctx& c, float cos_theta)      /* dir_gen.glsl:1-7; pow(x,1.5) == x*sqrt(x) */
{
    float g = c.sc->g;
    float g2 = g * g;
    float x = orc_fmaf_(-(2.0f * g), cos_theta, 1.0f + g2);
    return (0.5f * (1.0f - g2)) / (x * sqrtf(x));
}

/* rotationMatrix(axis, angle) * vec4(v, 1) with GLSL column-major mat4 construction (dir_gen.glsl:9-20,55-56) */
static inline V3 rotate(V3 axis, float angle, V3 v)
{
    axis = normalize(axis);
    float s, co;
    orc_sincosf(angle, &s, &co);
    float oc = 1.0f - co;
    /* columns of the mat4 as written in the shader */
    const float ox = oc * axis.x, oy = oc * axis.y, oz = oc * axis.z;
    V3 c0 = v3(orc_fmaf_(ox, axis.x, co), orc_fmaf_(ox, axis.y, -(axis.z * s)), orc_fmaf_(oz, axis.x, axis.y * s));
    V3 c1 = v3(orc_fmaf_(ox, axis.y, axis.z * s), orc_fmaf_(oy, axis.y, co), orc_fmaf_(oy, axis.z, -(axis.x * s)));
    V3 c2 = v3(orc_fmaf_(oz, axis.x, -(axis.y * s)), orc_fmaf_(oy, axis.z, axis.x * s), orc_fmaf_(oz, axis.z, co));
    return v3(orc_fmaf_(c2.x, v.z, orc_fmaf_(c1.x, v.y, c0.x * v.x)),
              orc_fmaf_(c2.y, v.z, orc_fmaf_(c1.y, v.y, c0.y * v.x)),
              orc_fmaf_(c2.z, v.z, orc_fmaf_(c1.z, v.y, c0.z * v.x)));
}

static inline V3 new_ray_dir(Ctx& c, V3 old_dir, bool phase_sampling)    /* dir_gen.glsl:22-64 */
{
    old_dir = normalize(old_dir);
    V3 ortho = old_dir.z < old_dir.x ? v3(old_dir.y, -old_dir.x, 0.0f) : v3(0.0f, -old_dir.z, old_dir.y);
    /* robustness: the shader's pick is the zero vector for dir == (-1,0,0) or (0,0,-1)
     * (normalize -> NaN; the centre pixel of an exact camera hits it); use +y there */
    if (ortho.x == 0.0f && ortho.y == 0.0f && ortho.z == 0.0f) ortho = v3(0.0f, 1.0f, 0.0f);
    ortho = normalize(ortho);
    float angle;
    if (phase_sampling) {
        float g = c.sc->g;
        float cos_theta;
        if (fabsf(g) < 0.001f) {
            cos_theta = 1.0f - 2.0f * c.rand(1.0f);
        } else {
            float sqr_term = (1.0f - g * g) / orc_fmaf_(2.0f * g, c.rand(1.0f), 1.0f - g);
            cos_theta = orc_fmaf_(-sqr_term, sqr_term, 1.0f + g * g) / (2.0f * g);
        }
        angle = orc_acosf_clamped(cos_theta);
    } else {
        angle = c.rand(ORC_PI);
    }
    V3 nd = rotate(ortho, angle, old_dir);
    angle = c.rand(ORC_TWO_PI);
    nd = rotate(old_dir, angle, nd);
    return normalize(nd);
}

/* ------------------------------------------------------------------ path_trace.glsl */
static inline float ratio_track(Ctx& c, V3 start, V3 end)       /* path_trace.glsl:24-43 */
{
    V3 d = sub(end, start);
    V3 dir = normalize(d);
    float t_max = length(d);
    float tr = 1.0f, t = 0.0f;
    uint32_t i = 0, flights = 0;
    for (; i < 128; i++) {
        t = orc_fmaf_(-orc_logf(1.0f - c.rand(1.0f)), c.inv_max_density, t);
        flights++;
        if (t >= t_max) break;
        V3 p = madd(dir, t, start);
        tr *= orc_fmaf_(-get_density(c, p), c.inv_max_density, 1.0f);
    }
    c.walk_done(flights);
    return tr;
}

static inline V3 trace_dir_light(Ctx& c, V3 pos, V3 dir)        /* path_trace.glsl:45-56 */
{
    const orc_scene* s = c.sc;
    if (s->dir_light_strength == 0.0f) return v3(0, 0, 0);
    V3 ld = v3(s->dir_light_dir[0], s->dir_light_dir[1], s->dir_light_dir[2]);
    V3 en, ex;
    find_entry_exit(c, pos, neg(normalize(ld)), &en, &ex);
    float tr = ratio_track(c, pos, ex);
    float phase = hg_phase(c, dot(ld, neg(dir)));
    float l = (1.0f * tr) * s->dir_light_strength * phase;
    return v3(l, l, l);
}

static inline V3 trace_point_light(Ctx& c, V3 pos, V3 dir)      /* path_trace.glsl:58-69 */
{
    const orc_scene* s = c.sc;
    if (s->point_light_strength == 0.0f) return v3(0, 0, 0);
    V3 lp = v3(s->point_light_pos[0], s->point_light_pos[1], s->point_light_pos[2]);
    float tr = ratio_track(c, lp, pos);
    float phase = hg_phase(c, dot(normalize(sub(lp, pos)), neg(dir)));
    return v3(((s->point_light_color[0] * s->point_light_strength) * tr) * phase,
              ((s->point_light_color[1] * s->point_light_strength) * tr) * phase,
              ((s->point_light_color[2] * s->point_light_strength) * tr) * phase);
}

/* texture(hdrEnvMap, uv).xyz * HDR_ENV_MAP_STRENGTH: RGBA32F, LINEAR, clamp-to-edge
 * (path_trace.glsl:71-80, src/HdrEnvMap.cpp:109-110,235); fp32 bilinear weights */
static inline V3 env_lookup(const Ctx& c, float u, float v)
{
    const orc_scene* s = c.sc;
    if (s->env == nullptr || s->env_w == 0) return v3(0, 0, 0);
    float fx = u * (float)s->env_w - 0.5f, fy = v * (float)s->env_h - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float wx = fx - flx, wy = fy - fly;
    int x0 = (int)flx, y0 = (int)fly, x1 = x0 + 1, y1 = y0 + 1;
    int mw = (int)s->env_w - 1, mh = (int)s->env_h - 1;
    x0 = std::min(std::max(x0, 0), mw); x1 = std::min(std::max(x1, 0), mw);
    y0 = std::min(std::max(y0, 0), mh); y1 = std::min(std::max(y1, 0), mh);
    const float* p00 = s->env + 4 * ((size_t)y0 * s->env_w + x0);
    const float* p10 = s->env + 4 * ((size_t)y0 * s->env_w + x1);
    const float* p01 = s->env + 4 * ((size_t)y1 * s->env_w + x0);
    const float* p11 = s->env + 4 * ((size_t)y1 * s->env_w + x1);
    float r[3];
    for (int k = 0; k < 3; k++) {
        float a = orc_fmaf_(wx, p10[k] - p00[k], p00[k]);
        float b = orc_fmaf_(wx, p11[k] - p01[k], p01[k]);
        r[k] = orc_fmaf_(wy, b - a, a) * s->env_strength;
    }
    return v3(r[0], r[1], r[2]);
}

static inline V3 sample_env_dir(const Ctx& c, V3 dir)           /* path_trace.glsl:71-86 */
{
    float phi = orc_atan2f(dir.z, dir.x);
    float theta = orc_asinf(dir.y);
    return env_lookup(c, orc_fmaf_(phi, 0.1591f, 0.5f), orc_fmaf_(theta, 0.3183f, 0.5f));
}

static inline V3 sample_env(Ctx& c, V3 pos, V3 dir)             /* path_trace.glsl:88-131, sampleCount 1 */
{
    if (c.sc->env_strength == 0.0f) return v3(0, 0, 0);
    V3 rdir = new_ray_dir(c, dir, false);
    float phase = hg_phase(c, dot(rdir, neg(dir)));
    V3 en, ex;
    find_entry_exit(c, pos, rdir, &en, &ex);
    float tr = ratio_track(c, pos, ex);
    V3 e = sample_env_dir(c, rdir);
    return v3((e.x * phase) * tr, (e.y * phase) * tr, (e.z * phase) * tr);   /* "/ float(1)" is exact */
}

static inline V3 trace_scene(Ctx& c, V3 pos, V3 dir)            /* path_trace.glsl:133-137; operands left to right */
{
    V3 a = trace_dir_light(c, pos, dir);
    V3 b = trace_point_light(c, pos, dir);
    V3 e = sample_env(c, pos, dir);
    return add(add(a, b), e);
}

static inline V3 delta_track(Ctx& c, V3 ro, V3 rd, bool* volume_exit)    /* path_trace.glsl:150-174 */
{
    *volume_exit = false;
    V3 en, ex;
    find_entry_exit(c, ro, rd, &en, &ex);
    float t_max = length(sub(ex, ro));
    float t = 0.0f;
    uint32_t flights = 0;
    for (uint32_t i = 0; i < 128; i++) {
        t = orc_fmaf_(-orc_logf(1.0f - c.rand(1.0f)), c.inv_max_density, t);
        flights++;
        if (t >= t_max) { *volume_exit = true; break; }
        V3 p = madd(rd, t, ro);
        if (get_density(c, p) * c.inv_max_density > c.rand(1.0f)) { c.walk_done(flights); return p; }
    }
    c.walk_done(flights);
    return madd(rd, c.rand(t_max), ro);
}

/* ------------------------------------------------------------------ camera ray (mc/render.comp:42-60, nrc/gen_rays.comp:53-72) */
static inline void camera_ray(const orc_camera* cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                              float* u_out, float* v_out, V3* ro, V3* rd)
{
    float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
    float u = (float)x * inv_w, v = (float)y * inv_h;
    float sx = orc_fmaf_(u, 2.0f, -1.0f), sy = orc_fmaf_(v, 2.0f, -1.0f);
    const float* m = cam->inv_proj_view;      /* column-major: m[4*col+row]; screen = (sx, sy, 0, 1) */
    float wx = orc_fmaf_(m[4], sy, orc_fmaf_(m[0], sx, m[12]));
    float wy = orc_fmaf_(m[5], sy, orc_fmaf_(m[1], sx, m[13]));
    float wz = orc_fmaf_(m[6], sy, orc_fmaf_(m[2], sx, m[14]));
    float ww = orc_fmaf_(m[7], sy, orc_fmaf_(m[3], sx, m[15]));
    V3 p = v3(wx / ww, wy / ww, wz / ww);
    *ro = v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    *rd = normalize(sub(p, *ro));
    *u_out = u; *v_out = v;
}

/* StoreNrcInferInput / StoreNrcTrainData normalisation (prep_infer_rays.comp:7-24, prep_train_rays.comp:40-54; quirks Q3-Q5) */
static inline void nrc_query(const Ctx& c, V3 pos, V3 dir, float* q)
{
    q[0] = pos.x / c.size.x + c.size.x / 2.0f;
    q[1] = pos.y / c.size.y + c.size.y / 2.0f;
    q[2] = pos.z / c.size.z + c.size.z / 2.0f;
    float theta = orc_atan2f(dir.z, dir.x);
    q[3] = theta / ORC_PI + 0.5f;
    float lxz = sqrtf(dir.x * dir.x + dir.z * dir.z);
    float phi = orc_acosf(dir.y / lxz);
    q[4] = phi / ORC_PI;
}

template <class F>
static void parallel_rows(uint32_t y0, uint32_t y1, int n_threads, F f)
{
    if (n_threads <= 1 || y1 - y0 <= 1) { for (uint32_t y = y0; y < y1; y++) f(y, 0); return; }
    std::atomic<uint32_t> next(y0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t]() { for (;;) { uint32_t y = next.fetch_add(1); if (y >= y1) break; f(y, t); } });
    for (auto& t : th) t.join();
}

} // namespace

/* ====================================================================== C ABI */
extern "C" {

uint32_t orc_hash(uint32_t x) { return hash1(x); }
float orc_random1(float x) { return random1(x); }

void orc_rng_kat(float u, float v, const float frame_random[4], int n, float* out)
{
    Ctx c; c.rng = 0; c.fetches = 0; c.sc = nullptr;
    init_random(c, u, v, frame_random);
    out[0] = c.rng;
    for (int i = 0; i < n; i++) out[1 + i] = c.rand(1.0f);
}

void orc_math_eval(int fn, const float* a, const float* b, int n, float* out, float* out2)
{
    for (int i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = orc_logf(a[i]); break;
        case 1: orc_sincosf(a[i], &out[i], &out2[i]); break;
        case 2: out[i] = orc_acosf(a[i]); break;
        case 3: out[i] = orc_asinf(a[i]); break;
        case 4: out[i] = orc_atan2f(a[i], b[i]); break;
        case 5: out[i] = orc_acosf_clamped(a[i]); break;
        case 6: out[i] = a[i] / b[i]; break;
        case 7: out[i] = sqrtf(a[i]); break;
        case 8: out[i] = orc_round_f16(a[i]); break;
        default: out[i] = 0.0f;
        }
    }
}

/* ---------------------------------------------------------------- mc/render.comp:7-84 */
void orc_mc_render(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H,
                   uint32_t y0, uint32_t y1, uint32_t path_length, const float frame_random[4],
                   float blend_factor, float* out_rgba, float* info, int n_threads, uint64_t* n_fetch)
{
    std::vector<uint64_t> fetch((size_t)std::max(n_threads, 1), 0);
    parallel_rows(y0, y1, n_threads, [&](uint32_t y, int tid) {
        Ctx c; ctx_init(c, sc);
        for (uint32_t x = 0; x < W; x++) {
            float u, v; V3 ro, rd;
            camera_ray(cam, W, H, x, y, &u, &v, &ro, &rd);
            init_random(c, u, v, frame_random);
            V3 entry, ex;
            find_entry_exit(c, ro, rd, &entry, &ex);
            V3 col; bool did_scatter = false;
            if (sky_sdf(c, entry) > 100000.0f) {
                col = sample_env_dir(c, rd);
            } else {
                /* TracePath, mc/render.comp:7-40 */
                V3 light = v3(0, 0, 0);
                V3 e2, x2;
                find_entry_exit(c, ro, rd, &e2, &x2);
                V3 cur = e2, dir = rd;
                float factor = 1.0f;
                bool vexit = false;
                for (uint32_t i = 0; i < path_length; i++) {
                    cur = delta_track(c, cur, dir, &vexit);
                    if (vexit) break;
                    did_scatter = true;
                    factor *= 0.5f;
                    V3 l = mul(trace_scene(c, cur, dir), factor);
                    light = add(light, l);
                    dir = new_ray_dir(c, dir, true);
                }
                col = light;
                if (!did_scatter) col = sample_env_dir(c, rd);
            }
            float a = did_scatter ? 1.0f : 0.0f;
            float* o = out_rgba + 4 * ((size_t)y * W + x);
            float ib = 1.0f - blend_factor;
            o[0] = blend_factor * col.x + ib * o[0];
            o[1] = blend_factor * col.y + ib * o[1];
            o[2] = blend_factor * col.z + ib * o[2];
            o[3] = blend_factor * a + ib * o[3];
            if (info) info[(size_t)y * W + x] = a;
        }
        fetch[tid] += c.fetches;
    });
    if (n_fetch) { uint64_t s = 0; for (auto f : fetch) s += f; *n_fetch = s; }
}

/* ---------------------------------------------------------------- nrc/gen_rays.comp:7-101 + nrc/prep_infer_rays.comp:26-46 */
static void nrc_gen_rays_impl(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H,
                      uint32_t y0, uint32_t y1, uint32_t primary_ray_length, float primary_ray_prob,
                      const float frame_random[4], float* primary_rgba, float* info,
                      float* nrc_origin, float* nrc_dir, float* infer_input,
                      int n_threads, uint64_t* n_fetch, uint16_t* walk_lengths, uint32_t walks_per_pixel)
{
    std::vector<uint64_t> fetch((size_t)std::max(n_threads, 1), 0);
    parallel_rows(y0, y1, n_threads, [&](uint32_t y, int tid) {
        Ctx c; ctx_init(c, sc);
        for (uint32_t x = 0; x < W; x++) {
            float u, v; V3 ro, rd;
            camera_ray(cam, W, H, x, y, &u, &v, &ro, &rd);
            init_random(c, u, v, frame_random);
            if (walk_lengths) { c.trace = walk_lengths + ((size_t)y * W + x) * walks_per_pixel; c.trace_n = 0; c.trace_max = walks_per_pixel; }
            V3 entry, ex;
            find_entry_exit(c, ro, rd, &entry, &ex);
            size_t pix = (size_t)y * W + x;
            V3 col; float thr = 1.0f; bool did_scatter = false;
            if (sky_sdf(c, entry) > 100000.0f) {
                col = sample_env_dir(c, rd);
                /* origin/dir images keep their previous content (never written on this branch) */
            } else {
                V3 light = v3(0, 0, 0);
                V3 e2, x2;
                find_entry_exit(c, ro, rd, &e2, &x2);
                V3 cur = e2, dir = rd;
                float factor = 1.0f;
                bool vexit = false;
                for (int i = 0;; i++) {                                   /* gen_rays.comp:21-43 */
                    cur = delta_track(c, cur, dir, &vexit);
                    if (vexit) break;
                    did_scatter = true;
                    factor *= 0.5f;
                    V3 l = mul(trace_scene(c, cur, dir), factor);
                    light = add(light, l);
                    dir = new_ray_dir(c, dir, true);
                    if ((uint32_t)i >= primary_ray_length) {
                        if (c.rand(1.0f) >= primary_ray_prob || i == 128) break;
                    }
                }
                float* po = nrc_origin + 4 * pix; po[0] = cur.x; po[1] = cur.y; po[2] = cur.z; po[3] = 0.0f;
                float* pd = nrc_dir + 4 * pix;    pd[0] = dir.x; pd[1] = dir.y; pd[2] = dir.z; pd[3] = 0.0f;
                col = light; thr = factor;
                if (!did_scatter) { col = sample_env_dir(c, rd); thr = 1.0f; }
                if (did_scatter && infer_input) {
                    nrc_query(c, cur, dir, infer_input + 5 * ((size_t)x * H + y));
                }
            }
            float* pc = primary_rgba + 4 * pix; pc[0] = col.x; pc[1] = col.y; pc[2] = col.z; pc[3] = thr;
            info[pix] = did_scatter ? 1.0f : 0.0f;
            if (!did_scatter && infer_input) {
                float* q = infer_input + 5 * ((size_t)x * H + y);
                q[0] = q[1] = q[2] = q[3] = q[4] = 0.0f;            /* vkCmdFillBuffer(0), NrcHpmRenderer.cu:1996 */
            }
        }
        fetch[tid] += c.fetches;
    });
    if (n_fetch) { uint64_t s = 0; for (auto f : fetch) s += f; *n_fetch = s; }
}

void orc_nrc_gen_rays(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H,
                      uint32_t y0, uint32_t y1, uint32_t primary_ray_length, float primary_ray_prob,
                      const float frame_random[4], float* primary_rgba, float* info,
                      float* nrc_origin, float* nrc_dir, float* infer_input,
                      int n_threads, uint64_t* n_fetch)
{
    nrc_gen_rays_impl(sc, cam, W, H, y0, y1, primary_ray_length, primary_ray_prob, frame_random, primary_rgba, info, nrc_origin, nrc_dir,
                      infer_input, n_threads, n_fetch, nullptr, 0);
}

/* the same frame; walk_lengths[(y * W + x) * walks_per_pixel + k] = free flights drawn by the pixel's k-th tracking walk, in program
 * order (delta, dir light, [point light,] environment, delta, ...), 0 beyond its last walk (caller zero-fills).  tests/walk_model.py */
void orc_nrc_walk_lengths(const orc_scene* sc, const orc_camera* cam, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
                          uint32_t primary_ray_length, float primary_ray_prob, const float frame_random[4], float* primary_rgba,
                          float* info, float* nrc_origin, float* nrc_dir, int n_threads, uint16_t* walk_lengths, uint32_t walks_per_pixel)
{
    nrc_gen_rays_impl(sc, cam, W, H, y0, y1, primary_ray_length, primary_ray_prob, frame_random, primary_rgba, info, nrc_origin, nrc_dir,
                      nullptr, n_threads, nullptr, walk_lengths, walks_per_pixel);
}

/* ---------------------------------------------------------------- nrc/clear.comp:5-9 + nrc/prep_train_rays.comp:7-138
 * Ring semantics made deterministic (the reference's atomics race inside one dispatch):
 *   pops happen in linear train-index order and read the ring as it was at frame start;
 *   pushes happen in linear train-index order after all pops. */
void orc_nrc_prep_train(const orc_scene* sc, uint32_t W, uint32_t H, uint32_t TW, uint32_t TH,
                        uint32_t x_dist, uint32_t y_dist, uint32_t train_spp, uint32_t train_ray_length,
                        uint32_t ring_size, const float frame_random[4],
                        const float* info, const float* nrc_origin, const float* nrc_dir,
                        uint32_t* ring_head_tail, float* ring,
                        float* train_input, float* train_target, int n_threads)
{
    const uint32_t T = TW * TH;
    if (ring_size > 0) { ring_head_tail[0] %= ring_size; ring_head_tail[1] %= ring_size; }  /* clear.comp */
    uint32_t head = ring_head_tail[0], tail = ring_head_tail[1];
    std::vector<uint8_t> scat(T);
    std::vector<uint32_t> pop_idx(T), push_idx(T);
    uint32_t n_pop = 0, n_push = 0;
    for (uint32_t i = 0; i < T; i++) {
        uint32_t tx = i % TW, ty = i / TW;
        uint32_t rx = tx * x_dist, ry = ty * y_dist;
        bool s = (rx < W && ry < H) ? (info[(size_t)ry * W + rx] == 1.0f) : false;   /* OOB imageLoad -> 0 */
        scat[i] = s;
        if (s) push_idx[i] = n_push++; else pop_idx[i] = n_pop++;
    }
    std::vector<float> ring_snapshot;
    if (ring_size > 0) ring_snapshot.assign(ring, ring + 6 * (size_t)ring_size);
    float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
    parallel_rows(0, TH, n_threads, [&](uint32_t ty, int) {
        Ctx c; ctx_init(c, sc);
        for (uint32_t tx = 0; tx < TW; tx++) {
            uint32_t i = ty * TW + tx;
            init_random(c, (float)tx * inv_w, (float)ty * inv_h, frame_random);     /* :108,111 (Q6) */
            V3 ro = v3(0, 0, 0);
            V3 rdir = normalize(v3(1.0f, 1.0f, 1.0f));
            if (scat[i]) {
                size_t p = (size_t)(ty * y_dist) * W + tx * x_dist;
                ro = v3(nrc_origin[4 * p], nrc_origin[4 * p + 1], nrc_origin[4 * p + 2]);
                rdir = v3(nrc_dir[4 * p], nrc_dir[4 * p + 1], nrc_dir[4 * p + 2]);
            } else if (ring_size > 0) {
                const float* r = ring_snapshot.data() + 6 * (size_t)((tail + pop_idx[i]) % ring_size);
                ro = v3(r[0], r[1], r[2]); rdir = v3(r[3], r[4], r[5]);
            }
            V3 target = v3(0, 0, 0);
            for (uint32_t s = 0; s < train_spp; s++) {                 /* TracePath, :77-99 */
                V3 light = v3(0, 0, 0);
                V3 en, ex;
                find_entry_exit(c, ro, rdir, &en, &ex);
                V3 cur = en, dir = rdir;
                float factor = 1.0f;
                bool vexit = false;
                for (uint32_t k = 0; k < train_ray_length; k++) {
                    cur = delta_track(c, cur, dir, &vexit);
                    if (vexit) break;
                    factor *= 0.5f;
                    light = add(light, mul(trace_scene(c, cur, dir), factor));
                    dir = new_ray_dir(c, dir, true);
                }
                target = add(target, light);
            }
            float fs = (float)train_spp;
            target = v3(target.x / fs, target.y / fs, target.z / fs);
            if (ring_size > 0) {                                         /* StoreNrcTrainData, :33-75 */
                nrc_query(c, ro, rdir, train_input + 5 * (size_t)i);
                train_target[3 * (size_t)i + 0] = fminf(8.0f, target.x);
                train_target[3 * (size_t)i + 1] = fminf(8.0f, target.y);
                train_target[3 * (size_t)i + 2] = fminf(8.0f, target.z);
                if (scat[i]) {
                    float* r = ring + 6 * (size_t)((head + push_idx[i]) % ring_size);
                    r[0] = ro.x; r[1] = ro.y; r[2] = ro.z; r[3] = rdir.x; r[4] = rdir.y; r[5] = rdir.z;
                }
            }
        }
    });
    if (ring_size > 0) {
        ring_head_tail[0] = head + n_push;     /* atomicAdd leaves the un-wrapped counters; clear.comp wraps next frame */
        ring_head_tail[1] = tail + n_pop;
    }
}

/* ---------------------------------------------------------------- nrc/render.comp:7-41 */
void orc_nrc_composite(uint32_t W, uint32_t H, uint32_t show_nrc, float blend_factor,
                       const float* primary_rgba, const float* info, const float* infer_output,
                       float* out_rgba)
{
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            size_t pix = (size_t)y * W + x, lin = (size_t)x * H + y;
            const float* p = primary_rgba + 4 * pix;
            float c[4] = {p[0], p[1], p[2], 1.0f};
            if (show_nrc == 1 && info[pix] == 1.0f)
                for (int k = 0; k < 3; k++) c[k] += fmaxf(0.0f, infer_output[3 * lin + k]) * p[3];
            float* o = out_rgba + 4 * pix;
            float ib = 1.0f - blend_factor;
            for (int k = 0; k < 4; k++) o[k] = blend_factor * c[k] + ib * o[k];
        }
}

/* ---------------------------------------------------------------- ref/cmp1.comp:23-41, norm.comp:17-23, cmp2.comp:23-38
 * deterministic (double) sums instead of float atomics */
void orc_compare(const float* ref_rgba, const float* own_rgba, uint32_t W, uint32_t H, float* r5)
{
    double mse = 0, ref_mean = 0, own_mean = 0; double n = 0;
    size_t N = (size_t)W * H;
    for (size_t i = 0; i < N; i++) {
        const float* r = ref_rgba + 4 * i; const float* o = own_rgba + 4 * i;
        if (r[3] == 0.0f) continue;
        n += 1;
        for (int k = 0; k < 3; k++) {
            double d = (double)o[k] - (double)r[k];
            mse += d * d; ref_mean += r[k]; own_mean += o[k];
        }
    }
    double inv = n > 0 ? 1.0 / (n * 3.0) : 0.0;
    mse *= inv; ref_mean *= inv; own_mean *= inv;
    double var = 0;
    for (size_t i = 0; i < N; i++) {
        const float* r = ref_rgba + 4 * i; const float* o = own_rgba + 4 * i;
        if (r[3] == 0.0f) continue;
        for (int k = 0; k < 3; k++) { double d = (double)o[k] - own_mean; var += d * d; }
    }
    var *= inv;
    r5[0] = (float)mse; r5[1] = (float)ref_mean; r5[2] = (float)own_mean; r5[3] = (float)var; r5[4] = (float)n;
}

} // extern "C"

/* ====================================================================== neural radiance cache arithmetic */
namespace {

struct Pcg32 {     /* O'Neill's pcg32 (XSH-RR) as tiny-cuda-nn vendors it (pcg32.h): the Trainer seeds pcg32{1337}, i.e. initseq = 1 */
    uint64_t state, inc;
    void seed(uint64_t init_state, uint64_t init_seq = 1u) { state = 0; inc = (init_seq << 1) | 1u; next(); state += init_state; next(); }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    /* pcg32::next_float: the MTGP trick -- a float in [1, 2) from the top 23 bits, minus 1 */
    float nextf() { uint32_t u = (next() >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f; }
};

struct Layer { uint32_t out, in; size_t off; };

/* HashGrid (AppConfig posID 0, src/AppConfig.cpp:19-27): 16 levels x 2 features, 2^19 entries per hashed level,
 * base resolution 16, per-level scale 2.0.  tiny-cuda-nn v1.6 grid encoding semantics (recalled; PARITY UNPINNED). */
constexpr uint32_t HG_LEVELS = 16, HG_FEATS = 2, HG_BASE_RES = 16;
struct NN {
    orc_nn_config cfg;
    uint32_t enc_raw, enc_dims;            /* enc_dims = padded to a multiple of 16 with 1.0 */
    std::vector<Layer> layers;             /* depth hidden matrices + output */
    std::vector<float> w, ema, m, v, grad; /* MLP parameters, then (posID 0) the hash-grid table */
    size_t n_mlp = 0;                      /* number of matrix (MLP) parameters */
    uint32_t hg_off[HG_LEVELS + 1] = {0};  /* per-level entry offsets */
    std::vector<float> dead_rows;          /* rows 3..15 of tiny-cuda-nn's padded 16 x width output matrix (initial values; never read by the model) */
    uint32_t step;
};

static uint32_t pos_dims(uint32_t id) { return id == 0 ? HG_LEVELS * HG_FEATS : id == 1 ? 3 : id == 2 ? 36 : id == 3 ? 72 : 0; }
static uint32_t dir_dims(uint32_t id) { return id == 0 ? 8 : id == 1 ? 2 : id == 2 ? 8 : 0; }

/* tiny-cuda-nn one_blob: quartic kernel CDF */
static inline float quartic_cdf(float x, float inv_radius)
{
    float u = x * inv_radius;
    float u2 = u * u;
    float u4 = u2 * u2;
    float p = (15.0f / 16.0f) * u * ((1.0f - (2.0f / 3.0f) * u2) + (1.0f / 5.0f) * u4) + 0.5f;
    /* fminf(fmaxf(p,0),1): NaN -> 0 */
    if (!(p > 0.0f)) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    return p;
}

static inline float hg_scale(uint32_t level) { return exp2f((float)level * 1.0f) * (float)HG_BASE_RES - 1.0f; }   /* per_level_scale 2 */
static inline uint32_t hg_resolution(float scale) { return (uint32_t)ceilf(scale) + 1u; }
static inline uint32_t hg_index(uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg)
{
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < 3 && stride <= hashmap_size; d++) { index += pg[d] * stride; stride *= resolution; }
    if (hashmap_size < stride) index = (pg[0] * 1u) ^ (pg[1] * 2654435761u) ^ (pg[2] * 805459861u);   /* coherent prime hash */
    return index % hashmap_size;
}
/* per level: the 8 corner entry indices (into the whole table) and trilinear weights of a position */
static inline void hg_corners(const NN& nn, uint32_t level, const float* x, uint32_t* idx8, float* w8)
{
    const float scale = hg_scale(level);
    const uint32_t res = hg_resolution(scale);
    const uint32_t hsize = nn.hg_off[level + 1] - nn.hg_off[level];
    float pos[3]; uint32_t pg[3];
    for (int d = 0; d < 3; d++) {
        pos[d] = fmaf(scale, x[d], 0.5f);
        float tmp = floorf(pos[d]);
        pg[d] = (uint32_t)(int)tmp;
        pos[d] -= tmp;
    }
    for (uint32_t c = 0; c < 8; c++) {
        float w = 1.0f; uint32_t pl[3];
        for (int d = 0; d < 3; d++) {
            if ((c & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1u; }
        }
        idx8[c] = nn.hg_off[level] + hg_index(hsize, res, pl);
        w8[c] = w;
    }
}

static void encode_one(const NN& nn, const float* in, float* out, const float* table = nullptr, bool table_fp16 = true)     /* out: enc_dims floats, fp16-rounded */
{
    uint32_t o = 0;
    const uint32_t pid = nn.cfg.pos_id, did = nn.cfg.dir_id;
    if (pid == 0) {            /* HashGrid: trilinear interpolation of 2 features per level */
        for (uint32_t l = 0; l < HG_LEVELS; l++) {
            uint32_t idx[8]; float w8[8];
            hg_corners(nn, l, in, idx, w8);
            float r[2] = {0.0f, 0.0f};
            for (int c = 0; c < 8; c++)
                for (int f = 0; f < 2; f++) {
                    float v = table[(size_t)idx[c] * 2 + f];
                    if (table_fp16) v = orc_round_f16(v);
                    r[f] = fmaf(w8[c], v, r[f]);
                }
            out[o++] = r[0]; out[o++] = r[1];
        }
    }
    /* position: input dims 0..2 (Composite: nested encodings consume dims in order, AppConfig.cpp:82-86) */
    if (pid == 3) {            /* Frequency n=12: sin(2^f*pi*x + s*pi/2), argument reduced exactly (Q3) */
        for (int d = 0; d < 3; d++)
            for (int f = 0; f < 12; f++) {
                float t = ldexpf(in[d], f);                 /* exact */
                float r = t - 2.0f * floorf(t * 0.5f);      /* exact: t mod 2 in [0,2) */
                double a = (double)r * 3.14159265358979323846;
                out[o++] = (float)sin(a);
                out[o++] = (float)cos(a);
            }
    } else if (pid == 1) {     /* Identity */
        for (int d = 0; d < 3; d++) out[o++] = in[d];
    } else if (pid == 2) {     /* TriangleWave n=12 */
        for (int d = 0; d < 3; d++)
            for (int f = 0; f < 12; f++) {
                float t = ldexpf(in[d], f);
                float r = t - 2.0f * floorf(t * 0.5f);
                out[o++] = fabsf(r - 1.0f);
            }
    }
    /* direction: input dims 3..4 */
    if (did == 0) {            /* OneBlob n_bins=4 */
        for (int d = 3; d < 5; d++) {
            float x = in[d];
            float cdf[5];
            for (int k = 0; k < 5; k++) {
                float b = (float)k * 0.25f;
                cdf[k] = (quartic_cdf(b - x, 4.0f) + quartic_cdf(b - x - 1.0f, 4.0f)) + quartic_cdf(b - x + 1.0f, 4.0f);
            }
            /* right edge of the last bin = left edge of bin 0, plus 1 (periodic wrap) */
            for (int k = 0; k < 4; k++) {
                float right = (k == 3) ? cdf[0] + 1.0f : cdf[k + 1];
                out[o++] = right - cdf[k];
            }
        }
    } else if (did == 1) {
        for (int d = 3; d < 5; d++) out[o++] = in[d];
    } else if (did == 2) {     /* TriangleWave n=4 */
        for (int d = 3; d < 5; d++)
            for (int f = 0; f < 4; f++) {
                float t = ldexpf(in[d], f);
                float r = t - 2.0f * floorf(t * 0.5f);
                out[o++] = fabsf(r - 1.0f);
            }
    }
    while (o < nn.enc_dims) out[o++] = 1.0f;
    for (uint32_t i = 0; i < nn.enc_dims; i++) out[i] = orc_round_f16(out[i]);
}

/* forward for one sample; acts[l] = activation after hidden layer l (post-ReLU); returns y[3] */
static void forward_one(const NN& nn, const std::vector<float>& wq, const float* enc, int mode,
                        std::vector<std::vector<float>>* acts, float* y)
{
    const uint32_t depth = nn.cfg.depth;
    std::vector<float> cur(enc, enc + nn.enc_dims), nxt;
    for (uint32_t l = 0; l <= depth; l++) {
        const Layer& L = nn.layers[l];
        nxt.assign(L.out, 0.0f);
        for (uint32_t j = 0; j < L.out; j++) {
            const float* wr = wq.data() + L.off + (size_t)j * L.in;
            double acc = 0.0;
            for (uint32_t k = 0; k < L.in; k++) acc += (double)wr[k] * (double)cur[k];
            float z = (float)acc;
            if (l < depth) { z = z > 0.0f ? z : 0.0f; if (mode == 1) z = orc_round_f16(z); }
            nxt[j] = z;
        }
        if (l < depth) { if (acts) (*acts)[l] = nxt; cur.swap(nxt); }
        else { y[0] = nxt[0]; y[1] = nxt[1]; y[2] = nxt[2]; }
    }
}

static std::vector<float> quantized_weights(const std::vector<float>& src, int mode)
{
    std::vector<float> q(src);
    if (mode == 1) for (auto& x : q) x = orc_round_f16(x);
    return q;
}

} // namespace

extern "C" {

void* orc_nn_create(const orc_nn_config* cfg)
{
    NN* nn = new NN();
    nn->cfg = *cfg;
    nn->enc_raw = pos_dims(cfg->pos_id) + dir_dims(cfg->dir_id);
    if (pos_dims(cfg->pos_id) == 0) { delete nn; return nullptr; }
    nn->enc_dims = (nn->enc_raw + 15u) / 16u * 16u;
    size_t off = 0;
    for (uint32_t l = 0; l <= cfg->depth; l++) {
        Layer L;
        L.in = (l == 0) ? nn->enc_dims : cfg->width;
        L.out = (l == cfg->depth) ? 3u : cfg->width;
        L.off = off; off += (size_t)L.in * L.out;
        nn->layers.push_back(L);
    }
    nn->n_mlp = off;
    size_t n_grid = 0;
    if (cfg->pos_id == 0) {                       /* per-level tables: dense while res^3 fits, else 2^log2_hashmap_size entries */
        const uint32_t log2_size = cfg->hashgrid_log2_size ? cfg->hashgrid_log2_size : 19u;
        uint32_t o = 0;
        for (uint32_t l = 0; l < HG_LEVELS; l++) {
            const uint32_t res = hg_resolution(hg_scale(l));
            const double dense = (double)res * res * res;
            uint32_t cnt = dense > 2147483647.0 ? 2147483647u : (uint32_t)dense;
            cnt = (cnt + 7u) / 8u * 8u;
            cnt = std::min(cnt, 1u << log2_size);
            nn->hg_off[l] = o; o += cnt;
        }
        nn->hg_off[HG_LEVELS] = o;
        n_grid = (size_t)o * HG_FEATS;
    }
    nn->w.assign(off + n_grid, 0.0f); nn->ema = nn->w; nn->m = nn->w; nn->v = nn->w; nn->grad = nn->w;
    /* tiny-cuda-nn v1.6 initialisation, recalled from upstream (PARITY UNPINNED: the submodule is empty):
     *   Trainer::initialize_params: pcg32 rng{seed = 1337} (stream 1) -> NetworkWithInputEncoding::initialize_params: the network
     *   first, the encoding's parameters after it.
     *   FullyFusedMLP::initialize_params: the matrices in order -- first (width x enc), hidden (width x width), output with its rows
     *   PADDED to 16 (16 x width) -- each GPUMatrix::initialize_xavier_uniform: scale = sqrt(6 / (fan_in + fan_out)) of the STORED
     *   shape (rows + columns), element i (row-major) = next_float() * 2 * scale - scale, evaluated left to right in fp32.
     *   The output matrix therefore draws 16 * width numbers under the bound sqrt(6 / (16 + width)); rows 3..15 produce the padded
     *   outputs nobody reads (this model keeps rows 0..2; orc_nn_tcnn_dead_rows hands out the rest for the tcnn-layout dump). */
    Pcg32 rng; rng.seed(cfg->seed, 1);
    nn->dead_rows.assign((size_t)13 * cfg->width, 0.0f);
    for (uint32_t l = 0; l <= cfg->depth; l++) {
        const Layer& L = nn->layers[l];
        const uint32_t rows = (l == cfg->depth) ? 16u : L.out;
        const float scale = 1.0f * sqrtf(6.0f / (float)(L.in + rows));
        for (size_t i = 0; i < (size_t)L.in * rows; i++) {
            const float x = rng.nextf() * 2.0f * scale - scale;
            if (i < (size_t)L.in * L.out) nn->w[L.off + i] = x;
            else nn->dead_rows[i - (size_t)L.in * L.out] = x;
        }
    }
    /* GridEncoding::initialize_params: generate_random_uniform<float>(rng, n_params, -1e-4, 1e-4) ON THE GPU: ceil(n / 4) threads in
     * blocks of 128, thread i skips ahead 4 i draws and writes draw 4 i + j to element i + n_threads * j (n_threads = the whole grid,
     * j = 0..3, elements beyond n dropped), value = next_float() * (upper - lower) + lower contracted to one FMA by nvcc. */
    if (n_grid > 0) {
        const size_t n_threads = ((n_grid + 3) / 4 + 127) / 128 * 128;
        const float lower = -1e-4f, upper = 1e-4f;
        for (size_t k = 0; k < 4 * n_threads; k++) {
            const float u = rng.nextf();
            const size_t idx = k / 4 + n_threads * (k % 4);
            if (idx < n_grid) nn->w[off + idx] = fmaf(u, upper - lower, lower);
        }
    }
    nn->ema = nn->w;
    nn->step = 0;
    return nn;
}

void orc_nn_destroy(void* p) { delete (NN*)p; }
uint32_t orc_nn_param_count(void* p) { return (uint32_t)((NN*)p)->w.size(); }
uint32_t orc_nn_encoded_dims(void* p) { return ((NN*)p)->enc_dims; }
uint32_t orc_nn_mlp_param_count(void* p) { return (uint32_t)((NN*)p)->n_mlp; }
void orc_nn_set_step(void* p, uint32_t step) { ((NN*)p)->step = step; }
const float* orc_nn_tcnn_dead_rows(void* p) { return ((NN*)p)->dead_rows.data(); }
/* the generator by itself (known-answer test: pcg-c-basic's published demo vector for seed 42, stream 54) */
void orc_pcg32(uint64_t seed, uint64_t seq, uint32_t n, uint32_t* out_u32, float* out_float)
{
    Pcg32 a; a.seed(seed, seq);
    for (uint32_t i = 0; i < n; i++) out_u32[i] = a.next();
    Pcg32 b; b.seed(seed, seq);
    for (uint32_t i = 0; i < n; i++) out_float[i] = b.nextf();
}

float* orc_nn_buffer(void* p, int which)
{
    NN* nn = (NN*)p;
    switch (which) { case 0: return nn->w.data(); case 1: return nn->ema.data(); case 2: return nn->m.data();
                     case 3: return nn->v.data(); default: return nn->grad.data(); }
}

void orc_nn_encode(void* p, const float* in, uint32_t n, float* out)
{
    NN* nn = (NN*)p;
    for (uint32_t i = 0; i < n; i++)
        encode_one(*nn, in + 5 * (size_t)i, out + (size_t)nn->enc_dims * i, nn->ema.data() + nn->n_mlp);
}

/* NeuralRadianceCache::Inference -> network->inference (src/NeuralRadianceCache.cu:134-145): EMA weights */
void orc_nn_forward(void* p, const float* in, uint32_t n, int use_ema, int mode, float* out)
{
    NN* nn = (NN*)p;
    const std::vector<float>& src = use_ema ? nn->ema : nn->w;
    std::vector<float> wq = quantized_weights(std::vector<float>(src.begin(), src.begin() + nn->n_mlp), mode);
    std::vector<float> enc(nn->enc_dims);
    for (uint32_t i = 0; i < n; i++) {
        encode_one(*nn, in + 5 * (size_t)i, enc.data(), src.data() + nn->n_mlp, mode == 1);
        forward_one(*nn, wq, enc.data(), mode, nullptr, out + 3 * (size_t)i);
    }
}

/* trainer->training_step minus the optimizer (src/NeuralRadianceCache.cu:147-156); loss a9, SURVEY App. B */
float orc_nn_backward(void* p, const float* in, const float* target, uint32_t n, uint32_t n_norm, int accumulate)
{
    NN* nn = (NN*)p;
    const uint32_t depth = nn->cfg.depth, width = nn->cfg.width;
    const float loss_scale = 128.0f;
    std::vector<float> wq = quantized_weights(std::vector<float>(nn->w.begin(), nn->w.begin() + nn->n_mlp), 1);
    std::vector<double> g(nn->w.size(), 0.0);
    const float* table = nn->w.data() + nn->n_mlp;
    std::vector<float> enc(nn->enc_dims);
    std::vector<std::vector<float>> acts(depth);
    std::vector<float> delta, prev;
    double loss_sum = 0.0;
    const float n_total = (float)(3.0 * (double)n_norm);
    for (uint32_t i = 0; i < n; i++) {
        float y[3];
        encode_one(*nn, in + 5 * (size_t)i, enc.data(), table, true);
        forward_one(*nn, wq, enc.data(), 1, &acts, y);
        const float* t = target + 3 * (size_t)i;
        float dy[3];
        /* tiny-cuda-nn element-wise losses (PARITY UNPINNED: the submodule is absent, formulas from its published
         * loss headers): value / n_total and dL/dy * loss_scale / n_total; relative losses keep their denominator constant */
        float lum2 = 0.0f;
        if (nn->cfg.loss_id == 0) {             /* RelativeL2Luminance */
            float lum = (0.299f * y[0] + 0.587f * y[1]) + 0.114f * y[2];
            lum2 = lum * lum + 0.01f;
        }
        for (int c = 0; c < 3; c++) {
            const float d = y[c] - t[c];
            float value, grad;
            switch (nn->cfg.loss_id) {
            case 0: value = d * d / lum2; grad = 2.0f * d / lum2; break;
            case 1: value = d * d; grad = 2.0f * d; break;                                      /* L2 */
            case 2: { float den = y[c] * y[c] + 0.01f; value = d * d / den; grad = 2.0f * d / den; break; }   /* RelativeL2 */
            case 3: value = fabsf(d); grad = copysignf(1.0f, d); break;                         /* L1 */
            case 4: { float sc = 1.0f / (fabsf(t[c]) + 0.01f); value = fabsf(d) * sc; grad = copysignf(sc, d); break; }   /* Mape */
            case 5: { float sc = 1.0f / (0.5f * (fabsf(y[c]) + fabsf(t[c])) + 0.01f); value = fabsf(d) * sc; grad = copysignf(sc, d); break; }   /* Smape */
            default: { float dv = fabsf(d) + 1.0f; value = logf(dv); grad = copysignf(1.0f / dv, d); break; }      /* LogL1 */
            }
            loss_sum += (double)(value / n_total);
            dy[c] = loss_scale * (grad / n_total);
        }
        delta.assign(3, 0.0f);
        for (int c = 0; c < 3; c++) delta[c] = orc_round_f16(dy[c]);
        for (int l = (int)depth; l >= 0; l--) {
            const Layer& L = nn->layers[l];
            const float* a_in = (l == 0) ? enc.data() : acts[l - 1].data();
            for (uint32_t j = 0; j < L.out; j++) {
                double dj = delta[j];
                if (dj == 0.0) continue;
                double* gr = g.data() + L.off + (size_t)j * L.in;
                for (uint32_t k = 0; k < L.in; k++) gr[k] += dj * (double)a_in[k];
            }
            if (l > 0) {
                prev.assign(width, 0.0f);
                for (uint32_t k = 0; k < L.in; k++) {
                    double acc = 0.0;
                    for (uint32_t j = 0; j < L.out; j++) acc += (double)wq[L.off + (size_t)j * L.in + k] * (double)delta[j];
                    float d = (a_in[k] > 0.0f) ? (float)acc : 0.0f;
                    prev[k] = orc_round_f16(d);
                }
                delta.swap(prev);
            } else if (nn->cfg.pos_id == 0) {
                /* dL/d(encoding) for the trainable grid: W0^T delta_0 (fp16, like every delta), scattered to the 8 corners */
                for (uint32_t lv = 0; lv < HG_LEVELS; lv++) {
                    uint32_t idx[8]; float w8[8];
                    hg_corners(*nn, lv, in + 5 * (size_t)i, idx, w8);
                    for (int f = 0; f < 2; f++) {
                        const uint32_t k = lv * 2 + f;
                        double acc = 0.0;
                        for (uint32_t j = 0; j < L.out; j++) acc += (double)wq[L.off + (size_t)j * L.in + k] * (double)delta[j];
                        const float de = orc_round_f16((float)acc);
                        if (de == 0.0f) continue;
                        for (int c = 0; c < 8; c++) g[nn->n_mlp + (size_t)idx[c] * 2 + f] += (double)w8[c] * (double)de;
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < g.size(); i++) {
        float gi = (float)(g[i] / (double)loss_scale);
        nn->grad[i] = accumulate ? nn->grad[i] + gi : gi;
    }
    return (float)loss_sum;
}

/* EMA{Adam | SGD}: SURVEY App. B (tcnn defaults beta1 .9, beta2 .999, eps 1e-8, l2_reg 1e-8) */
void orc_nn_optimizer_step(void* p)
{
    NN* nn = (NN*)p;
    nn->step += 1;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, l2 = 1e-8f;
    const double t = (double)nn->step;
    const float lr = nn->cfg.learning_rate * (float)(sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    const double d = (double)nn->cfg.ema_decay;
    const float ema_old = (float)(d * (1.0 - pow(d, t - 1.0)));
    const float ema_new = (float)(1.0 - d);
    const float ema_div = (float)(1.0 - pow(d, t));
    if (nn->cfg.optimizer_id == 1u) {
        /* tiny-cuda-nn sgd.h (defaults l2_reg 1e-8, no bias correction): w -= lr * (g + l2 * w) for every parameter */
        for (size_t i = 0; i < nn->w.size(); i++) {
            float w = nn->w[i];
            const float g = nn->grad[i] + l2 * w;
            w = w - nn->cfg.learning_rate * g;
            nn->w[i] = w;
            nn->ema[i] = (nn->ema[i] * ema_old + w * ema_new) / ema_div;
        }
        return;
    }
    for (size_t i = 0; i < nn->w.size(); i++) {
        float w = nn->w[i];
        /* tiny-cuda-nn Adam: L2 only on matrix (MLP) weights; grid entries with a zero gradient are left untouched */
        const bool matrix = i < nn->n_mlp;
        if (!matrix && nn->grad[i] == 0.0f) {
            nn->ema[i] = (nn->ema[i] * ema_old + w * ema_new) / ema_div;
            continue;
        }
        float g = nn->grad[i] + (matrix ? l2 * w : 0.0f);
        float m = nn->m[i] = b1 * nn->m[i] + (1.0f - b1) * g;
        float v = nn->v[i] = b2 * nn->v[i] + (1.0f - b2) * (g * g);
        w = w - lr * m / (sqrtf(v) + eps);
        nn->w[i] = w;
        nn->ema[i] = (nn->ema[i] * ema_old + w * ema_new) / ema_div;
    }
}

} // extern "C"
