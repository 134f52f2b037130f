// Stress harness for the two non-determinism observations of round 2 (VERDICT r02 "Next round" 1):
//   (i)  tiles-vs-whole: the eight column tiles of a blended multi-frame render (NRC inference + compositing, training off) must
//        reassemble to the single-renderer frame bit for bit (tests/test_gpu_baseline_configs.py, configs[3]);
//   (ii) pipelined-vs-single-stream: the four-stream frame graph with training on must give the framebuffer, loss and weights of
//        the single-stream order bit for bit (tests/test_gpu_integrator.py::test_pipelined_streams_equal_single_stream_bitwise).
// Each process renders both comparisons `iters` times in FRESH renderers / caches while, optionally, a host thread keeps a
// perturbing kernel in flight on a fifth, high-priority stream (wave priority raised, one wave or one LDS-heavy workgroup per CU), so
// that the product's kernels are co-resident with foreign waves the way an RCCL kernel would be.  On the first mismatch it prints
// FNV-1a hashes of every intermediate buffer of both sides (primary, info, infer_input, infer_output, train rays, weights) to name
// the first buffer that differs, and exits 1.
//
//   stress_main <mode: tiles|pipe|both> <iters> <perturb: 0|1|2> [width height frames]
// Linked against libnrc_hpm.so (rpath); tools/stress.sh runs it >= 100 times in fresh processes under the environment variants
// NRC_DEBUG=poison_alloc,guard_alloc, GPU_MAX_HW_QUEUES=2/4/8 and against the diagnostic -DNRC_DIAG_LOWPRIO=8 build (camera kernels at wave priority 0 beside raised neighbours).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <nrc_hpm.hpp>

#define HIPCHK(e)                                                                                     \
    do {                                                                                              \
        hipError_t _e = (e);                                                                          \
        if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); std::exit(3); } \
    } while (0)

// ---- perturbing kernels ---------------------------------------------------------------------------------------------------------
__global__ void k_spin(unsigned long long cycles, unsigned* sink)
{
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) acc = acc * 1664525u + 1013904223u;
    if (acc == 0xdeadbeefu) sink[0] = acc;
}
// the same spin with other instruction mixes (which execution resource does the perturbation go through?): 3 full-rate integer
// (shift / xor / add, no multiply), 4 transcendental (v_sin_f32 / v_rcp_f32), 5 no vector work at all (scalar clock polling only),
// 6 as 1 without the raised priority
template <int KIND>
__global__ void k_spin_kind(unsigned long long cycles, unsigned* sink)
{
    if (KIND != 6) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = threadIdx.x;
    float f = 0.001f * (float)threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        if (KIND == 3) acc = (acc ^ (acc << 5)) + 0x9e3779b9u;
        else if (KIND == 4) f = __builtin_amdgcn_sinf(f) + __builtin_amdgcn_rcpf(f + 2.0f);
        else if (KIND == 6) acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0xdeadbeefu || f == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(1024) void k_spin_lds(unsigned long long cycles, unsigned* sink)
{
    __shared__ unsigned lds[16 * 1024];      // 64 KB: competes with k_infer / k_train_fwd_bwd for a CU's LDS
    __builtin_amdgcn_s_setprio(3);
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) lds[i] = i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) acc = lds[acc & 16383u] * 1664525u + 1013904223u;
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

struct Perturber {
    std::atomic<bool> stop{false};
    std::thread th;
    hipStream_t s = nullptr;
    unsigned* sink = nullptr;
    unsigned long launches = 0;
    void start(int kind)
    {
        if (kind == 0) return;
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        HIPCHK(hipMalloc(&sink, 4));
        th = std::thread([this, kind] {
            HIPCHK(hipSetDevice(0));
            unsigned n = 0;
            while (!stop.load()) {
                // bursts of 20-60 us at irregular intervals
                const unsigned long long cyc = 2000ull + (unsigned long long)((n * 2654435761u) >> 20);      // 100 MHz clock: 20 .. 61 us
                if (kind == 1) hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s, cyc, sink);
                else if (kind == 3) hipLaunchKernelGGL(k_spin_kind<3>, dim3(256), dim3(64), 0, s, cyc, sink);
                else if (kind == 4) hipLaunchKernelGGL(k_spin_kind<4>, dim3(256), dim3(64), 0, s, cyc, sink);
                else if (kind == 5) hipLaunchKernelGGL(k_spin_kind<5>, dim3(256), dim3(64), 0, s, cyc, sink);
                else if (kind == 6) hipLaunchKernelGGL(k_spin_kind<6>, dim3(256), dim3(64), 0, s, cyc, sink);
                else hipLaunchKernelGGL(k_spin_lds, dim3(128), dim3(1024), 0, s, cyc, sink);
                if ((++n & 7u) == 0) HIPCHK(hipStreamSynchronize(s));
                launches++;
            }
            HIPCHK(hipStreamSynchronize(s));
        });
    }
    void finish()
    {
        if (!th.joinable()) return;
        stop.store(true);
        th.join();
        HIPCHK(hipStreamDestroy(s));
        HIPCHK(hipFree(sink));
    }
};

// ---- scene ----------------------------------------------------------------------------------------------------------------------
static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// a lumpy ellipsoid cloud, u8, max 255 (not the oracle's cloud: both sides of every comparison run on the GPU)
static std::vector<uint8_t> make_volume(uint32_t n)
{
    std::vector<uint8_t> v((size_t)n * n * n);
    const float blobs[6][4] = {{0.5f, 0.5f, 0.5f, 0.30f}, {0.38f, 0.55f, 0.45f, 0.2f}, {0.62f, 0.45f, 0.55f, 0.22f},
                               {0.5f, 0.62f, 0.6f, 0.16f}, {0.45f, 0.4f, 0.35f, 0.15f}, {0.58f, 0.52f, 0.68f, 0.14f}};
    for (uint32_t z = 0; z < n; z++)
        for (uint32_t y = 0; y < n; y++)
            for (uint32_t x = 0; x < n; x++) {
                const float fx = (x + 0.5f) / n, fy = (y + 0.5f) / n, fz = (z + 0.5f) / n;
                float d = 0.0f;
                for (auto& b : blobs) {
                    const float r2 = ((fx - b[0]) * (fx - b[0]) + (fy - b[1]) * (fy - b[1]) * 1.8f + (fz - b[2]) * (fz - b[2])) / (b[3] * b[3]);
                    d += std::max(0.0f, 1.0f - r2);
                }
                const float noise = (hash32((x >> 2) + 977u * (y >> 2) + 131071u * (z >> 2)) & 0xffff) / 65535.0f;
                d = std::min(1.0f, std::max(0.0f, d * (0.55f + 0.6f * noise) - 0.08f));
                v[((size_t)z * n + y) * n + x] = (uint8_t)(d * 255.0f);
            }
    v[(((size_t)n / 2) * n + n / 2) * n + n / 2] = 255;
    return v;
}
static std::vector<float> make_sky(uint32_t w, uint32_t h)
{
    std::vector<float> e((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const float t = (float)y / (h - 1), s = (float)x / w;
            float* p = &e[((size_t)y * w + x) * 4];
            const float sun = std::exp(-60.0f * ((s - 0.3f) * (s - 0.3f) + (t - 0.7f) * (t - 0.7f)));
            p[0] = 0.3f + 0.5f * t + 6.0f * sun; p[1] = 0.4f + 0.5f * t + 5.0f * sun; p[2] = 0.6f + 0.4f * t + 4.0f * sun; p[3] = 1.0f;
        }
    return e;
}

// ---- helpers --------------------------------------------------------------------------------------------------------------------
static uint64_t fnv(const void* p, size_t n)
{
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static std::vector<unsigned char> download(const void* d, size_t bytes)
{
    std::vector<unsigned char> h(bytes);
    HIPCHK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
    return h;
}
static const char* kBufNames[9] = {"primary", "info", "origin", "dir", "infer_input", "infer_output", "train_input", "train_target", "ring"};

struct Snapshot {
    std::vector<float> image;
    std::vector<std::vector<unsigned char>> bufs;      // the nine renderer buffers of the last frame
    std::vector<std::vector<unsigned char>> sets[6];   // gen_rays output sets (the last six frames): primary, info, infer_input
    std::vector<float> w, ema;
    float loss = 0.0f;
};

struct Setup {
    nrc_scene scene{};
    std::vector<uint8_t> vol;
    std::vector<float> sky;
    std::vector<float> randoms;
    uint32_t frames = 8;
};

static void frame_randoms(Setup& s, uint32_t seed)
{
    s.randoms.resize((size_t)s.frames * 4);
    uint32_t x = seed;
    for (float& r : s.randoms) { x = hash32(x + 0x9e3779b9u); r = (float)(x >> 8) * (1.0f / 16777216.0f); }
}

// renders `frames` blended frames of the (global gw x gh) view on tile `tile` (nullptr: whole frame) in a fresh cache + renderer
static bool g_keep_sets = false;
static Snapshot render(const Setup& su, const nrc_config& cfg, uint32_t gw, uint32_t gh, uint32_t lw, const nrc_tile* tile, bool train, bool keep_bufs)
{
    en::Camera camera(en::vec3(64.0f, 0.0f, 0.0f), en::vec3(-1.0f, 0.0f, 0.0f), en::vec3(0.0f, 1.0f, 0.0f), (float)gw / (float)gh,
                      en::radians(60.0f), 0.1f, 100.0f);
    nrc_cache_t* c = nullptr;
    nrc_renderer_t* r = nullptr;
    en::nrc_check(nrc_cache_create(&cfg, &c));
    en::nrc_check(nrc_renderer_create(lw, gh, 1, camera.Matrices(), &cfg, &su.scene, c, tile, nullptr, &r));
    if (g_keep_sets) en::nrc_check(nrc_renderer_set_full_vertex_images(r, 1));      // (diagnostic builds put their probes into the w components)
    for (uint32_t f = 0; f < su.frames; f++) {
        en::nrc_check(nrc_renderer_set_frame_random(r, &su.randoms[(size_t)f * 4]));
        en::nrc_check(nrc_renderer_render(r, train ? 1 : 0));        // no host synchronisation between frames
    }
    Snapshot s;
    const float* d_img = nrc_renderer_framebuffer(r);
    HIPCHK(hipDeviceSynchronize());
    s.image.resize((size_t)lw * gh * 4);
    HIPCHK(hipMemcpy(s.image.data(), d_img, s.image.size() * 4, hipMemcpyDeviceToHost));
    if (g_keep_sets)
        for (int k = 0; k < 6; k++)
            for (int b : {0, 1, 4, 2, 3}) {
                size_t bytes = 0;
                void* p = nrc_renderer_buffer(r, b + 16 * (k + 1), &bytes);
                s.sets[k].push_back(download(p, bytes));
            }
    if (keep_bufs)
        for (int b = 0; b < 9; b++) {
            size_t bytes = 0;
            void* p = nrc_renderer_buffer(r, b, &bytes);
            s.bufs.push_back(download(p, bytes));
        }
    if (train) {
        s.loss = nrc_cache_get_loss(c);
        const uint32_t n = nrc_cache_param_count(c);
        s.w.resize(n); s.ema.resize(n);
        en::nrc_check(nrc_cache_get_params(c, 0, s.w.data()));
        en::nrc_check(nrc_cache_get_params(c, 1, s.ema.data()));
    }
    en::nrc_check(nrc_renderer_destroy(r));
    en::nrc_check(nrc_cache_destroy(c));
    return s;
}

static void report(const char* side, const Snapshot& s)
{
    std::printf("  %-8s image %016llx loss %.9g w %016llx ema %016llx", side, (unsigned long long)fnv(s.image.data(), s.image.size() * 4), s.loss,
                (unsigned long long)fnv(s.w.data(), s.w.size() * 4), (unsigned long long)fnv(s.ema.data(), s.ema.size() * 4));
    for (size_t b = 0; b < s.bufs.size(); b++) std::printf(" %s %016llx", kBufNames[b], (unsigned long long)fnv(s.bufs[b].data(), s.bufs[b].size()));
    std::printf("\n");
}

static size_t count_diff(const std::vector<float>& a, const std::vector<float>& b, size_t* first)
{
    size_t n = 0;
    *first = (size_t)-1;
    for (size_t i = 0; i < a.size(); i++)
        if (std::memcmp(&a[i], &b[i], 4) != 0) { if (n++ == 0) *first = i; }
    return n;
}

// global column of local column i of rank `r` (nrc_tile mapping, include/nrc_hpm.h)
static uint32_t global_col(uint32_t i, uint32_t r, uint32_t world, uint32_t block) { return (r + (i / block) * world) * block + i % block; }

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: stress_main <tiles|pipe|pipeq2|both> <iters> <perturb 0|1|2> [width height frames]\n"); return 2; }
    const std::string mode = argv[1];
    const int iters = std::atoi(argv[2]), perturb = std::atoi(argv[3]);
    const uint32_t GW = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 3840, GH = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 2160;
    try {
        HIPCHK(hipSetDevice(0));
        Setup su;
        su.frames = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 8;
        su.vol = make_volume(160);
        su.sky = make_sky(64, 32);
        su.scene.density = su.vol.data(); su.scene.nx = su.scene.ny = su.scene.nz = 160;
        su.scene.density_factor = 0.6f; su.scene.g = 0.8f;
        su.scene.dir_light_dir[0] = 0.0f; su.scene.dir_light_dir[1] = 7.96e-4f; su.scene.dir_light_dir[2] = -1.0f;
        su.scene.dir_light_strength = 8.0f;
        su.scene.point_light_color[0] = su.scene.point_light_color[1] = su.scene.point_light_color[2] = 1.0f;
        su.scene.env_strength = 0.1f; su.scene.env = su.sky.data(); su.scene.env_w = 64; su.scene.env_h = 32;
        nrc_config cfg;
        nrc_config_default(&cfg);
        cfg.train_batch_count = 1; cfg.log2_train_batch_size = 11; cfg.log2_infer_batch_size = 21;

        Perturber pert;
        pert.start(perturb);
        int bad_runs = 0;
        for (int it = 0; it < iters; it++) {
            frame_randoms(su, 1000u + (uint32_t)it);
            if (mode == "tiles" || mode == "both") {
                const uint32_t world = 8, block = 8;
                g_keep_sets = true;
                Snapshot whole = render(su, cfg, GW, GH, GW, nullptr, false, false);
                std::vector<float> got((size_t)GW * GH * 4);
                std::vector<Snapshot> parts;
                for (uint32_t r = 0; r < world; r++) {
                    const uint32_t lw = GW / world;
                    nrc_tile t{r, world, GW, GH, block};
                    parts.push_back(render(su, cfg, GW, GH, lw, &t, false, false));
                    const Snapshot& part = parts.back();
                    for (uint32_t y = 0; y < GH; y++)
                        for (uint32_t i = 0; i < lw; i++)
                            std::memcpy(&got[((size_t)y * GW + global_col(i, r, world, block)) * 4], &part.image[((size_t)y * lw + i) * 4], 16);
                }
                size_t first = 0;
                const size_t nd = count_diff(got, whole.image, &first);
                if (nd) {
                    bad_runs++;
                    const size_t px = first / 4;
                    std::printf("MISMATCH tiles-vs-whole iter %d: %zu floats differ, first at pixel (x %zu, y %zu) channel %zu: tiles %.9g whole %.9g\n", it, nd,
                                px % GW, px / GW, first % 4, got[first], whole.image[first]);
                    // every differing pixel with its place in the tile side's and the whole side's query order (x * H + y): the
                    // inference kernel hands 32 consecutive queries to a wave half, two such tiles to a wave
                    size_t shown = 0;
                    for (size_t p = 0; p < (size_t)GW * GH && shown < 96; p++) {
                        if (std::memcmp(&got[p * 4], &whole.image[p * 4], 16) == 0) continue;
                        const uint32_t x = (uint32_t)(p % GW), y = (uint32_t)(p / GW);
                        const uint32_t strip = x / block, r = strip % world, lx = (strip / world) * block + x % block;
                        const size_t qt = (size_t)lx * GH + y, qw = (size_t)x * GH + y;
                        std::printf("   px (%u, %u) rank %u tile-side query %zu = 64 * %zu + %zu | whole-side query %zu = 64 * %zu + %zu (batch %zu) | d rgb %.3g %.3g %.3g of %.3g\n",
                                    x, y, r, qt, qt / 64, qt % 64, qw, qw / 64, qw % 64, qw >> 21, got[p * 4] - whole.image[p * 4],
                                    got[p * 4 + 1] - whole.image[p * 4 + 1], got[p * 4 + 2] - whole.image[p * 4 + 2], whole.image[p * 4]);
                        shown++;
                    }
                    // the gen_rays outputs of the last four frames (the four buffer sets), tile side against whole side, at the pixels
                    // that differ: names the kernel (k_gen_rays when primary / info / query differ; inference or compositing otherwise)
                    shown = 0;
                    for (size_t p = 0; p < (size_t)GW * GH && shown < 20; p++) {
                        if (std::memcmp(&got[p * 4], &whole.image[p * 4], 16) == 0) continue;
                        shown++;
                        const uint32_t x = (uint32_t)(p % GW), y = (uint32_t)(p / GW);
                        const uint32_t strip = x / block, r = strip % world, lx = (strip / world) * block + x % block, lw = GW / world;
                        for (int k = 0; k < 6; k++) {
                            const float* pt = (const float*)&parts[r].sets[k][0][((size_t)y * lw + lx) * 16];
                            const float* pw = (const float*)&whole.sets[k][0][((size_t)y * GW + x) * 16];
                            const float* it = (const float*)&parts[r].sets[k][1][((size_t)y * lw + lx) * 4];
                            const float* iw = (const float*)&whole.sets[k][1][((size_t)y * GW + x) * 4];
                            const float* qt = (const float*)&parts[r].sets[k][2][((size_t)lx * GH + y) * 20];
                            const float* qw = (const float*)&whole.sets[k][2][((size_t)x * GH + y) * 20];
                            const bool dp = std::memcmp(pt, pw, 16) != 0, di = std::memcmp(it, iw, 4) != 0, dq = std::memcmp(qt, qw, 20) != 0;
                            const float* ot = (const float*)&parts[r].sets[k][3][((size_t)y * lw + lx) * 16];
                            const float* ow = (const float*)&whole.sets[k][3][((size_t)y * GW + x) * 16];
                            const float* dt = (const float*)&parts[r].sets[k][4][((size_t)y * lw + lx) * 16];
                            const float* dw = (const float*)&whole.sets[k][4][((size_t)y * GW + x) * 16];
                            if (dp || di || dq)
                                std::printf("   px (%u, %u) set %d: origin image (product: vertex; -DNRC_DIAG_LASTDIR: incoming direction) %s, final dir %s, [diag build: rng before the last new_ray_dir %s]; LASTDIR rng %08x in %08x %08x %08x out_tiles %08x %08x %08x out_whole %08x %08x %08x moved_in_registers tiles %g whole %g\n",
                                            x, y, k, std::memcmp(ot, ow, 12) ? "DIFF" : "same", std::memcmp(dt, dw, 12) ? "DIFF" : "same", std::memcmp(ot + 3, ow + 3, 4) ? "DIFF" : "same",
                                            ((const unsigned*)ow)[3], ((const unsigned*)ow)[0], ((const unsigned*)ow)[1], ((const unsigned*)ow)[2], ((const unsigned*)dt)[0], ((const unsigned*)dt)[1],
                                            ((const unsigned*)dt)[2], ((const unsigned*)dw)[0], ((const unsigned*)dw)[1], ((const unsigned*)dw)[2], dt[3], dw[3]);
                            if (dp || di || dq)
                                std::printf("   px (%u, %u) set %d: primary %s info %s query %s | tiles prim %.9g %.9g %.9g %.9g info %g q %.9g %.9g %.9g %.9g %.9g | whole prim %.9g %.9g %.9g %.9g info %g q %.9g %.9g %.9g %.9g %.9g\n",
                                            x, y, k, dp ? "DIFF" : "same", di ? "DIFF" : "same", dq ? "DIFF" : "same", pt[0], pt[1], pt[2], pt[3], it[0], qt[0], qt[1], qt[2], qt[3], qt[4],
                                            pw[0], pw[1], pw[2], pw[3], iw[0], qw[0], qw[1], qw[2], qw[3], qw[4]);
                        }
                    }
                    std::printf("  (sets hold the last six of the %u frames; no line above = the gen_rays outputs of those frames agree at the differing pixels)\n", su.frames);
                    g_keep_sets = false;
                    // which side moved: render both again and compare each with its first rendering
                    Snapshot whole2 = render(su, cfg, GW, GH, GW, nullptr, false, true);
                    size_t f2 = 0;
                    std::printf("  whole frame rendered again: %zu floats differ from its first rendering\n", count_diff(whole2.image, whole.image, &f2));
                    const uint32_t r_bad = (uint32_t)(((first / 4) % GW) / block) % world;
                    nrc_tile t{r_bad, world, GW, GH, block};
                    Snapshot part2 = render(su, cfg, GW, GH, GW / world, &t, false, true);
                    size_t nd2 = 0;
                    for (uint32_t y = 0; y < GH; y++)
                        for (uint32_t i = 0; i < GW / world; i++)
                            nd2 += std::memcmp(&whole.image[((size_t)y * GW + global_col(i, r_bad, world, block)) * 4], &part2.image[((size_t)y * (GW / world) + i) * 4], 16) != 0;
                    std::printf("  tile %u rendered again: %zu pixels differ from the whole frame\n", r_bad, nd2);
                    report("whole#2", whole2);
                }
                g_keep_sets = false;
            }
            if (mode == "replay") {
                // the SAME frame (same random numbers, no blending, training off) rendered 4 * reps times without host synchronisation:
                // the four gen_rays output sets must hold identical images and queries; a set that differs names k_gen_rays, identical
                // sets with a differing framebuffer name inference / compositing
                en::Camera camera(en::vec3(64.0f, 0.0f, 0.0f), en::vec3(-1.0f, 0.0f, 0.0f), en::vec3(0.0f, 1.0f, 0.0f), (float)GW / (float)GH,
                                  en::radians(60.0f), 0.1f, 100.0f);
                nrc_cache_t* c = nullptr;
                nrc_renderer_t* r = nullptr;
                en::nrc_check(nrc_cache_create(&cfg, &c));
                en::nrc_check(nrc_renderer_create(GW, GH, 0, camera.Matrices(), &cfg, &su.scene, c, nullptr, nullptr, &r));
                const int reps = (int)su.frames;
                std::vector<float> first_img;
                for (int rep = 0; rep < reps; rep++) {
                    for (int k = 0; k < 6; k++) {
                        en::nrc_check(nrc_renderer_set_frame_random(r, &su.randoms[0]));
                        en::nrc_check(nrc_renderer_render(r, 0));
                    }
                    std::vector<std::vector<unsigned char>> sets[6];
                    for (int k = 0; k < 6; k++)
                        for (int b : {0, 1, 4}) {
                            size_t bytes = 0;
                            void* p = nrc_renderer_buffer(r, b + 16 * (k + 1), &bytes);
                            sets[k].push_back(download(p, bytes));
                        }
                    std::vector<float> img((size_t)GW * GH * 4);
                    HIPCHK(hipMemcpy(img.data(), nrc_renderer_framebuffer(r), img.size() * 4, hipMemcpyDeviceToHost));
                    if (first_img.empty()) first_img = img;
                    const char* names[3] = {"primary", "info", "infer_input"};
                    const size_t stride[3] = {16, 4, 20};
                    bool any = false;
                    for (int k = 1; k < 6; k++)
                        for (int b = 0; b < 3; b++) {
                            if (sets[k][b] == sets[0][b]) continue;
                            any = true;
                            size_t shown = 0;
                            const size_t n = sets[k][b].size() / stride[b];
                            for (size_t e = 0; e < n && shown < 40; e++) {
                                if (std::memcmp(&sets[k][b][e * stride[b]], &sets[0][b][e * stride[b]], stride[b]) == 0) continue;
                                // primary / info are [y][x]; infer_input is [x][y]
                                const size_t x = b == 2 ? e / GH : e % GW, y = b == 2 ? e % GH : e / GW;
                                const float* a0 = (const float*)&sets[0][b][e * stride[b]];
                                const float* ak = (const float*)&sets[k][b][e * stride[b]];
                                std::printf("REPLAY rep %d set %d %s px (%zu, %zu) lane %zu:", rep, k, names[b], x, y, (y & 7) * 8 + (x & 7));
                                for (size_t q = 0; q < stride[b] / 4; q++) std::printf(" %.9g|%.9g", a0[q], ak[q]);
                                std::printf("\n");
                                shown++;
                            }
                        }
                    size_t f0 = 0;
                    const size_t nd = count_diff(img, first_img, &f0);
                    if (nd) { any = true; std::printf("REPLAY rep %d framebuffer: %zu floats differ from the first round's (first at pixel %zu, %zu)\n", rep, nd, (f0 / 4) % GW, (f0 / 4) / GW); }
                    if (any) bad_runs++;
                }
                en::nrc_check(nrc_renderer_destroy(r));
                en::nrc_check(nrc_cache_destroy(c));
            }
            // pipeq2 (round 6): the same comparison with quirk Q2 fixed -- 32-vertex train paths, whose traces of four consecutive frames
            // overlap on streams of their own (k_prep_train<1> / <2>, staging sets) -- against the one-launch kernel of the single-stream order
            if (mode == "pipe" || mode == "both" || mode == "pipeq2") {
                const uint32_t W = 1920, H = 1080;
                nrc_config c2 = cfg;
                c2.log2_train_batch_size = 14;
                if (mode == "pipeq2") { c2.compat_fix = NRC_FIX_Q2_TRAIN_RAY_LEN; c2.train_ray_length = 32; }
                // NRC_DEBUG is a list (csrc/nrc_common.hpp): single_stream is added to what the caller set (poison_alloc, guard_alloc) and removed again
                const char* dbg0 = getenv("NRC_DEBUG");
                const std::string dbg = dbg0 ? dbg0 : "";
                setenv("NRC_DEBUG", (dbg.empty() ? std::string("single_stream") : dbg + ",single_stream").c_str(), 1);
                Snapshot base = render(su, c2, W, H, W, nullptr, true, true);
                if (dbg.empty()) unsetenv("NRC_DEBUG"); else setenv("NRC_DEBUG", dbg.c_str(), 1);
                for (int rep = 0; rep < 3; rep++) {
                    Snapshot p = render(su, c2, W, H, W, nullptr, true, true);
                    size_t first = 0;
                    const size_t nd = count_diff(p.image, base.image, &first);
                    const bool same = nd == 0 && p.loss == base.loss && p.w == base.w && p.ema == base.ema && p.bufs == base.bufs;
                    if (!same) {
                        bad_runs++;
                        std::printf("MISMATCH pipelined-vs-single-stream iter %d rep %d: %zu image floats differ (first %zu)\n", it, rep, nd, first);
                        report("single", base);
                        report("piped", p);
                        for (size_t b = 0; b < p.bufs.size(); b++)
                            if (p.bufs[b] != base.bufs[b]) {
                                size_t i = 0;
                                while (p.bufs[b][i] == base.bufs[b][i]) i++;
                                std::printf("  buffer %s differs first at byte %zu (float index %zu)\n", kBufNames[b], i, i / 4);
                            }
                    }
                }
            }
        }
        pert.finish();
        char msg[512];
        const int guards = nrc_debug_check_guards(msg, sizeof(msg));
        std::printf("stress %s iters %d perturb %d (%lu perturbing launches) frames %u: %d mismatching comparisons; guard check %d %s\n", mode.c_str(), iters, perturb,
                    pert.launches, su.frames, bad_runs, guards, msg);
        return (bad_runs || guards > 0) ? 1 : 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
}
