// Micro-reproducer: does a VALU instruction that consumes the result of a transcendental instruction (v_rcp_f32, v_sqrt_f32, ...)
// ONE wait state later -- the spacing the compiler guarantees on gfx940-class targets -- always see the result in all 64 lanes
// when waves of other queues with raised priority compete for the same SIMD?
//
// Background: the two non-determinism events of round 2 reproduce as a k_gen_rays wave whose lanes 48..63
// -- the last of the four 16-lane passes of a wave64 VALU instruction -- leave new_ray_dir with a slightly different direction
// while high-priority waves (k_infer: v_sin / v_cos / MFMA) are co-resident.  In k_gen_rays 58 of 80 transcendental results are
// consumed at distance 1 (one independent instruction in between).
//
// The victim kernel runs v_rcp_f32 / v_sqrt_f32 with the dependent instruction behind a chosen separator (inline asm) and compares
// every result with the same computation behind a VALU instruction + s_nop 7; the aggressor keeps trans / MFMA / LDS work with
// s_setprio 3 in flight on a high-priority stream.  Output: mismatches by lane, per distance.
//
//   hipcc --offload-arch=gfx950 -O3 -o trans_hazard tools/trans_hazard.hip -pthread
//   ./trans_hazard [seconds per case = 6] [aggressor kind: 0 none, 1 trans, 2 mfma, 3 both = 3]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CHK(e)                                                                                      \
    do {                                                                                            \
        hipError_t _e = (e);                                                                        \
        if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); std::exit(3); } \
    } while (0)

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// e = fma(-x, rcp(x), 1) and m = bits(sqrt(x)) - 1 (the first step of the correctly rounded sqrt's fix-up) with the consumer behind a
// chosen separator.  MODE: 0 back to back (the compiler never emits that on this target); 1 one independent VALU instruction;
// 10 "s_nop 0" -- what the compiler emits for 44 of the 80 transcendental results in k_gen_rays; 11 "s_nop 1"; 12 one SALU
// instruction (s_mov_b32); 13 one v_readlane_b32 (an SGPR reload from a spill lane); 8 the reference: VALU + "s_nop 7"
#define NRC_TRANS_CASE(MODE, SEP)                                                                                                  \
    template <>                                                                                                                     \
    __device__ __forceinline__ float rcp_use<MODE>(float x, float& side)                                                            \
    {                                                                                                                               \
        float r, e;                                                                                                                 \
        unsigned sg;                                                                                                                \
        asm volatile("v_rcp_f32 %0, %4\n\t" SEP "\n\tv_fma_f32 %1, -%4, %0, 1.0" : "=&v"(r), "=&v"(e), "=&v"(side), "=&s"(sg) : "v"(x));     \
        return e;                                                                                                                   \
    }                                                                                                                               \
    template <>                                                                                                                     \
    __device__ __forceinline__ unsigned sqrt_use<MODE>(float x, float& side)                                                        \
    {                                                                                                                               \
        float y;                                                                                                                    \
        unsigned m, sg;                                                                                                             \
        asm volatile("v_sqrt_f32 %0, %4\n\t" SEP "\n\tv_add_u32 %1, -1, %0" : "=&v"(y), "=&v"(m), "=&v"(side), "=&s"(sg) : "v"(x));           \
        return m;                                                                                                                   \
    }
template <int MODE>
__device__ __forceinline__ float rcp_use(float x, float& side);
template <int MODE>
__device__ __forceinline__ unsigned sqrt_use(float x, float& side);
NRC_TRANS_CASE(0, "")
NRC_TRANS_CASE(1, "v_mul_f32 %2, %4, %4")
NRC_TRANS_CASE(10, "s_nop 0")
NRC_TRANS_CASE(11, "s_nop 1")
NRC_TRANS_CASE(12, "s_mov_b32 %3, 7")
NRC_TRANS_CASE(13, "v_readlane_b32 %3, %4, 5")
NRC_TRANS_CASE(8, "v_mul_f32 %2, %4, %4\n\ts_nop 7")

template <int DIST>
__global__ __launch_bounds__(256, 5) void k_victim(unsigned seed, unsigned iters, unsigned* __restrict__ hist /* [2][64] */, unsigned long long* __restrict__ checks)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned s = mix(seed * 0x9e3779b9u + blockIdx.x * 256u + threadIdx.x);
    float side = 0.0f, acc = 0.0f;
    unsigned bad_rcp = 0, bad_sqrt = 0;
    for (unsigned i = 0; i < iters; i++) {
        s = mix(s + i);
        // the destination register of the trans instruction held something else before: dirty it, as real code does
        const float x = __builtin_bit_cast(float, (s & 0x007fffffu) | 0x3f800000u) * 3.7f;      // [3.7, 7.4)
        const float e = rcp_use<DIST>(x, side);
        const float e_ref = rcp_use<8>(x, side);
        bad_rcp += __builtin_bit_cast(unsigned, e) != __builtin_bit_cast(unsigned, e_ref);
        const unsigned m = sqrt_use<DIST>(x, side);
        const unsigned m_ref = sqrt_use<8>(x, side);
        bad_sqrt += m != m_ref;
        acc += side;
    }
    if (bad_rcp) atomicAdd(&hist[lane], bad_rcp);
    if (bad_sqrt) atomicAdd(&hist[64 + lane], bad_sqrt);
    if (lane == 0 && (threadIdx.x >> 6) == 0) atomicAdd(checks, (unsigned long long)iters * 256ull);
    if (acc == 1.2345f) hist[0] = 0;
}

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_aggressor(int kind, unsigned long long cycles, float* __restrict__ sink)
{
    __builtin_amdgcn_s_setprio(3);
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.01f * (threadIdx.x + j)); b[j] = (_Float16)(0.02f * j); }
    f32x16 c;
    for (int j = 0; j < 16; j++) c[j] = 0.0f;
    float t = 0.001f * threadIdx.x, u = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        if (kind & 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                t = __builtin_amdgcn_sinf(t) + __builtin_amdgcn_cosf(u);
                u = __builtin_amdgcn_rcpf(t + 2.0f) + __builtin_amdgcn_sqrtf(u + 1.0f);
            }
        }
        if (kind & 2) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
        }
    }
    if (c[0] + t + u == 123.456f) sink[0] = c[1];
}

template <int DIST>
static void run_case(double seconds, int aggressor, unsigned* d_hist, unsigned long long* d_checks, float* d_sink)
{
    CHK(hipMemset(d_hist, 0, 128 * 4));
    CHK(hipMemset(d_checks, 0, 8));
    std::atomic<bool> stop{false};
    std::thread th;
    hipStream_t sa = nullptr;
    if (aggressor) {
        int lo = 0, hi = 0;
        CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CHK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
        th = std::thread([&] {
            CHK(hipSetDevice(0));
            unsigned n = 0;
            while (!stop.load()) {
                const unsigned long long cyc = 1000ull + ((n * 2654435761u) >> 19);      // 10 .. 90 us
                hipLaunchKernelGGL(k_aggressor, dim3(512), dim3(512), 0, sa, aggressor, cyc, d_sink);
                if ((++n & 7u) == 0) CHK(hipStreamSynchronize(sa));
            }
            CHK(hipStreamSynchronize(sa));
        });
    }
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, nullptr));
    double elapsed = 0.0;
    unsigned launch = 0;
    while (elapsed < seconds * 1e3) {
        for (int k = 0; k < 8; k++) hipLaunchKernelGGL(k_victim<DIST>, dim3(8100), dim3(256), 0, nullptr, launch++, 400u, d_hist, d_checks);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        elapsed = ms;
    }
    stop.store(true);
    if (th.joinable()) th.join();
    if (sa) CHK(hipStreamDestroy(sa));
    CHK(hipDeviceSynchronize());
    unsigned hist[128];
    unsigned long long checks = 0;
    CHK(hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&checks, d_checks, 8, hipMemcpyDeviceToHost));
    unsigned long long bad[2] = {0, 0}, quarter[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int k = 0; k < 2; k++)
        for (int l = 0; l < 64; l++) { bad[k] += hist[64 * k + l]; quarter[k][l / 16] += hist[64 * k + l]; }
    std::printf("separator mode %d, aggressor %d: %.3g checks each | v_rcp_f32 -> v_fma_f32: %llu wrong (lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu) | "
                "v_sqrt_f32 -> v_add_u32: %llu wrong (%llu, %llu, %llu, %llu)\n",
                DIST, aggressor, (double)checks, bad[0], quarter[0][0], quarter[0][1], quarter[0][2], quarter[0][3], bad[1], quarter[1][0], quarter[1][1], quarter[1][2], quarter[1][3]);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 6.0;
    const int aggressor = argc > 2 ? std::atoi(argv[2]) : 3;
    CHK(hipSetDevice(0));
    unsigned* d_hist;
    unsigned long long* d_checks;
    float* d_sink;
    CHK(hipMalloc(&d_hist, 128 * 4));
    CHK(hipMalloc(&d_checks, 8));
    CHK(hipMalloc(&d_sink, 64));
    std::printf("separator modes: 0 none, 1 one VALU instruction, 10 s_nop 0 (the compiler's usual choice), 11 s_nop 1, 12 one SALU instruction, 13 one v_readlane_b32\n");
    run_case<0>(seconds, 0, d_hist, d_checks, d_sink);
    run_case<10>(seconds, 0, d_hist, d_checks, d_sink);
    run_case<10>(seconds, aggressor, d_hist, d_checks, d_sink);
    run_case<12>(seconds, aggressor, d_hist, d_checks, d_sink);
    run_case<13>(seconds, aggressor, d_hist, d_checks, d_sink);
    run_case<11>(seconds, aggressor, d_hist, d_checks, d_sink);
    run_case<1>(seconds, aggressor, d_hist, d_checks, d_sink);
    return 0;
}
