// Micro-reproducer attempt for the cause of round 2's non-determinism: the packed-FP32 instructions LLVM's SLP
// vectoriser made of new_ray_dir's second rotation write a register pair IN PLACE while reading it with crossed halves --
//     v_pk_fma_f32 v[14:15], v[14:15], v[48:49], v[16:17] op_sel:[1,0,0] op_sel_hi:[0,0,1]      D.lo = f(S0.hi), D.hi = f(S0.lo)
//     v_pk_mov_b32 v[44:45], v[88:89], v[44:45] op_sel:[1,0]                                    D.lo = S0.hi,    D.hi = S1.lo (= old D.lo)
// -- and a k_gen_rays wave that ran them next to waves of a higher issue priority left lanes 48..63 with a different result.  This
// program runs such instructions at wave priority 0 under an aggressor at priority 3 and compares every result with unpacked arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/_build/pk_overlap_hazard tools/pk_overlap_hazard.hip -pthread
//   tools/_build/pk_overlap_hazard [seconds per case = 5]
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

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float unit(unsigned s) { return __builtin_bit_cast(float, (s & 0x007fffffu) | 0x3f800000u) - 1.5f; }
__device__ __forceinline__ bool differ(float a, float b) { return __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b); }

// CASE 0: in-place v_pk_fma_f32 with crossed halves of source 0; 1: in-place v_pk_mov_b32; 2: the same fma NOT in place (control);
// 4: a chain of 0 -> 1 -> 0 as in the kernel
template <int CASE>
__global__ __launch_bounds__(256) void k_victim(unsigned seed, unsigned iters, unsigned* __restrict__ hist, unsigned long long* __restrict__ checks)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned s = mix(seed * 0x9e3779b9u + blockIdx.x * 256u + threadIdx.x);
    unsigned bad = 0;
    for (unsigned i = 0; i < iters; i++) {
        s = mix(s + i);
        const f2 a = {unit(s), unit(mix(s ^ 1u))}, b = {unit(mix(s ^ 2u)), unit(mix(s ^ 3u))}, c = {unit(mix(s ^ 4u)), unit(mix(s ^ 5u))};
        f2 d = a, e;
        if (CASE == 0 || CASE == 4) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(b), "v"(c));
            e = f2{__builtin_fmaf(a.y, b.x, c.x), __builtin_fmaf(a.x, b.x, c.y)};
        }
        if (CASE == 1 || CASE == 4) {
            const f2 d0 = CASE == 4 ? e : a;
            asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(d) : "v"(b));
            e = f2{b.y, d0.x};
        }
        if (CASE == 4) {
            const f2 d0 = e;
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(c), "v"(b));
            e = f2{__builtin_fmaf(d0.y, c.x, b.x), __builtin_fmaf(d0.x, c.x, b.y)};
        }
        if (CASE == 5 || CASE == 6) {
            // the instruction the assembly bisection of the failing build ended at (tools/asm_patch_experiment.sh):
            //     v_pk_fma_f32 v[54:55], v[88:89], v[16:17], v[54:55] op_sel:[0,1,0]      D = S0 * S1.hi + D, accumulator in place,
            // its accumulator halves written by a v_mul_f32 and a v_mov_b32 just before (CASE 6 reproduces that too)
            f2 acc = c;
            float spare;
            if (CASE == 6) asm volatile("v_mul_f32 %0, %3, %4\n\tv_mov_b32 %1, %5\n\tv_mul_f32 %2, %3, %3" : "=&v"(acc.x), "=&v"(acc.y), "=&v"(spare) : "v"(b.x), "v"(a.y), "v"(b.y));
            const f2 acc0 = CASE == 6 ? f2{b.x * a.y, b.y} : c;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
            d = acc;
            e = f2{__builtin_fmaf(a.x, b.y, acc0.x), __builtin_fmaf(a.y, b.y, acc0.y)};
        }
        if (CASE == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            e = f2{__builtin_fmaf(a.y, b.x, c.x), __builtin_fmaf(a.x, b.x, c.y)};
        }
        bad += differ(d.x, e.x) | differ(d.y, e.y);
    }
    if (bad) atomicAdd(&hist[lane], bad);
    if (lane == 0 && (threadIdx.x >> 6) == 0) atomicAdd(checks, (unsigned long long)iters * 256ull);
}

__global__ void k_aggressor(unsigned long long cycles, unsigned* sink)      // the stress harness' k_spin
{
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) acc = acc * 1664525u + 1013904223u;
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

template <int CASE>
static void run_case(const char* what, double seconds, bool aggressor, unsigned* d_hist, unsigned long long* d_checks, unsigned* d_sink)
{
    CHK(hipMemset(d_hist, 0, 64 * 4));
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
                const unsigned long long cyc = 2000ull + (unsigned long long)((n * 2654435761u) >> 20);
                hipLaunchKernelGGL(k_aggressor, dim3(256), dim3(64), 0, sa, cyc, d_sink);
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
        for (int k = 0; k < 8; k++) hipLaunchKernelGGL(k_victim<CASE>, dim3(8100), dim3(256), 0, nullptr, launch++, 300u, d_hist, d_checks);
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
    unsigned hist[64];
    unsigned long long checks = 0;
    CHK(hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&checks, d_checks, 8, hipMemcpyDeviceToHost));
    unsigned long long bad = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; l++) { bad += hist[l]; q[l / 16] += hist[l]; }
    std::printf("%-78s aggressor %d: %.3g checks, %llu wrong (lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu)\n", what, aggressor ? 1 : 0, (double)checks, bad,
                q[0], q[1], q[2], q[3]);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 5.0;
    CHK(hipSetDevice(0));
    unsigned *d_hist, *d_sink;
    unsigned long long* d_checks;
    CHK(hipMalloc(&d_hist, 64 * 4));
    CHK(hipMalloc(&d_checks, 8));
    CHK(hipMalloc(&d_sink, 64));
    run_case<0>("v_pk_fma_f32 in place, source 0 with crossed halves", seconds, false, d_hist, d_checks, d_sink);
    run_case<0>("v_pk_fma_f32 in place, source 0 with crossed halves", seconds, true, d_hist, d_checks, d_sink);
    run_case<1>("v_pk_mov_b32 in place (D.lo = S0.hi, D.hi = old D.lo)", seconds, true, d_hist, d_checks, d_sink);
    run_case<2>("v_pk_fma_f32 with crossed halves, separate destination (control)", seconds, true, d_hist, d_checks, d_sink);
    run_case<4>("chain: in-place fma -> in-place mov -> in-place fma", seconds, true, d_hist, d_checks, d_sink);
    run_case<5>("v_pk_fma_f32 D, S0, S1, D op_sel:[0,1,0] (the bisection's instruction)", seconds, true, d_hist, d_checks, d_sink);
    run_case<6>("... behind v_mul_f32 / v_mov_b32 writes of its accumulator halves", seconds, true, d_hist, d_checks, d_sink);
    return 0;
}
