// Micro-reproducer for the stale scratch reload behind round 2's non-determinism.
// k_gen_rays spilled one value (8 bytes per lane); under co-residency with high-priority waves of other queues the reload returned,
// in lanes 48..63, what ANOTHER wave had spilled to the same scratch slot.  This program asks whether that needs our renderer at
// all: a "victim" kernel (default-priority stream, the launch shape of k_gen_rays: 4-wave workgroups, waves of very unequal
// duration) writes a per-wave, per-lane signature to its private (scratch) memory, works for a while, reads it back and counts
// the lanes whose value changed; an "aggressor" thread keeps short kernels with raised wave priority (s_setprio 3), MFMA work and
// LDS in flight on a high-priority stream.
//
//   hipcc --offload-arch=gfx950 -O3 -o scratch_repro tools/scratch_repro.hip -pthread
//   ./scratch_repro [seconds=20] [aggressor 0|1=1] [victim blocks=8100]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

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

struct Bad {
    unsigned wave, lane, round, got0, got1, want0, want1, launch;
};

// the spill's own instructions (inline asm) on the kernel's private segment
__global__ __launch_bounds__(256) void k_victim(unsigned launch, unsigned rounds, unsigned* __restrict__ hist, unsigned* __restrict__ n_bad,
                                                Bad* __restrict__ bad, const unsigned* __restrict__ noise, unsigned n_noise)
{
    volatile unsigned priv[3];      // forces a private segment (12 bytes per lane, k_gen_rays' size) and the scratch set-up
    priv[2] = launch;
    const unsigned lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned acc = wave;
    for (unsigned r = 0; r < rounds; r++) {
        const unsigned s0 = mix(launch * 0x9e3779b9u + wave * 64u + lane + r * 7919u), s1 = ~s0;
        {   // exactly the spill instruction k_gen_rays had: scratch_store_dwordx2 off, v[a:b], off
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 v = {s0, s1};
            asm volatile("scratch_store_dwordx2 off, %0, off" ::"v"(v) : "memory");
        }
        // unequal work between store and reload: 0 .. ~60 us of dependent integer work with a few scattered loads, like a path walk
        const unsigned spins = (mix(wave * 31u + r) & 0xfffu) + 16u;
        for (unsigned i = __builtin_amdgcn_readfirstlane(0); i < __builtin_amdgcn_readfirstlane(spins); i++) {
            acc = mix(acc + i);
            if ((i & 63u) == 0u) acc += noise[acc % n_noise];
        }
        unsigned g0, g1;
        {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 v;
            asm volatile("scratch_load_dwordx2 %0, off, off\n\ts_waitcnt vmcnt(0)" : "=v"(v)::"memory");
            g0 = v.x; g1 = v.y;
        }
        if (g0 != s0 || g1 != s1) {
            atomicAdd(&hist[lane], 1u);
            const unsigned k = atomicAdd(n_bad, 1u);
            if (k < 256u) bad[k] = Bad{wave, lane, r, g0, g1, s0, s1, launch};
        }
    }
    if (acc == 0x12345678u) hist[0] = acc;
}

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_aggressor(unsigned long long cycles, float* __restrict__ sink)
{
    __shared__ float lds[12 * 1024];
    __builtin_amdgcn_s_setprio(3);
    for (int i = threadIdx.x; i < 12 * 1024; i += 512) lds[i] = (float)i;
    __syncthreads();
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.01f * (threadIdx.x + j)); b[j] = (_Float16)(0.02f * j); }
    f32x16 c;
    for (int j = 0; j < 16; j++) c[j] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned k = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        k = (k * 1664525u + 1013904223u);
        c[0] += lds[k % (12u * 1024u)];
    }
    if (c[0] == 123.456f) sink[0] = c[1];
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 20.0;
    const int aggressor = argc > 2 ? std::atoi(argv[2]) : 1;
    const unsigned blocks = argc > 3 ? (unsigned)std::atoi(argv[3]) : 8100u;
    CHK(hipSetDevice(0));
    unsigned *d_hist, *d_nbad, *d_noise;
    Bad* d_bad;
    float* d_sink;
    const unsigned n_noise = 1u << 22;
    CHK(hipMalloc(&d_hist, 64 * 4)); CHK(hipMemset(d_hist, 0, 64 * 4));
    CHK(hipMalloc(&d_nbad, 4)); CHK(hipMemset(d_nbad, 0, 4));
    CHK(hipMalloc(&d_bad, 256 * sizeof(Bad)));
    CHK(hipMalloc(&d_noise, n_noise * 4)); CHK(hipMemset(d_noise, 1, n_noise * 4));
    CHK(hipMalloc(&d_sink, 64));
    std::atomic<bool> stop{false};
    unsigned long aggr_launches = 0;
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
                const unsigned long long cyc = 1000ull + ((n * 2654435761u) >> 19);      // 10 .. 90 us on the 100 MHz clock
                hipLaunchKernelGGL(k_aggressor, dim3(256), dim3(512), 0, sa, cyc, d_sink);
                if ((++n & 7u) == 0) CHK(hipStreamSynchronize(sa));
                aggr_launches++;
            }
            CHK(hipStreamSynchronize(sa));
        });
    }
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned launch = 0;
    double elapsed = 0.0;
    CHK(hipEventRecord(e0, nullptr));
    while (elapsed < seconds * 1e3) {
        for (int k = 0; k < 16; k++) hipLaunchKernelGGL(k_victim, dim3(blocks), dim3(256), 0, nullptr, launch++, 6u, d_hist, d_nbad, d_bad, d_noise, n_noise);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        elapsed = ms;
    }
    stop.store(true);
    if (th.joinable()) th.join();
    CHK(hipDeviceSynchronize());
    unsigned hist[64], nbad = 0;
    CHK(hipMemcpy(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&nbad, d_nbad, 4, hipMemcpyDeviceToHost));
    std::vector<Bad> bad(256);
    CHK(hipMemcpy(bad.data(), d_bad, 256 * sizeof(Bad), hipMemcpyDeviceToHost));
    const double checks = (double)launch * blocks * 256.0 * 6.0;
    std::printf("victim launches %u (%u blocks x 256 threads x 6 store/reload rounds) in %.1f s, aggressor %s (%lu launches): %u stale reloads of %.3g lane-checks\n",
                launch, blocks, elapsed * 1e-3, aggressor ? "on" : "off", aggr_launches, nbad, checks);
    if (nbad) {
        std::printf("by lane:");
        for (int l = 0; l < 64; l++) std::printf(" %u", hist[l]);
        std::printf("\n");
        for (unsigned k = 0; k < nbad && k < 24; k++) {
            const Bad& b = bad[k];
            // whose signature is it?  the same lane of another wave / round / launch has signature mix(launch * c + wave * 64 + lane + round * 7919)
            std::printf("  launch %u wave %u lane %u round %u: got %08x %08x want %08x %08x (got1 == ~got0: %s)\n", b.launch, b.wave, b.lane, b.round, b.got0, b.got1,
                        b.want0, b.want1, b.got1 == ~b.got0 ? "yes: a complete signature of another wave/round" : "no");
        }
    }
    return nbad ? 1 : 0;
}
