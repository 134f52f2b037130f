// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate on this device (random operands), with and without one
// ds_read_b128 per MFMA, at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

template <int LDS_READS, int CHAIN>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* lw = (uint4*)smem;
    for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) lw[i] = make_uint4(0x3c003c00u + i, 0x38003c00u, 0x3c003800u + 7 * i, 0x34003c00u);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    half8 a[8], b;
    for (int j = 0; j < 8; j++) { uint4 v = lw[j * 64 + lane]; a[j] = __builtin_bit_cast(half8, v); }
    b = a[3];
    f32x16 acc[4];
    for (int q = 0; q < 4; q++) for (int e = 0; e < 16; e++) acc[q][e] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) {
            half8 aa = a[j];
            if (LDS_READS) { uint4 v = lw[j * 64 + lane]; aa = __builtin_bit_cast(half8, v); }
            if (CHAIN) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, acc[j & 1], 0, 0, 0);
            else acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b, acc[j & 3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int q = 0; q < 4; q++) for (int e = 0; e < 16; e++) s += acc[q][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int L, int C>
void run(const char* name, int threads, int blocks, int iters)
{
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<L, C>), dim3(blocks), dim3(threads), 8192, 0, out, iters, clk);
    hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k<L, C>), dim3(blocks), dim3(threads), 8192, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double mhz = (double)h[0] / (double)h[1] * 100.0;
    double flop = (double)blocks * (threads / 64) * iters * 8.0 * 32768.0;
    printf("%-34s threads %4d blocks %4d: %.3f ms  %.0f TFLOP/s  clock %.0f MHz  (%.1f%% of at-clock peak)\n", name, threads, blocks, ms,
           flop / ms / 1e9, mhz, flop / ms / 1e9 / (1024.0 * 1024.0 * mhz * 1e6 / 1e12) * 100.0);
    hipFree(out); hipFree(clk);
}

int main()
{
    const int it = 20000;
    run<0, 0>("regs, 4 independent acc", 256, 256, it);
    run<0, 0>("regs, 4 independent acc", 512, 256, it);
    run<0, 0>("regs, 4 independent acc", 1024, 256, it);
    run<0, 1>("regs, 2 chained acc", 256, 256, it);
    run<0, 1>("regs, 2 chained acc", 512, 256, it);
    run<1, 1>("1 ds_read_b128/MFMA, 2 chained", 256, 256, it);
    run<1, 1>("1 ds_read_b128/MFMA, 2 chained", 512, 256, it);
    run<1, 1>("1 ds_read_b128/MFMA, 2 chained", 1024, 256, it);
    run<1, 1>("1 ds_read_b128/MFMA, 2 chained", 512, 512, it);
    return 0;
}
