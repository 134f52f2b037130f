// Microbenchmark behind DESIGN.md's k_gen_rays analysis: (1) issue cost of the VALU instruction classes the integrator is made
// of (v_fma_f32, v_pk_fma_f32, v_lshl_add_u32 / v_xor_b32 of the hash RNG, v_cvt, v_mad_u32_u24, v_sqrt_f32) at 1 / 2 / 4 / 8
// waves per SIMD; (2) rate of dependent-free random 1-byte gathers from a volume-sized buffer (16.8 MB / 134 MB), linear index.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k_valu(float* out, int iters, unsigned long long* clk)
{
    float a[8];
    uint32_t u[8];
    f2 p[8];
    for (int j = 0; j < 8; j++) { a[j] = 1.0f + 0.001f * (threadIdx.x + j); u[j] = threadIdx.x * 77u + j; p[j] = f2{a[j], a[j] * 0.5f}; }
    const float m = 0.9999f, c = 1e-6f;
    const f2 m2 = f2{m, m}, c2 = f2{c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(m2), "v"(c2));
                if (KIND == 2) asm volatile("v_lshl_add_u32 %0, %0, 10, %0" : "+v"(u[j]));
                if (KIND == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 4) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[j]) : "v"(a[j]));
                if (KIND == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[j]));
                if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(m2));
                if (KIND == 9) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
                if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 11) asm volatile("v_lshrrev_b32 %0, 6, %0" : "+v"(u[j]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 8; j++) s += a[j] + (float)u[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run_valu(const char* name)
{
    const int iters = 4000, blocks = 256;
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 2048 * 4); hipMalloc(&clk, blocks * 8);
    printf("%-18s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256 * wps > 512 ? 512 : 256 * wps, nblk = blocks * ((256 * wps) / threads);
        hipLaunchKernelGGL((k_valu<KIND>), dim3(nblk), dim3(threads), 0, 0, out, iters, clk);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k_valu<KIND>), dim3(nblk), dim3(threads), 0, 0, out, iters, clk);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nblk); hipMemcpy(h.data(), clk, nblk * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= nblk;
        // cycles of SIMD time per wave-instruction: a SIMD hosts `wps` waves, each issuing iters*32 instructions
        printf("  %dw/SIMD: %5.2f cyc/inst", wps, cyc / ((double)iters * 32.0 * wps));
    }
    printf("\n");
    hipFree(out); hipFree(clk);
}

// ---- random 1-byte gathers (independent, `INFLIGHT` per lane per trip)
template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ vol, uint32_t nvox_mask, int trips, uint32_t* out, int coherent)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t wave_seed = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 40503u;
    uint32_t acc = 0;
    for (int t = 0; t < trips; t++) {
        uint32_t idx[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) {
            s = s * 1664525u + 1013904223u;
            // coherent: the 64 lanes of a wave stay inside one 64 KB neighbourhood (a thin tube of the volume)
            idx[k] = coherent ? (((wave_seed + t * 977u) << 16) + (s >> 16)) & nvox_mask : (s >> 4) & nvox_mask;
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += vol[idx[k]];
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int INFLIGHT>
void run_gather(const char* name, size_t nvox, int coherent)
{
    uint8_t* vol; uint32_t* out;
    hipMalloc(&vol, nvox); hipMemset(vol, 1, nvox);
    const int blocks = 256 * 8, trips = 256 / INFLIGHT;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_gather<INFLIGHT>), dim3(blocks), dim3(256), 0, 0, vol, (uint32_t)(nvox - 1), trips, out, coherent);
    hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k_gather<INFLIGHT>), dim3(blocks), dim3(256), 0, 0, vol, (uint32_t)(nvox - 1), trips, out, coherent);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double n = (double)blocks * 256 * trips * INFLIGHT;
    printf("%-44s %6.1f MB  %d in flight: %.3f ms for %.0f M gathers = %.1f G gathers/s\n", name, nvox / 1048576.0, INFLIGHT, ms, n / 1e6, n / ms / 1e6);
    hipFree(vol); hipFree(out);
}

int main()
{
    run_valu<0>("v_fma_f32");
    run_valu<1>("v_pk_fma_f32");
    run_valu<7>("v_mul_f32");
    run_valu<8>("v_pk_mul_f32");
    run_valu<2>("v_lshl_add_u32");
    run_valu<3>("v_xor_b32");
    run_valu<11>("v_lshrrev_b32");
    run_valu<4>("v_cvt_u32_f32");
    run_valu<5>("v_mad_u32_u24");
    run_valu<9>("v_max3_u32");
    run_valu<10>("v_cndmask_b32");
    run_valu<6>("v_sqrt_f32");
    run_gather<1>("random bytes, whole buffer", (size_t)1 << 24, 0);
    run_gather<2>("random bytes, whole buffer", (size_t)1 << 24, 0);
    run_gather<4>("random bytes, whole buffer", (size_t)1 << 24, 0);
    run_gather<2>("random bytes, whole buffer", (size_t)1 << 27, 0);
    run_gather<2>("random bytes, 64 KB neighbourhood per wave-trip", (size_t)1 << 24, 1);
    run_gather<2>("random bytes, 64 KB neighbourhood per wave-trip", (size_t)1 << 27, 1);
    return 0;
}
