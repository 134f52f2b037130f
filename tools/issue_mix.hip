// Microbenchmark behind DESIGN.md's round-4 analysis of k_gen_rays: does a SIMD's instruction issue have ONE budget for everything a wave
// issues (vector, scalar, s_nop), or do scalar instructions and wait states ride along with the vector stream of other waves?
// Streams of 32 instructions per unit at 1 / 2 / 4 / 5 / 8 waves per SIMD; printed: shader clocks (s_memtime) of SIMD time per unit.
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_mix.hip -o tools/_build/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define V8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, unsigned long long* clk)
{
    float a[8];
    f2 p[8];
    uint32_t u[8];
    for (int j = 0; j < 8; j++) { a[j] = 1.0f + 0.001f * (threadIdx.x + j); p[j] = f2{a[j], a[j] * 0.5f}; u[j] = threadIdx.x * 77u + j; }
    const float m = 0.9999f, c = 1e-6f;
    const f2 m2 = f2{m, m}, c2 = f2{c, c};
    unsigned long long s0 = 0x5555, s1 = 0x3333;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));                       // 32 independent-ish (8 chains)
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_and_b64 %1, %1, %4" : "+v"(a[j]), "+s"(s0) : "v"(m), "v"(c), "s"(s1) : "scc");      // + 32 SALU
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(a[j]) : "v"(m), "v"(c));            // + 32 s_nop 0
                if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));                       // ONE dependent chain
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(p[0]) : "v"(m2), "v"(c2));      // dependent packed chain + the compiler's wait state
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(m2), "v"(c2));                  // packed, 8 chains
                if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 10, %0\n\tv_lshrrev_b32 %1, 6, %0\n\tv_xor_b32 %0, %1, %0" : "+v"(u[0]), "=&v"(u[1]));   // hash-like dependent chain (3 per unit slot)
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_and_b64 %1, %1, %4\n\ts_or_b64 %1, %1, %4" : "+v"(a[j]), "+s"(s0) : "v"(m), "v"(c), "s"(s1) : "scc");   // + 64 SALU
                if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "s"(s1));     // select on an SGPR mask
                if (KIND == 9) asm volatile("v_cmp_gt_f32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(a[j]), "=&s"(s0) : "v"(m));      // compare -> SGPR mask -> select
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)(s0 & 1);
    for (int j = 0; j < 8; j++) s += a[j] + p[j].x + p[j].y + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int vec_per_unit)
{
    const int iters = 2000, cus = 256;
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4); hipMalloc(&clk, cus * 8 * 8);
    printf("%-44s", name);
    fflush(stdout);
    for (int wps : {1, 2, 4, 5, 8}) {
        const int nblk = cus * wps;          // 256-thread blocks = one wave per SIMD each; wps blocks per CU when spread evenly
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL((k_mix<KIND>), dim3(nblk), dim3(256), 0, 0, out, iters, clk);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(nblk); hipMemcpy(h.data(), clk, nblk * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= nblk;
        // clocks of SIMD time per unit (one wave's `vec_per_unit` vector instructions + whatever rides with them)
        printf("  %dw %6.2f", wps, cyc / ((double)iters * 32.0 * wps));
        fflush(stdout);
    }
    printf("   (clocks of SIMD time per unit of %d vector instruction%s)\n", vec_per_unit, vec_per_unit > 1 ? "s" : "");
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("v_fma_f32, 8 chains", 1);
    run<1>("v_fma_f32 + s_and_b64", 1);
    run<7>("v_fma_f32 + 2 SALU", 1);
    run<2>("v_fma_f32 + s_nop 0", 1);
    run<3>("v_fma_f32, one dependent chain", 1);
    run<5>("v_pk_fma_f32, 8 chains", 1);
    run<4>("v_pk_fma_f32 dependent + s_nop 0", 1);
    run<6>("hash step (3 dependent int ops)", 3);
    run<8>("v_cndmask_b32 on an SGPR mask", 1);
    run<9>("v_cmp -> SGPR -> v_cndmask", 2);
    return 0;
}
