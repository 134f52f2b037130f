// Microbenchmark behind DESIGN.md's k_gen_rays analysis, part 2: what a wave-wide gather of scattered BYTES costs on gfx950.
// One independent 1-byte load per lane per trip from a buffer of 32 KB (L1-resident) ... 128 MB, via the raw-buffer path the
// integrator uses or via global loads, with all 64 lanes or a subset active, from fully divergent or partly shared addresses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o tools/_build/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0 raw buffer byte, 1 global byte, 2 global dword
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ vol, uint32_t mask, int trips, uint32_t* out, uint32_t active_lanes,
                                                uint32_t share)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)vol, 0, (int)(mask + 1u), 0x00020000);
    const uint32_t lane = threadIdx.x & 63u;
    // `share` lanes in a row use the same random stream -> the same address (coalescible)
    uint32_t s = ((blockIdx.x * 256u + threadIdx.x) / share) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if (lane < active_lanes) {
        for (int t = 0; t < trips; t++) {
            uint32_t idx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { s = s * 1664525u + 1013904223u; idx[k] = (s >> 4) & mask; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (MODE == 0) acc += __builtin_amdgcn_raw_buffer_load_b8(r, (int)idx[k], 0, 0);
                if (MODE == 1) acc += vol[idx[k]];
                if (MODE == 2) acc += ((const uint32_t*)vol)[idx[k] >> 2];
            }
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, size_t bytes, uint32_t active, uint32_t share)
{
    uint8_t* vol; uint32_t* out;
    if (hipMalloc(&vol, bytes) != hipSuccess) return;
    (void)hipMemset(vol, 1, bytes);
    const int blocks = 256 * 8, trips = 64;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, vol, (uint32_t)(bytes - 1), trips, out, active, share);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, vol, (uint32_t)(bytes - 1), trips, out, active, share);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double lanes = (double)blocks * 4 * active * trips * 4, insts = (double)blocks * 4 * trips * 4;
    printf("%-14s %9.3f MB  %2u lanes  share %2u: %.3f ms  %6.1f G lane-gathers/s  %5.2f G wave-instr/s  (%.1f cycles/wave-instr/CU at 2.4 GHz)\n", name,
           bytes / 1048576.0, active, share, ms, lanes / ms / 1e6, insts / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / insts);
    (void)hipFree(vol); (void)hipFree(out);
}

int main()
{
    for (size_t b : {(size_t)32 << 10, (size_t)256 << 10, (size_t)4 << 20, (size_t)16 << 20, (size_t)128 << 20}) run<0>("raw buffer u8", b, 64, 1);
    for (uint32_t a : {32u, 16u, 8u}) run<0>("raw buffer u8", (size_t)16 << 20, a, 1);
    for (uint32_t sh : {2u, 4u, 16u, 64u}) run<0>("raw buffer u8", (size_t)16 << 20, 64, sh);
    run<1>("global u8", (size_t)32 << 10, 64, 1);
    run<1>("global u8", (size_t)16 << 20, 64, 1);
    run<2>("global dword", (size_t)32 << 10, 64, 1);
    run<2>("global dword", (size_t)16 << 20, 64, 1);
    return 0;
}
