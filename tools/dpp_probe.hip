// what do the DPP cross-lane controls do on this GPU?  hipcc --offload-arch=gfx950 tools/dpp_probe.hip -o tools/_build/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void k(int* out) { out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, (int)threadIdx.x, CTRL, 0xf, 0xf, false); }
template <int CTRL>
void run(const char* name)
{
    int* d; int h[64];
    hipMalloc(&d, 256);
    k<CTRL><<<1, 64>>>(d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-12s", name);
    for (int i = 0; i < 64; i++) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
int main()
{
    run<0x138>("wave_shr:1");
    run<0x111>("row_shr:1");
    run<0x142>("row_bcast15");
    run<0x143>("row_bcast31");
    run<0x13C>("wave_ror:1");
    return 0;
}
