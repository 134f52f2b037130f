// the lane-scan form of a sequential recurrence (ratio_groups / delta_groups) against the plain loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__host__ __device__ inline uint32_t hash1(uint32_t x) { x += (x << 10); x ^= (x >> 6); x += (x << 3); x ^= (x >> 11); x += (x << 15); return x; }
__host__ __device__ inline float fc(uint32_t m) { uint32_t u = (m & 0x007fffffu) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f; }
__host__ __device__ inline float random1(float x) { uint32_t u; memcpy(&u, &x, 4); return fc(hash1(u)); }
__device__ inline float lane_below(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
}
__global__ void k(float* out, uint32_t L, uint32_t Lc, float R0)
{
    const uint32_t lane = threadIdx.x, j = lane & (L - 1u);
    const bool first = j == 0u;
    float s1 = R0;
    for (uint32_t d = 0; d < Lc; d++) { const float lb = lane_below(s1); s1 = random1(first ? R0 : lb); }      // (`first ? R0 : lane_below(s1)` would evaluate the move on the other lanes only)
    out[lane] = s1;
}
int main()
{
    float* d; float h[64];
    hipMalloc(&d, 256);
    int bad = 0;
    for (uint32_t L = 2; L <= 32; L *= 2)
        for (uint32_t Lc = 2; Lc <= L; Lc *= 2) {
            k<<<1, 64>>>(d, L, Lc, 0.37f);
            hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
            for (int lane = 0; lane < 64; lane++) {
                uint32_t j = lane & (L - 1);
                if (j >= Lc) continue;
                float r = 0.37f;
                for (uint32_t q = 0; q <= j; q++) r = random1(r);
                if (memcmp(&r, &h[lane], 4)) { if (bad < 10) printf("L %u Lc %u lane %d: %g want %g\n", L, Lc, lane, h[lane], r); bad++; }
            }
        }
    printf("mismatches: %d\n", bad);
    return 0;
}
