// hipExtLaunchKernelGGL's stop event as a cross-stream dependency and as a timestamp, and what a launch costs the host with / without
// separate hipEventRecord calls:  hipcc --offload-arch=gfx950 -O2 tools/ext_launch_probe.hip -o tools/_build/ext_launch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(unsigned* p, unsigned v, int spin)
{
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
__global__ void consumer(const unsigned* p, unsigned* out, unsigned i) { out[i] = *p; }
__global__ void nop() {}
int main()
{
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned *p, *out;
    const int N = 2000;
    CK(hipMalloc(&p, 4)); CK(hipMalloc(&out, 4 * N)); CK(hipMemset(out, 0, 4 * N)); CK(hipMemset(p, 0, 4));
    hipEvent_t ev[8], back;
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));
    // 1) ordering: consumer i on stream b must see the value producer i wrote on stream a (stop event of the ext launch)
    for (int i = 0; i < N; i++) {
        if (i > 0) CK(hipStreamWaitEvent(a, back, 0));            // producer i+1 must not overwrite before consumer i has read
        hipExtLaunchKernelGGL(producer, dim3(64), dim3(64), 0, a, nullptr, ev[i & 7], 0, p, (unsigned)(i + 1), 2000);   // 20 us of spinning
        CK(hipStreamWaitEvent(b, ev[i & 7], 0));
        hipExtLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, nullptr, back, 0, (const unsigned*)p, out, (unsigned)i);
    }
    CK(hipDeviceSynchronize());
    static unsigned h[N];
    CK(hipMemcpy(h, out, 4 * N, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; i++) bad += h[i] != (unsigned)(i + 1);
    printf("cross-stream order through ext-launch stop events: %d of %d wrong\n", bad, N);
    // 2) timing events through the ext launch
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    hipExtLaunchKernelGGL(producer, dim3(64), dim3(64), 0, a, t0, t1, 0, p, 1u, 10000);      // 100 us
    CK(hipStreamSynchronize(a));
    float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
    printf("start/stop events of one ext launch around a 100 us kernel: %.1f us\n", ms * 1e3);
    // 3) host cost
    const int M = 20000;
    CK(hipDeviceSynchronize());
    auto c0 = std::chrono::steady_clock::now();
    for (int i = 0; i < M; i++) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a); (void)hipEventRecord(ev[i & 7], a); }
    auto c1 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    auto c2 = std::chrono::steady_clock::now();
    for (int i = 0; i < M; i++) hipExtLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr, ev[i & 7], 0);
    auto c3 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    auto c4 = std::chrono::steady_clock::now();
    for (int i = 0; i < M; i++) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a);
    auto c5 = std::chrono::steady_clock::now();
    CK(hipDeviceSynchronize());
    auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    printf("host per call: launch + hipEventRecord %.2f us, ext launch with stop event %.2f us, plain launch %.2f us\n", us(c0, c1) / M, us(c2, c3) / M, us(c4, c5) / M);
    return 0;
}
