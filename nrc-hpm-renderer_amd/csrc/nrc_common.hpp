// nrc_common.hpp -- shared host-side helpers of libnrc_hpm (error convention, HIP checks, config parsing).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/nrc_hpm.h"

namespace nrc {

// Reference error convention: Log::Error(msg, true) throws std::runtime_error("SkyRenderer ERROR: " + msg)
// (src/Log.cpp:16-20); ASSERT_CUDA does the same (include/engine/cuda_common.hpp:14).
[[noreturn]] inline void fail(const std::string& msg) { throw std::runtime_error("SkyRenderer ERROR: " + msg); }

struct HipError : std::runtime_error {
    explicit HipError(const std::string& m) : std::runtime_error("SkyRenderer ERROR: " + m) {}
};

#define NRC_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::nrc::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +     \
                                  __FILE__ + ":" + std::to_string(__LINE__) + ")");                     \
    } while (0)

// a collective failed, or a peer did not answer within the communicator's deadline: the communicator has been aborted (NRC_ERR_COMM)
struct CommError : std::runtime_error {
    explicit CommError(const std::string& m) : std::runtime_error("SkyRenderer ERROR: " + m) {}
};

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- THE environment switch of the product build: NRC_DEBUG="name[=value],name[=value],..." --------------------------------------------
// Diagnostic and test switches only -- nothing a host needs in order to use the library, nothing that changes a result (the tests that set
// them compare the two sides bit for bit).  Read when the object that consults it is created (a cache, a renderer), so a test can change
// it between two objects of one process.  (-DNRC_DIAG builds add the NRC_INFER_* shape sweeps of tools/bench_mlp.py.)
//   poison_alloc        every device allocation is filled with 0xFF bytes at creation (fp32 / fp16 NaN, index 0xFFFFFFFF): what a kernel reads
//                       without anyone having written it shows up instead of hiding behind the zeros of a freshly booted box
//   guard_alloc         4 KiB of 0xA5 canaries around every allocation; nrc_debug_check_guards() and every free verify them
//   assume_xcds=N       the code paths of a device that does not have eight XCDs (placements for speed are switched off)
//   single_stream       the renderer's frame on ONE stream in the reference's order (the four-stream graph must equal it bit for bit)
//   no_live_list        renderer inference over the whole query buffer instead of the frame's live-query list
//   zero_dead_queries   gen_rays writes the zero queries of pixels that did not scatter (the reference's zero-filled buffer)
//   fused_composite     compositing as the epilogue of the inference launch (6 x 64 model; measured slower in the frame)
//   dense_grid_exchange the HashGrid table gradient through the dense all-reduce instead of the all-gathered lists
//   no_fused_opt        three optimizer launches (k_adam_ema / k_sgd_ema, k_pack, k_pack_grid) instead of the fused one
//   grid_no_bins        every pair of the table gradient through the fixed-point shadow (no per-bin lists)
//   wgrad_old=0|1       round 3's k_wgrad (1) or k_wgrad2 (0) whatever the width;   train_gen_old=0|1, train_gen_nt=N: likewise k_train_gen
//   wave_priority_raise=0|1   nrc_set_wave_priority_raise for hosts that cannot call it (tests/cpp/stress_main)
inline bool debug_switch(const char* name, long* value = nullptr)
{
    const char* e = getenv("NRC_DEBUG");
    if (!e) return false;
    const size_t n = std::strlen(name);
    for (const char* p = e; *p;) {
        const char* end = std::strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : std::strlen(p);
        if (len >= n && std::strncmp(p, name, n) == 0 && (len == n || p[n] == '=')) {
            if (value) *value = len > n ? std::strtol(p + n + 1, nullptr, 10) : 1;
            return true;
        }
        p += len + (end ? 1 : 0);
    }
    return false;
}
inline long debug_value(const char* name, long fallback)
{
    long v = fallback;
    return debug_switch(name, &v) ? v : fallback;
}

// ---- events as part of a launch.  A hipEventRecord costs the host 3.5 us, a kernel launch 2.2 us -- and a launch that carries its own
// start / stop event (hipExtLaunchKernelGGL) 1.6 us (tools/ext_launch_probe.hip: the stop event orders other streams and carries a time
// stamp exactly like a recorded one).  A frame has eleven events behind ten launches.  The frame graph ARMS the event that marks the end of
// a stage (arm_launch_tail), the stage's LAST kernel launch goes through launch_last() and takes it along; a stage whose last launch does
// not (a path nobody marked) leaves it armed and finish_launch_tail() records it the ordinary way -- marking is an optimisation, never a
// condition of correctness.  One thread drives one renderer: the slot is thread-local.
struct LaunchTail {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchTail& launch_tail();      // (nrc_api.hip)
inline void arm_launch_tail(hipEvent_t stop, hipEvent_t start = nullptr) { launch_tail() = LaunchTail{start, stop}; }
template <class K, class... A>
inline void launch_last(K kernel, dim3 grid, dim3 block, uint32_t shmem, hipStream_t s, A... args)
{
    LaunchTail& t = launch_tail();
    if (t.start != nullptr || t.stop != nullptr) {
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, s, t.start, t.stop, 0u, args...);
        t = LaunchTail{};
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
    }
}
inline void finish_launch_tail(hipStream_t s)
{
    LaunchTail& t = launch_tail();
    const LaunchTail left = t;
    t = LaunchTail{};      // (cleared first: a failing record below must not leave the slot armed)
    if (left.start != nullptr) NRC_HIP(hipEventRecord(left.start, s));      // nobody launched: an empty interval, but a recorded event
    if (left.stop != nullptr) NRC_HIP(hipEventRecord(left.stop, s));
}
// An armed tail must not outlive the stage it was armed for: an exception between arm and finish would hand its events to the next
// launch_last of this thread -- another stream, another renderer.  Stages therefore arm through a scope; its end records what no launch took.
struct LaunchTailScope {
    hipStream_t s;
    bool open = false;
    explicit LaunchTailScope(hipStream_t s_) : s(s_) {}
    LaunchTailScope(hipStream_t s_, hipEvent_t stop, hipEvent_t start = nullptr) : s(s_) { arm(stop, start); }
    LaunchTailScope(const LaunchTailScope&) = delete;
    LaunchTailScope& operator=(const LaunchTailScope&) = delete;
    void arm(hipEvent_t stop, hipEvent_t start = nullptr) { arm_launch_tail(stop, start); open = true; }
    void finish() { if (open) { open = false; finish_launch_tail(s); } }
    ~LaunchTailScope() { if (open) { try { finish_launch_tail(s); } catch (...) { launch_tail() = LaunchTail{}; } } }
};

// XCDs (accelerator complex dies, each with its own L2) of the current device.  Several launch mappings hand the work of one table level /
// one screen band to the workgroups of ONE XCD (workgroup b runs on XCD b mod count: round-robin dispatch) -- a placement for speed only,
// never for correctness; they are written for the MI355X's eight and are switched off on any other count (another partition mode, another chip).
inline int device_xcds()
{
    { long v = 0; if (debug_switch("assume_xcds", &v)) return (int)v; }      // tests: the paths of a device that does not have eight
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) return 0;
    return n;
}

// Every device allocation of the library goes through these two (nrc_api.hip); NRC_DEBUG=poison_alloc / guard_alloc (above) make them
// fill what they hand out with NaN bytes / fence it with canaries that nrc_debug_check_guards() and every dev_free verify.
void dev_alloc(void** p, size_t bytes, const char* what = "");
void dev_free(void* p);
template <class T>
inline void dev_alloc(T** p, size_t bytes, const char* what = "") { dev_alloc(reinterpret_cast<void**>(p), bytes, what); }

// O'Neill's pcg32 (XSH-RR) as tiny-cuda-nn vendors it (pcg32.h): its Trainer seeds pcg32{1337}, i.e. stream initseq = 1
struct Pcg32 {
    uint64_t state = 0, inc = 1;
    void seed(uint64_t init_state, uint64_t init_seq = 1u)
    {
        state = 0; inc = (init_seq << 1) | 1u; next(); state += init_state; next();
    }
    uint32_t next()
    {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    // pcg32::next_float: a float in [1, 2) from the top 23 bits (the MTGP trick), minus 1
    float nextf()
    {
        const uint32_t u = (next() >> 9) | 0x3f800000u;
        float f;
        __builtin_memcpy(&f, &u, 4);
        return f - 1.0f;
    }
};

}  // namespace nrc

// Every kernel of the library that can share a SIMD with k_gen_rays raises its waves to the SAME user wave priority, the highest
// (s_setprio 3).  Reason (tests/cpp/stress_main.cpp): a k_gen_rays wave that shared its SIMD with waves of a
// HIGHER issue priority now and then -- 2-3 % of 72-frame runs when the inference kernel alone is raised; twice in ~60 runs of
// round 2's build, whose streams differed in queue priority only -- left new_ray_dir with a different direction in lanes 48..63
// although every input was identical.  The code that did it was found later (packed FP32 instructions with operand swizzles that
// the SLP vectoriser had made of the second rotation; the integrator is compiled with -fno-slp-vectorize now, csrc/Makefile) and the
// kernel no longer fails in that arrangement; the common priority stays as the second guard -- it costs nothing against none
// (raising the camera kernels ALONE starves the side streams: frame 0.279 -> 0.298 ms).
//   bit: 1 inference / training kernels (nrc_mlp.hip), 2 k_composite, 4 train-ray kernels, 8 camera kernels, 16 helpers
//   -DNRC_DIAG_LOWPRIO=<mask>: DIAGNOSTIC builds leave those groups at the default priority 0 (8 = the configuration that fails)
#ifndef NRC_WAVE_PRIORITY
#define NRC_WAVE_PRIORITY 3
#endif
#ifndef NRC_DIAG_LOWPRIO
#define NRC_DIAG_LOWPRIO 0
#endif
// The run-time side of the guard (round 4): 1 = raise (default); 0 = every kernel of the library stays at the default priority -- still
// ONE priority for the whole library -- for a process whose FOREIGN kernels run beside it (RCCL's all-reduce on the training stream of a
// multi-GPU job, the runtime's fill and copy kernels): those cannot be raised, and at priority 0 under waves at 3 a 28 MB hipMemsetAsync
// took 184 us in the frame against 65 among equals (7 alone), while the frame rate of every preset is the same either way (8 008-8 039
// against 7 980 Msamples/s on the default one).  Set by nrc_set_wave_priority_raise; nrc_cache_comm_init lowers it for world > 1.
// One copy per translation unit (each .hip file is its own code object): mlp_ / integrator_set_wave_priority_raise write them.
#if defined(__HIPCC__)
namespace nrc {
static __device__ int g_raise_wave_priority = 1;
}
#ifdef NRC_NO_PRIORITY_SWITCH      // A/B build: the raise unconditional, as before the switch
#define NRC_RAISE_WAVE_PRIORITY(bit) do { if (!((NRC_DIAG_LOWPRIO) & (bit))) __builtin_amdgcn_s_setprio(NRC_WAVE_PRIORITY); } while (0)
#else
#define NRC_RAISE_WAVE_PRIORITY(bit)                                                                    \
    do {                                                                                                \
        if (!((NRC_DIAG_LOWPRIO) & (bit)) && nrc::g_raise_wave_priority != 0) __builtin_amdgcn_s_setprio(NRC_WAVE_PRIORITY); \
    } while (0)
#endif
#endif
