// nrc_common.hpp -- shared host-side helpers of libnrc_hpm (error convention, HIP checks, config parsing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
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

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Every device allocation of the library goes through these two (nrc_api.hip).  Diagnostic environment switches, read once:
//   NRC_POISON_ALLOC=1  every allocation is filled with 0xFF bytes at creation (fp32 NaN, fp16 NaN, index 0xFFFFFFFF): whatever a
//                       kernel reads without anyone having written it shows up in the result instead of hiding behind the zeros
//                       of a freshly booted box;
//   NRC_GUARD_ALLOC=1   4 KiB of 0xA5 canary bytes in front of and behind every allocation; nrc_debug_check_guards() (and every
//                       dev_free) verifies them -- a kernel that stores outside its buffer is named by the allocation it ran over.
void dev_alloc(void** p, size_t bytes, const char* what = "");
void dev_free(void* p);
template <class T>
inline void dev_alloc(T** p, size_t bytes, const char* what = "") { dev_alloc(reinterpret_cast<void**>(p), bytes, what); }

// O'Neill's pcg32 (XSH-RR): the generator tiny-cuda-nn uses for weight init (seed 1337)
struct Pcg32 {
    uint64_t state = 0, inc = 1;
    void seed(uint64_t init_state, uint64_t init_seq)
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
    float nextf() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

}  // namespace nrc

// DIAGNOSTIC builds only (-DNRC_DIAG_SETPRIO=<mask>, tests/cpp/stress_main.cpp): side-stream kernels raise their waves' issue
// priority -- bit 0 the inference / training kernels (nrc_mlp.hip), bit 1 k_composite, bit 2 the train-ray kernels (k_train_scan,
// k_prep_train, k_ring_push).  With it, co-resident k_gen_rays waves misbehave a few per cent of the time;
// never defined in the product.
#ifdef NRC_DIAG_SETPRIO
#define NRC_RAISE_WAVE_PRIORITY(bit) do { if ((NRC_DIAG_SETPRIO) & (bit)) __builtin_amdgcn_s_setprio(3); } while (0)
#else
#define NRC_RAISE_WAVE_PRIORITY(bit) do { } while (0)
#endif
