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
