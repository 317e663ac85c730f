// Shared host-side helpers of libbpmf_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

namespace bpmf {

// Thread-local last error text returned by bpmf_last_error().
char* last_error_buf();
void set_error(const char* fmt, ...);

constexpr int CSUM_CHUNK = 1024;             // spec constant, see oracle/bpmf_oracle.c
constexpr float MAX_NORM = 1000.0f;          // r_t * r_d >= this (E_t*E_d <~ 1e-6) -> CC = 0

// bench hook: events around the dominant kernels (util.hip)
void profile_mark(int which, int edge, hipStream_t stream);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace bpmf

#define BPMF_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            bpmf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return -2;                                                                \
        }                                                                             \
    } while (0)

#define BPMF_LAUNCH_CHECK() BPMF_HIP_CHECK(hipGetLastError())
