// Shared host-side helpers for the C-ABI core (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "paddle_sparse_hip.h"

namespace psa {

// Thread-local message returned by psa_last_error().
void set_error(const std::string& msg);

inline hipStream_t as_stream(psa_stream_t s) {
  return reinterpret_cast<hipStream_t>(s);
}

constexpr int kWave = 64;  // gfx950 wavefront

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline bool aligned(const void* p, size_t a) {
  return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0;
}

// Zero-fill by a kernel of the core (4-byte aligned pointer and size).  Used
// instead of hipMemsetAsync everywhere: memset NODES of a captured HIP graph
// replay wrongly from the second launch on with the ROCm runtime torch 2.10
// ships (seen as wrong sort results on the second replay; tests/test_graph_capture_gpu.py
// now holds the replays to the eager results), kernels replay fine.
int zero_async(void* p, size_t bytes, hipStream_t s);

}  // namespace psa

#define PSA_REQUIRE(cond, msg)                                        \
  do {                                                                \
    if (!(cond)) {                                                    \
      psa::set_error(std::string(__func__) + ": " + (msg));           \
      return PSA_ERR_INVALID_ARG;                                     \
    }                                                                 \
  } while (0)

#define PSA_HIP(expr)                                                 \
  do {                                                                \
    hipError_t _e = (expr);                                           \
    if (_e != hipSuccess) {                                           \
      psa::set_error(std::string(__func__) + ": " #expr " failed: " + \
                     hipGetErrorString(_e));                          \
      return PSA_ERR_HIP;                                             \
    }                                                                 \
  } while (0)

#define PSA_ZERO(ptr, bytes, stream)                                  \
  do {                                                                \
    const int _z = psa::zero_async((ptr), (bytes), (stream));         \
    if (_z != PSA_OK) return _z;                                      \
  } while (0)

// Launch errors (bad configuration) surface through hipGetLastError.
#define PSA_LAUNCH_CHECK() PSA_HIP(hipGetLastError())
