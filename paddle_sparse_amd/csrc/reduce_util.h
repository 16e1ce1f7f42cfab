// Accumulator types and the mean division of the segmented reducers
// (coalesce.hip, chain.hip).
#pragma once

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

namespace psa {

template <typename T>
struct Acc {
  using type = T;
  static __device__ type load(const T* p) { return *p; }
  static __device__ void store(T* p, type v) { *p = v; }
};
template <>
struct Acc<__half> {
  using type = float;
  static __device__ type load(const __half* p) { return __half2float(*p); }
  static __device__ void store(__half* p, type v) { *p = __float2half(v); }
};
template <>
struct Acc<__hip_bfloat16> {
  using type = float;
  static __device__ type load(const __hip_bfloat16* p) { return __bfloat162float(*p); }
  static __device__ void store(__hip_bfloat16* p, type v) { *p = __float2bfloat16(v); }
};

template <typename A>
__device__ __forceinline__ A mean_div(A acc, int64_t cnt) {
  if constexpr (std::is_integral<A>::value) {
    // pytorch_scatter: div_(count, rounding_mode="floor")
    A q = acc / static_cast<A>(cnt);
    if ((acc % static_cast<A>(cnt) != 0) && ((acc < 0) != (cnt < 0))) --q;
    return q;
  } else {
    return acc / static_cast<A>(cnt);
  }
}

}  // namespace psa
