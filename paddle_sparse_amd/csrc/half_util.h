// Helpers shared by the half-width forward (spmm_half.hip) and the half-width passes over the CSC view
// (spmm_half_bw.hip): type tags and exact widening / one-rounding narrowing of 8 packed 2-byte floats.
// (Two translation units so that the build compiles them side by side: together they hold ~500 kernel
// instantiations and took 125 s of a 130 s build as one file.)
#pragma once

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

namespace psa_half {

enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

struct F16 {};
struct BF16 {};

// 8 packed 2-byte floats (one 16-byte load) -> 8 fp32, exactly
template <typename T>
__device__ __forceinline__ void widen8(const uint4& raw, float (&f)[8]) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (std::is_same<T, BF16>::value) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    } else {
      const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
      const float2 v = __half22float2(h);
      f[2 * i] = v.x;
      f[2 * i + 1] = v.y;
    }
  }
}

// 8 fp32 -> 8 packed 2-byte floats, round to nearest even
template <typename T>
__device__ __forceinline__ uint4 narrow8(const float (&f)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (std::is_same<T, BF16>::value) {
      const __hip_bfloat162 p = __float22bfloat162_rn(make_float2(f[2 * i], f[2 * i + 1]));
      w[i] = *reinterpret_cast<const uint32_t*>(&p);
    } else {
      const __half2 p = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
      w[i] = *reinterpret_cast<const uint32_t*>(&p);
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <typename T>
__device__ __forceinline__ float widen1(const void* p, int64_t i) {
  const uint16_t h = static_cast<const uint16_t*>(p)[i];
  if constexpr (std::is_same<T, BF16>::value) return __uint_as_float(static_cast<uint32_t>(h) << 16);
  else return __half2float(*reinterpret_cast<const __half*>(&h));
}

__device__ __forceinline__ int64_t shfl_i64(int64_t x, int src) {
  return __shfl(static_cast<long long>(x), src);
}

// test / bench hook shared by both files (psa_spmm_half_set_variant): 0 = default, 1 = several rows per wave for
// K <= 128, 2 = one row per wave with U = 8, 3 = no XCD mixing, 4 = 64-bit addressing forced
extern int g_half_variant;

}  // namespace psa_half
