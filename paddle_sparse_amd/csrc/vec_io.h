// 16-byte vector loads / stores (plain and non-temporal) and 64-bit lane
// shuffles shared by the SpMM kernels (gfx950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psa {

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<2> {
  using T = float2;
};
template <>
struct Vec<4> {
  using T = float4;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&dst)[VEC]) {
  using T = typename Vec<VEC>::T;
  const T v = *reinterpret_cast<const T*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) dst[i] = f[i];
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&src)[VEC]) {
  using T = typename Vec<VEC>::T;
  T v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = src[i];
  *reinterpret_cast<T*>(p) = v;
}

// Non-temporal stores for the outputs (out, arg_out, arg_bytes, grad_value): they
// are written once and never read by the writing kernel.  Measured on config 3
// (one process, interleaved, variant 17 = ordinary stores): spmm_sum 1.684 ->
// 1.606 ms (0.86 -> 0.90 of the HBM peak), spmm_max 2.157 -> 2.036 ms.  The PMC
// traffic is the same both ways (9.09 GB fetched, 1.02 GB written, TCC hit rate
// 8.5 %): the write stream travels better beside the gathers, B is not cached
// any better.  Marking the col / value loads the same way changed nothing.
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* p, float (&dst)[VEC]) {
  if constexpr (VEC == 1) {
    dst[0] = __builtin_nontemporal_load(p);
  } else {
    typedef float V __attribute__((ext_vector_type(VEC)));
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#pragma unroll
    for (int i = 0; i < VEC; ++i) dst[i] = v[i];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec_nt(float* p, const float (&src)[VEC]) {
  if constexpr (VEC == 1) {
    __builtin_nontemporal_store(src[0], p);
  } else {
    typedef float V __attribute__((ext_vector_type(VEC)));
    V v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = src[i];
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
  }
}

template <int VEC>
__device__ __forceinline__ void store_arg_nt(int64_t* p, const int64_t (&src)[VEC]) {
  if constexpr (VEC == 1) {
    __builtin_nontemporal_store(src[0], p);
  } else {
    typedef long V __attribute__((ext_vector_type(VEC)));
    V v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = src[i];
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
  }
}

// Row-local form of arg_out, the operand of the one-pass min/max backward (it compares the
// entry of (row, k) with a per-edge tag of the same form instead of reading 8-byte edge ids):
//   width 1  (index & 127) | 0x80 on rows of more than 128 edges, where equal bytes only name a
//            candidate that is then tested against arg_out itself;
//   width 2  index & 0xffff: exact for rows of up to 65 535 edges, arg_out not needed at all.
// "No winner" (arg_out == nnz: an empty row, or a row none of whose products beat the init — all NaN,
// all -inf under max) is all ones in either width: local_index >= deg tells it (nnz - row start >= deg
// always), and no tag of an exact row equals it (tags of rows up to 128 edges are < 0x80, of rows up to
// 65 535 edges < 0xffff), so the one-pass backward routes nothing there — as the sentinel in arg_out does.
constexpr int kByteExact = 128;      // rows up to this many edges: width 1 is exact
constexpr int kWordExact = 65535;    // ... width 2 is exact

__device__ __forceinline__ uint32_t arg_local(int64_t local_index, int64_t deg, int width) {
  const uint32_t d = static_cast<uint32_t>(local_index);
  if (local_index >= deg) return width == 2 ? 0xffffu : 0xffu;
  return width == 2 ? (d & 0xffffu) : ((d & 127u) | (deg > kByteExact ? 0x80u : 0u));
}

// four consecutive entries (elem % 4 == 0) of the [M, K] array at `base`
__device__ __forceinline__ void store_arg_local4(uint8_t* base, int64_t elem, const uint32_t (&f)[4], int width) {
  if (width == 2) {
    const unsigned long long v = static_cast<unsigned long long>(f[0] | (f[1] << 16)) |
                                 (static_cast<unsigned long long>(f[2] | (f[3] << 16)) << 32);
    __builtin_nontemporal_store(v, reinterpret_cast<unsigned long long*>(base + 2 * elem));
  } else {
    __builtin_nontemporal_store(f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24), reinterpret_cast<uint32_t*>(base + elem));
  }
}

__device__ __forceinline__ void store_arg_local1(uint8_t* base, int64_t elem, uint32_t f, int width) {
  if (width == 2) reinterpret_cast<uint16_t*>(base)[elem] = static_cast<uint16_t>(f);
  else base[elem] = static_cast<uint8_t>(f);
}

__device__ __forceinline__ int64_t shfl_i64(int64_t x, int src) {
  return __shfl(static_cast<long long>(x), src);
}

}  // namespace psa
