// scatter(src, index, dim=0, dim_size, reduce) — column reductions.
//
// Stands in for paddle_scatter.scatter at paddle_sparse/reduce.py:40-42
// (SparseTensor.sum/mean/min/max(dim=0): values scattered by `col`).
// Semantics = pytorch_scatter: untouched rows -> 0, mean = sum / count
// (floor division for integer dtypes), min/max return values only.
// Atomics: one element per lane, consecutive lanes on consecutive elements of
// a row, so a wave instruction covers whole 256-B runs whenever D >= 64.
#include <type_traits>

#include "common.h"

namespace {

constexpr int kThreads = 256;
enum { R_SUM = 0, R_MEAN = 1, R_MIN = 2, R_MAX = 3 };

template <typename T>
struct Lim;
template <>
struct Lim<float> {
  static __device__ float lo() { return -__builtin_inff(); }
  static __device__ float hi() { return __builtin_inff(); }
};
template <>
struct Lim<double> {
  static __device__ double lo() { return -__builtin_inf(); }
  static __device__ double hi() { return __builtin_inf(); }
};
template <>
struct Lim<int32_t> {
  static __device__ int32_t lo() { return INT32_MIN; }
  static __device__ int32_t hi() { return INT32_MAX; }
};
template <>
struct Lim<long long> {
  static __device__ long long lo() { return INT64_MIN; }
  static __device__ long long hi() { return INT64_MAX; }
};

template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
scatter_init_kernel(T* __restrict__ out, int64_t total) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= total) return;
  out[g] = RED == R_MIN ? Lim<T>::hi() : (RED == R_MAX ? Lim<T>::lo() : T(0));
}

template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
scatter_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
               int64_t n, int64_t D, int64_t dim_size, T* __restrict__ out,
               unsigned int* __restrict__ count) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= n * D) return;
  const int64_t i = D == 1 ? g : g / D;
  const int64_t d = D == 1 ? 0 : g - i * D;
  const int64_t r = index[i];
  if (r < 0 || r >= dim_size) return;  // never store out of bounds
  T* dst = out + r * D + d;
  const T x = src[g];
  if (RED == R_MIN)
    __hip_atomic_fetch_min(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (RED == R_MAX)
    __hip_atomic_fetch_max(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_fetch_add(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (count && d == 0) atomicAdd(count + r, 1u);
}

template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
scatter_finish_kernel(T* __restrict__ out, const unsigned int* __restrict__ count,
                      int64_t dim_size, int64_t D) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= dim_size * D) return;
  const int64_t r = D == 1 ? g : g / D;
  const unsigned int c = count[r];
  if (c == 0) {
    out[g] = T(0);
  } else if (RED == R_MEAN) {
    const T a = out[g];
    if constexpr (std::is_integral<T>::value) {
      T q = a / static_cast<T>(c);
      if ((a % static_cast<T>(c) != 0) && (a < 0)) --q;  // floor
      out[g] = q;
    } else {
      out[g] = a / static_cast<T>(c);
    }
  }
}

template <typename T, int RED>
int run(const void* src, const int64_t* index, int64_t n, int64_t D,
        int64_t dim_size, void* out, unsigned int* count, hipStream_t s) {
  T* o = static_cast<T*>(out);
  const int64_t total = dim_size * D;
  const int64_t ob = psa::ceil_div(total, kThreads);
  const int64_t nb = psa::ceil_div(n * D, kThreads);
  PSA_REQUIRE(ob <= 0x7fffffff && nb <= 0x7fffffff, "too many elements for one launch");
  hipLaunchKernelGGL((scatter_init_kernel<T, RED>), dim3(static_cast<unsigned>(ob)),
                     dim3(kThreads), 0, s, o, total);
  if (n > 0) {
    hipLaunchKernelGGL((scatter_kernel<T, RED>), dim3(static_cast<unsigned>(nb)),
                       dim3(kThreads), 0, s, static_cast<const T*>(src), index, n,
                       D, dim_size, o, RED == R_SUM ? nullptr : count);
  }
  if (RED != R_SUM) {
    hipLaunchKernelGGL((scatter_finish_kernel<T, RED>), dim3(static_cast<unsigned>(ob)),
                       dim3(kThreads), 0, s, o, count, dim_size, D);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

template <typename T>
int dispatch(int reduce, const void* src, const int64_t* index, int64_t n,
             int64_t D, int64_t dim_size, void* out, unsigned int* count,
             hipStream_t s) {
  switch (reduce) {
    case PSA_SUM: return run<T, R_SUM>(src, index, n, D, dim_size, out, count, s);
    case PSA_MEAN: return run<T, R_MEAN>(src, index, n, D, dim_size, out, count, s);
    case PSA_MIN: return run<T, R_MIN>(src, index, n, D, dim_size, out, count, s);
    default: return run<T, R_MAX>(src, index, n, D, dim_size, out, count, s);
  }
}

}  // namespace

extern "C" {

size_t psa_scatter_workspace_bytes(int64_t dim_size) {
  return static_cast<size_t>(dim_size > 0 ? dim_size : 1) * sizeof(unsigned int);
}

int psa_scatter_reduce(int reduce, int dtype, const void* src,
                       const int64_t* index, int64_t n, int64_t D,
                       int64_t dim_size, void* out, void* workspace,
                       size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(n >= 0 && D >= 0 && dim_size >= 0, "negative size");
  if (dim_size == 0 || D == 0) return PSA_OK;
  PSA_REQUIRE(out != nullptr, "out is NULL");
  PSA_REQUIRE(n == 0 || (src && index), "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  unsigned int* count = nullptr;
  if (reduce != PSA_SUM) {
    if (workspace == nullptr || workspace_bytes < psa_scatter_workspace_bytes(dim_size)) {
      psa::set_error("psa_scatter_reduce: workspace too small");
      return PSA_ERR_WORKSPACE;
    }
    count = static_cast<unsigned int*>(workspace);
    PSA_ZERO(count, sizeof(unsigned int) * dim_size, s);
  }
  switch (dtype) {
    case PSA_F32: return dispatch<float>(reduce, src, index, n, D, dim_size, out, count, s);
    case PSA_F64: return dispatch<double>(reduce, src, index, n, D, dim_size, out, count, s);
    case PSA_I32: return dispatch<int32_t>(reduce, src, index, n, D, dim_size, out, count, s);
    case PSA_I64: return dispatch<long long>(reduce, src, index, n, D, dim_size, out, count, s);
    default:
      psa::set_error("psa_scatter_reduce: dtype not supported (f32/f64/i32/i64)");
      return PSA_ERR_UNSUPPORTED;
  }
}

}  // extern "C"
