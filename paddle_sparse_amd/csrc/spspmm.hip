// Sparse x sparse product C = A @ B (README.md:308-353 `spspmm`), expand step.
//
// The product is formed the expand / sort / compress way so that it runs on
// the path's own kernels: every stored A entry e = (i, c, a) meets every
// stored B entry (c, j, b) of B's row c and emits the pair (i * n + j, a * b);
// psa_sort_pairs_u32 / psa_index_sort order the pairs by key (stable), and
// psa_unique_* + psa_segment_reduce add the runs.  Emission order is A's
// storage order and, inside one A entry, B's storage order, so with the
// stable sort the terms of each C entry arrive in the order a sequential
// row-by-row (Gustavson) product adds them.
//
// Work is one thread per PRODUCT, not per A entry: `owner[p]` (psa_ptr2ind of
// the product offsets) names the A entry of product p, so rows of B of any
// length spread evenly and both output streams are written coalesced.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// counts[e] = number of stored entries in B's row colA[e].
__global__ void __launch_bounds__(kThreads)
spspmm_count_kernel(const int64_t* __restrict__ colA, int64_t nnzA,
                    const int64_t* __restrict__ rowptrB,
                    int64_t* __restrict__ counts) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= nnzA) return;
  const int64_t c = colA[e];
  counts[e] = rowptrB[c + 1] - rowptrB[c];
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
spspmm_expand_kernel(const int64_t* __restrict__ rowA, const int64_t* __restrict__ colA,
                     const T* __restrict__ valA, const int64_t* __restrict__ rowptrB,
                     const int64_t* __restrict__ colB, const T* __restrict__ valB,
                     const int64_t* __restrict__ offsets, const int64_t* __restrict__ owner,
                     int64_t total, int64_t n, int64_t* __restrict__ keys,
                     T* __restrict__ vals) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (p >= total) return;
  const int64_t e = owner[p];
  const int64_t q = rowptrB[colA[e]] + (p - offsets[e]);
  // n < 0: packed form (inner << 32) | outer, for the walk over the CSC views
  keys[p] = n < 0 ? ((colB[q] << 32) | rowA[e]) : rowA[e] * n + colB[q];
  if (vals) vals[p] = (valA ? valA[e] : T(1)) * (valB ? valB[q] : T(1));
}

template <typename T>
int launch_expand(const int64_t* rowA, const int64_t* colA, const void* valA,
                  const int64_t* rowptrB, const int64_t* colB, const void* valB,
                  const int64_t* offsets, const int64_t* owner, int64_t total,
                  int64_t n, int64_t* keys, void* vals, hipStream_t s) {
  const int64_t blocks = psa::ceil_div(total, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "too many products for one launch");
  hipLaunchKernelGGL(spspmm_expand_kernel<T>, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, rowA, colA, static_cast<const T*>(valA),
                     rowptrB, colB, static_cast<const T*>(valB), offsets, owner, total,
                     n, keys, static_cast<T*>(vals));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // namespace

extern "C" {

int psa_spspmm_count(const int64_t* colA, int64_t nnzA, const int64_t* rowptrB,
                     int64_t* counts, psa_stream_t stream) {
  PSA_REQUIRE(nnzA >= 0, "negative size");
  if (nnzA == 0) return PSA_OK;
  PSA_REQUIRE(colA && rowptrB && counts, "NULL pointer");
  const int64_t blocks = psa::ceil_div(nnzA, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "nnzA too large for one launch");
  hipLaunchKernelGGL(spspmm_count_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, psa::as_stream(stream), colA, nnzA, rowptrB, counts);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_spspmm_expand(int dtype, const int64_t* rowA, const int64_t* colA,
                      const void* valA, const int64_t* rowptrB, const int64_t* colB,
                      const void* valB, const int64_t* offsets, const int64_t* owner,
                      int64_t total, int64_t n, int64_t* keys, void* vals,
                      psa_stream_t stream) {
  PSA_REQUIRE(total >= 0, "negative size");
  if (total == 0) return PSA_OK;
  PSA_REQUIRE(rowA && colA && rowptrB && colB && offsets && owner && keys, "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  switch (dtype) {
    case PSA_F32:
      return launch_expand<float>(rowA, colA, valA, rowptrB, colB, valB, offsets, owner,
                                  total, n, keys, vals, s);
    case PSA_F64:
      return launch_expand<double>(rowA, colA, valA, rowptrB, colB, valB, offsets, owner,
                                   total, n, keys, vals, s);
    case PSA_I32:
      return launch_expand<int32_t>(rowA, colA, valA, rowptrB, colB, valB, offsets, owner,
                                    total, n, keys, vals, s);
    case PSA_I64:
      return launch_expand<int64_t>(rowA, colA, valA, rowptrB, colB, valB, offsets, owner,
                                    total, n, keys, vals, s);
    default:
      psa::set_error("psa_spspmm_expand: dtype must be f32, f64, i32 or i64");
      return PSA_ERR_INVALID_ARG;
  }
}

}  // extern "C"
