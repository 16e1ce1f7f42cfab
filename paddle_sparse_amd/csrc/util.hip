// Streaming helpers around the sort: key construction, permutation gather,
// permutation inverse.  All are single-pass, coalesced on the index side.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// keys[i] = a[i] * mul + b[i]; *unsorted |= keys[i] < keys[i-1]
// (paddle_sparse/storage.py:159-163 builds the same key and the same test
// with three elementwise passes and a host sync).
__global__ void __launch_bounds__(kThreads)
make_keys_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                 int64_t mul, int64_t n, int64_t* __restrict__ keys,
                 int32_t* __restrict__ unsorted) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const int64_t k = a[i] * mul + b[i];
    keys[i] = k;
    if (i > 0) bad = k < a[i - 1] * mul + b[i - 1];
  }
  if (unsorted && __any(bad)) {
    if ((threadIdx.x & 63) == 0) *unsorted = 1;
  }
}

// out[i, :] = src[perm[i], :], rows of `chunks` pieces of sizeof(T) bytes.
template <typename T>
__global__ void __launch_bounds__(kThreads)
gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ perm,
                   int64_t n, int64_t chunks, T* __restrict__ out) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= n * chunks) return;
  const int64_t i = chunks == 1 ? g : g / chunks;
  const int64_t c = chunks == 1 ? 0 : g - i * chunks;
  out[g] = src[perm[i] * chunks + c];
}

__global__ void __launch_bounds__(kThreads)
invert_perm_kernel(const int64_t* __restrict__ perm, int64_t n,
                   int64_t* __restrict__ inv) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}

template <typename T>
int launch_gather(const void* src, const int64_t* perm, int64_t n,
                  int64_t row_bytes, void* out, hipStream_t s) {
  const int64_t chunks = row_bytes / static_cast<int64_t>(sizeof(T));
  const int64_t blocks = psa::ceil_div(n * chunks, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "too many elements for one launch");
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, static_cast<const T*>(src), perm, n,
                     chunks, static_cast<T*>(out));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // namespace

extern "C" {

int psa_make_keys(const int64_t* a, const int64_t* b, int64_t mul, int64_t n,
                  int64_t* keys, int32_t* unsorted_flag, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(a && b && keys, "NULL pointer");
  const int64_t blocks = psa::ceil_div(n, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "n too large for one launch");
  hipLaunchKernelGGL(make_keys_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, psa::as_stream(stream), a, b, mul, n,
                     keys, unsorted_flag);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_gather_rows(const void* src, const int64_t* perm, int64_t n,
                    int64_t row_bytes, void* out, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && row_bytes >= 0, "negative size");
  if (n == 0 || row_bytes == 0) return PSA_OK;
  PSA_REQUIRE(src && perm && out, "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  if (row_bytes % 16 == 0 && psa::aligned(src, 16) && psa::aligned(out, 16))
    return launch_gather<float4>(src, perm, n, row_bytes, out, s);
  if (row_bytes % 8 == 0 && psa::aligned(src, 8) && psa::aligned(out, 8))
    return launch_gather<uint64_t>(src, perm, n, row_bytes, out, s);
  if (row_bytes % 4 == 0 && psa::aligned(src, 4) && psa::aligned(out, 4))
    return launch_gather<uint32_t>(src, perm, n, row_bytes, out, s);
  if (row_bytes % 2 == 0 && psa::aligned(src, 2) && psa::aligned(out, 2))
    return launch_gather<uint16_t>(src, perm, n, row_bytes, out, s);
  return launch_gather<uint8_t>(src, perm, n, row_bytes, out, s);
}

int psa_invert_permutation(const int64_t* perm, int64_t n, int64_t* inv,
                           psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(perm && inv, "NULL pointer");
  const int64_t blocks = psa::ceil_div(n, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "n too large for one launch");
  hipLaunchKernelGGL(invert_perm_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, psa::as_stream(stream), perm, n, inv);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
