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

// hi[i] = keys[i] / div, lo[i] = keys[i] % div: a sorted key stream already
// holds both indices, so no permutation gather is needed to recover them.
// Keys and divisors below 2^32 take the 32-bit division; the generic 64-bit
// one (a few dozen instructions, still far under the memory time) is the rest.
__global__ void __launch_bounds__(kThreads)
split_keys_kernel(const int64_t* __restrict__ keys, int64_t n, int64_t div,
                  int64_t* __restrict__ hi, int64_t* __restrict__ lo) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  int64_t q;
  if ((static_cast<uint64_t>(div) >> 32) == 0 && (static_cast<uint64_t>(k) >> 32) == 0) {
    q = static_cast<uint32_t>(k) / static_cast<uint32_t>(div);
  } else {
    q = k / div;
  }
  if (hi) hi[i] = q;
  if (lo) lo[i] = k - q * div;
}

// out[i, :] = src[perm[i], :], rows of `chunks` pieces of sizeof(T) bytes.
// shift >= 0: chunks == 1 << shift (the usual case: 2^k-byte rows) — the row of a piece is a shift, not a
// 64-bit division (~40 VALU instructions at 4 issue cycles each: the pack of 1.4 M 512-byte rows of the halo
// exchange ran at 1.4 TB/s with it).
template <typename T>
__global__ void __launch_bounds__(kThreads)
gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ perm,
                   int64_t n, int64_t chunks, int64_t stride, int64_t first, T* __restrict__ out, int shift) {
  // rows of `stride` pieces in src, of which the window [first, first + chunks) is taken
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= n * chunks) return;
  int64_t i, c;
  if (shift >= 0) {
    i = g >> shift;
    c = g & (chunks - 1);
  } else {
    i = g / chunks;
    c = g - i * chunks;
  }
  out[g] = src[perm[i] * stride + first + c];
}

__global__ void __launch_bounds__(kThreads)
invert_perm_kernel(const int64_t* __restrict__ perm, int64_t n,
                   int64_t* __restrict__ inv) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}

// out[index[i]] += 1 (colcount, storage.py:414-418 scatter_add of ones).
__global__ void __launch_bounds__(kThreads)
bincount_kernel(const int64_t* __restrict__ index, int64_t n, int64_t size,
                unsigned long long* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t b = index[i];
  if (b >= 0 && b < size) atomicAdd(out + b, 1ull);
}

// ---- counts -> pointer (exclusive scan with a trailing total) -------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;

__device__ __forceinline__ int64_t wave_incl_scan(int64_t x, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t o = __shfl_up(static_cast<long long>(x), off);
    if (lane >= off) x += o;
  }
  return x;
}

__global__ void __launch_bounds__(kThreads)
scan_block_sums_kernel(const int64_t* __restrict__ in, int64_t n,
                       int64_t* __restrict__ block_sums) {
  __shared__ int64_t wsum[kThreads / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  int64_t acc = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j * kThreads + threadIdx.x;
    acc += i < n ? in[i] : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(static_cast<long long>(acc), off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += wsum[w];
    block_sums[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(1024)
scan_sums_kernel(int64_t* __restrict__ block_sums, int64_t nb) {
  __shared__ int64_t wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t b = tid * per;
  const int64_t e = b + per < nb ? b + per : nb;
  int64_t sum = 0;
  for (int64_t i = b; i < e; ++i) sum += block_sums[i];
  const int64_t incl = wave_incl_scan(sum, lane);
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int64_t run = incl - sum;
  for (int w = 0; w < wave; ++w) run += wtot[w];
  for (int64_t i = b; i < e; ++i) {
    const int64_t c = block_sums[i];
    block_sums[i] = run;
    run += c;
  }
}

// ptr[0] = 0, ptr[i+1] = counts[0] + ... + counts[i]
__global__ void __launch_bounds__(kThreads)
scan_write_kernel(const int64_t* __restrict__ in, int64_t n,
                  const int64_t* __restrict__ block_offsets,
                  int64_t* __restrict__ ptr) {
  __shared__ int64_t wsum[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kScanTile + wave * (kScanItems * 64);
  int64_t incl[kScanItems];
  int64_t carry = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = chunk + j * 64 + lane;
    const int64_t x = i < n ? in[i] : 0;
    incl[j] = wave_incl_scan(x, lane) + carry;
    carry = __shfl(static_cast<long long>(incl[j]), 63);
  }
  if (lane == 0) wsum[wave] = carry;
  __syncthreads();
  int64_t base = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = chunk + j * 64 + lane;
    if (i < n) ptr[i + 1] = base + incl[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ptr[0] = 0;
}

template <typename T>
int launch_gather(const void* src, const int64_t* perm, int64_t n,
                  int64_t row_bytes, void* out, hipStream_t s, int64_t src_row_bytes = -1,
                  int64_t offset_bytes = 0) {
  const int64_t chunks = row_bytes / static_cast<int64_t>(sizeof(T));
  const int64_t stride = src_row_bytes < 0 ? chunks : src_row_bytes / static_cast<int64_t>(sizeof(T));
  const int64_t first = offset_bytes / static_cast<int64_t>(sizeof(T));
  const int64_t blocks = psa::ceil_div(n * chunks, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "too many elements for one launch");
  int shift = -1;
  if ((chunks & (chunks - 1)) == 0) {
    shift = 0;
    while ((int64_t{1} << shift) < chunks) ++shift;
  }
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, static_cast<const T*>(src), perm, n,
                     chunks, stride, first, static_cast<T*>(out), shift);
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

int psa_split_keys(const int64_t* keys, int64_t n, int64_t div, int64_t* hi,
                   int64_t* lo, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0 || (hi == nullptr && lo == nullptr)) return PSA_OK;
  PSA_REQUIRE(div > 0, "div must be positive");
  PSA_REQUIRE(keys != nullptr, "keys is NULL");
  const int64_t blocks = psa::ceil_div(n, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "n too large for one launch");
  hipLaunchKernelGGL(split_keys_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, psa::as_stream(stream), keys, n, div, hi, lo);
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

int psa_gather_rows_window(const void* src, int64_t src_row_bytes, int64_t offset_bytes,
                           int64_t width_bytes, const int64_t* perm, int64_t n, void* out,
                           psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && width_bytes >= 0 && offset_bytes >= 0, "negative size");
  PSA_REQUIRE(offset_bytes + width_bytes <= src_row_bytes, "window exceeds the source row");
  if (n == 0 || width_bytes == 0) return PSA_OK;
  PSA_REQUIRE(src && perm && out, "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  const int64_t all = src_row_bytes | offset_bytes | width_bytes;
  if (all % 16 == 0 && psa::aligned(src, 16) && psa::aligned(out, 16))
    return launch_gather<float4>(src, perm, n, width_bytes, out, s, src_row_bytes, offset_bytes);
  if (all % 4 == 0 && psa::aligned(src, 4) && psa::aligned(out, 4))
    return launch_gather<uint32_t>(src, perm, n, width_bytes, out, s, src_row_bytes, offset_bytes);
  return launch_gather<uint8_t>(src, perm, n, width_bytes, out, s, src_row_bytes, offset_bytes);
}

int psa_bincount(const int64_t* index, int64_t n, int64_t size, int64_t* out,
                 psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && size >= 0, "negative size");
  hipStream_t s = psa::as_stream(stream);
  if (size > 0) {
    PSA_REQUIRE(out != nullptr, "out is NULL");
    PSA_ZERO(out, sizeof(int64_t) * size, s);
  }
  if (n == 0 || size == 0) return PSA_OK;
  PSA_REQUIRE(index != nullptr, "index is NULL");
  const int64_t blocks = psa::ceil_div(n, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "n too large for one launch");
  hipLaunchKernelGGL(bincount_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, index, n, size,
                     reinterpret_cast<unsigned long long*>(out));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

size_t psa_count2ptr_workspace_bytes(int64_t n) {
  return static_cast<size_t>(psa::ceil_div(n > 0 ? n : 1, kScanTile)) * sizeof(int64_t) + 256;
}

int psa_count2ptr(const int64_t* counts, int64_t n, int64_t* ptr_out,
                  void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  PSA_REQUIRE(ptr_out != nullptr, "ptr_out is NULL");
  hipStream_t s = psa::as_stream(stream);
  if (n == 0) {
    PSA_ZERO(ptr_out, sizeof(int64_t), s);
    return PSA_OK;
  }
  PSA_REQUIRE(counts != nullptr, "counts is NULL");
  if (workspace == nullptr || workspace_bytes < psa_count2ptr_workspace_bytes(n)) {
    psa::set_error("psa_count2ptr: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  const int64_t nb = psa::ceil_div(n, kScanTile);
  PSA_REQUIRE(nb <= 0x7fffffff, "n too large for one launch");
  int64_t* sums = static_cast<int64_t*>(workspace);
  const dim3 grid(static_cast<unsigned>(nb)), block(kThreads);
  hipLaunchKernelGGL(scan_block_sums_kernel, grid, block, 0, s, counts, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, nb);
  hipLaunchKernelGGL(scan_write_kernel, grid, block, 0, s, counts, n, sums, ptr_out);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
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
