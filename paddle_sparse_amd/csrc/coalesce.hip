// Coalesce building blocks on SORTED keys, and the segmented reducer.
//
// The reference coalesces with ~10 Paddle ops, 3 host syncs and a
// paddle_scatter.segment_csr call (paddle_sparse/storage.py:454-486):
//   mask = key[i] > key[i-1]; row[mask], col[mask]; ptr = nonzero(mask)++[nnz];
//   value = segment_csr(value, ptr, reduce)
// Here: one counting pass (psa_unique_count), ONE host read of the count to
// size the outputs, one writing pass that emits ptr/row/col together
// (psa_unique_write), and a segmented reduce that gathers value[perm[i]]
// itself (psa_segment_reduce) so the permuted value array is never
// materialised.  psa_segment_reduce with perm == NULL and ptr == rowptr is
// the reference's reduce(dim=1) (paddle_sparse/reduce.py:50-51).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "coalesce_internal.h"
#include "common.h"
#include "reduce_util.h"

namespace {

using psa::Acc;
using psa::mean_div;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;  // 2048 keys per block

// Head flags of one wave's 512-key chunk, 64 keys (one per lane) at a time.
// Returns, for row j, the ballot of "key differs from its predecessor".
struct HeadScan {
  const int64_t* keys;
  int64_t n;
  int64_t chunk;  // first index of this wave's chunk
  int lane;
  int64_t prev_last;  // key of the element before the current row's lane 0

  __device__ HeadScan(const int64_t* k, int64_t n_, int64_t chunk_, int lane_)
      : keys(k), n(n_), chunk(chunk_), lane(lane_), prev_last(0) {}

  // mask of heads in row j; `key` receives this lane's key (valid if i < n)
  __device__ unsigned long long row(int j, int64_t& key, int64_t& i) {
    i = chunk + static_cast<int64_t>(j) * 64 + lane;
    const bool valid = i < n;
    key = valid ? keys[i] : 0;
    int64_t prev = __shfl_up(static_cast<long long>(key), 1);
    if (lane == 0) {
      if (j == 0) {
        prev = (i > 0 && valid) ? keys[i - 1] : 0;
      } else {
        prev = prev_last;
      }
    }
    prev_last = __shfl(static_cast<long long>(key), 63);
    const bool head = valid && (i == 0 || key != prev);
    return __ballot(head);
  }
};

__global__ void __launch_bounds__(kThreads)
unique_count_kernel(const int64_t* __restrict__ keys, int64_t n,
                    uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[kWaves];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kTile + wave * (kItems * 64);
  HeadScan hs(keys, n, chunk, lane);
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t key, i;
    cnt += static_cast<uint32_t>(__popcll(hs.row(j, key, i)));
  }
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += wsum[w];
    block_counts[blockIdx.x] = t;
  }
}

// Single block: exclusive scan of block_counts[nb] in place; total -> *count.
__global__ void __launch_bounds__(1024)
scan_blocks_kernel(uint32_t* __restrict__ block_counts, int64_t nb,
                   int64_t* __restrict__ count, int chain_status, const uint32_t* __restrict__ fault) {
  __shared__ uint32_t wtot[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t b = tid * per;
  const int64_t e = b + per < nb ? b + per : nb;
  uint32_t sum = 0;
  for (int64_t i = b; i < e; ++i) sum += block_counts[i];
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  uint32_t run = base + incl - sum;
  for (int64_t i = b; i < e; ++i) {
    const uint32_t c = block_counts[i];
    block_counts[i] = run;
    run += c;
  }
  if (tid == 1023) {
    *count = static_cast<int64_t>(base) + incl;
    // the coalesce chain's status words {count, flags, range seen, inversion seen} (chain.hip):
    // flag bits folded here, with the sort's fault word, instead of in launches of their own
    if (chain_status == 1) count[1] = (count[2] ? 1 : 0) | (count[3] ? 2 : 0) | ((fault != nullptr && *fault != 0) ? 4 : 0);
    // psa_unique_count_after_sort: count[1] = did a look-back of the sort that produced these keys give up
    if (chain_status == 2) count[1] = (fault != nullptr && *fault != 0) ? 1 : 0;
  }
}

__global__ void __launch_bounds__(kThreads)
unique_write_kernel(const int64_t* __restrict__ keys, int64_t n, int64_t N,
                    const uint32_t* __restrict__ block_offsets,
                    const int64_t* __restrict__ count,
                    int64_t* __restrict__ ptr_out, int64_t* __restrict__ row_out,
                    int64_t* __restrict__ col_out, int packed) {
  __shared__ uint32_t wsum[kWaves];
  if (packed) col_out = row_out + *count;  // [2, count] index: the col row follows the row row
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kTile + wave * (kItems * 64);
  HeadScan hs(keys, n, chunk, lane);
  unsigned long long masks[kItems];
  int64_t key[kItems];
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t i;
    masks[j] = hs.row(j, key[j], i);
    cnt += static_cast<uint32_t>(__popcll(masks[j]));
  }
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  uint32_t base = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int64_t i = chunk + static_cast<int64_t>(j) * 64 + lane;
    if ((masks[j] >> lane) & 1ull) {
      const int64_t s = static_cast<int64_t>(base) + __popcll(masks[j] & lt_mask);
      if (ptr_out) ptr_out[s] = i;
      if (row_out) {
        const int64_t r = key[j] / N;
        row_out[s] = r;
        col_out[s] = key[j] - r * N;
      }
    }
    base += static_cast<uint32_t>(__popcll(masks[j]));
  }
  if (ptr_out && blockIdx.x == 0 && threadIdx.x == 0) ptr_out[*count] = n;
}

// ---- segmented reduce ------------------------------------------------------
enum { R_SUM = 0, R_MEAN = 1, R_MIN = 2, R_MAX = 3 };

// One thread per output element (s, d); sequential, in segment order — the
// same order as the reference's segment_csr, so fp sums are reproducible.
template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
segment_reduce_kernel(const T* __restrict__ src, const int64_t* __restrict__ perm,
                      const int64_t* __restrict__ ptr, int64_t nseg, int64_t D,
                      T* __restrict__ out, const int64_t* __restrict__ nseg_dev) {
  using A = typename Acc<T>::type;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (nseg_dev) nseg = *nseg_dev < nseg ? *nseg_dev : nseg;
  if (g >= nseg * D) return;
  const int64_t s = D == 1 ? g : g / D;
  const int64_t d = D == 1 ? 0 : g - s * D;
  const int64_t b = ptr[s], e = ptr[s + 1];
  A acc = A(0);
  if (e > b) {
    acc = Acc<T>::load(src + (perm ? perm[b] : b) * D + d);
    for (int64_t i = b + 1; i < e; ++i) {
      const A x = Acc<T>::load(src + (perm ? perm[i] : i) * D + d);
      if (RED == R_MIN) acc = x < acc ? x : acc;
      else if (RED == R_MAX) acc = x > acc ? x : acc;
      else acc = acc + x;
    }
    if (RED == R_MEAN) acc = mean_div<A>(acc, e - b);
  }
  Acc<T>::store(out + g, acc);
}

// unique_write_kernel (packed index, no ptr) and segment_reduce_kernel in one launch, for 4-byte values that rode
// the sort as its payload: the thread that writes a distinct (row, col) pair also reduces the pair's run of the sorted
// payload — sequentially, in run order, i.e. with segment_reduce_kernel's order and bits.  Runs average 1.05 entries on
// the inputs this is for; a heavily duplicated input serialises on its few long runs exactly as the thread-per-segment
// reducer does (the run count is not known to the host when the launch is enqueued, so that is the reducer it gets).
// One launch and one ptr array (8 bytes per entry written and read) less per coalesce.
template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
unique_write_reduce_kernel(const int64_t* __restrict__ keys, int64_t n, int64_t N,
                           const uint32_t* __restrict__ block_offsets, const int64_t* __restrict__ count,
                           int64_t* __restrict__ row_out, const T* __restrict__ payload, T* __restrict__ out) {
  using A = typename Acc<T>::type;
  __shared__ uint32_t wsum[kWaves];
  int64_t* col_out = row_out + *count;  // [2, count] index: the col row follows the row row
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kTile + wave * (kItems * 64);
  HeadScan hs(keys, n, chunk, lane);
  unsigned long long masks[kItems];
  int64_t key[kItems];
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t i;
    masks[j] = hs.row(j, key[j], i);
    cnt += static_cast<uint32_t>(__popcll(masks[j]));
  }
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  uint32_t base = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int64_t i = chunk + static_cast<int64_t>(j) * 64 + lane;
    if ((masks[j] >> lane) & 1ull) {
      const int64_t s = static_cast<int64_t>(base) + __popcll(masks[j] & lt_mask);
      const int64_t r = key[j] / N;
      row_out[s] = r;
      col_out[s] = key[j] - r * N;
      A acc = Acc<T>::load(payload + i);
      int64_t k = i + 1;
      while (k < n && keys[k] == key[j]) {
        const A x = Acc<T>::load(payload + k);
        if (RED == R_MIN) acc = x < acc ? x : acc;
        else if (RED == R_MAX) acc = x > acc ? x : acc;
        else acc = acc + x;
        ++k;
      }
      if (RED == R_MEAN) acc = mean_div<A>(acc, k - i);
      Acc<T>::store(out + s, acc);
    }
    base += static_cast<uint32_t>(__popcll(masks[j]));
  }
}

// One wave per segment, D == 1, for long segments (row reductions of skewed
// matrices): lanes stride the segment, shuffle tree at the end.
template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
segment_reduce_wave_kernel(const T* __restrict__ src,
                           const int64_t* __restrict__ perm,
                           const int64_t* __restrict__ ptr, int64_t nseg,
                           T* __restrict__ out, const int64_t* __restrict__ nseg_dev) {
  using A = typename Acc<T>::type;
  const int lane = threadIdx.x & 63;
  const int64_t s = static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (nseg_dev) nseg = *nseg_dev < nseg ? *nseg_dev : nseg;
  if (s >= nseg) return;
  const int64_t b = ptr[s], e = ptr[s + 1];
  A acc = A(0);
  bool has = false;
  for (int64_t i = b + lane; i < e; i += 64) {
    const A x = Acc<T>::load(src + (perm ? perm[i] : i));
    if (!has) {
      acc = x;
      has = true;
    } else if (RED == R_MIN) {
      acc = x < acc ? x : acc;
    } else if (RED == R_MAX) {
      acc = x > acc ? x : acc;
    } else {
      acc = acc + x;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const A o = __shfl_xor(acc, off);
    const bool ohas = __shfl_xor(static_cast<int>(has), off);
    if (ohas) {
      if (!has) acc = o;
      else if (RED == R_MIN) acc = o < acc ? o : acc;
      else if (RED == R_MAX) acc = o > acc ? o : acc;
      else acc = acc + o;
      has = true;
    }
  }
  if (lane == 0) {
    if (RED == R_MEAN && e > b) acc = mean_div<A>(acc, e - b);
    Acc<T>::store(out + s, acc);
  }
}

template <typename T>
int launch_segment(int reduce, const void* src, const int64_t* perm,
                   const int64_t* ptr, int64_t nseg, int64_t D, int64_t n_hint,
                   void* out, hipStream_t s, const int64_t* nseg_dev = nullptr) {
  const T* sp = static_cast<const T*>(src);
  T* op = static_cast<T*>(out);
  const bool wave = D == 1 && n_hint >= 32 * nseg;
  if (wave) {
    const int64_t blocks = psa::ceil_div(nseg, kWaves);
    PSA_REQUIRE(blocks <= 0x7fffffff, "too many segments for one launch");
    const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
#define PSA_W(R) hipLaunchKernelGGL((segment_reduce_wave_kernel<T, R>), grid, block, 0, s, sp, perm, ptr, nseg, op, nseg_dev)
    if (reduce == PSA_SUM) PSA_W(R_SUM);
    else if (reduce == PSA_MEAN) PSA_W(R_MEAN);
    else if (reduce == PSA_MIN) PSA_W(R_MIN);
    else PSA_W(R_MAX);
#undef PSA_W
  } else {
    const int64_t blocks = psa::ceil_div(nseg * D, kThreads);
    PSA_REQUIRE(blocks <= 0x7fffffff, "too many elements for one launch");
    const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
#define PSA_S(R) hipLaunchKernelGGL((segment_reduce_kernel<T, R>), grid, block, 0, s, sp, perm, ptr, nseg, D, op, nseg_dev)
    if (reduce == PSA_SUM) PSA_S(R_SUM);
    else if (reduce == PSA_MEAN) PSA_S(R_MEAN);
    else if (reduce == PSA_MIN) PSA_S(R_MIN);
    else PSA_S(R_MAX);
#undef PSA_S
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

size_t unique_ws_bytes(int64_t n) {
  const int64_t nb = psa::ceil_div(n > 0 ? n : 1, kTile);
  return static_cast<size_t>(nb) * sizeof(uint32_t) + 256;
}

int segment_dispatch(int reduce, int dtype, const void* src, const int64_t* perm, const int64_t* ptr,
                     int64_t nseg, int64_t D, int64_t n_hint, void* out, hipStream_t s,
                     const int64_t* nseg_dev) {
  switch (dtype) {
    case PSA_F32: return launch_segment<float>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    case PSA_F64: return launch_segment<double>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    case PSA_I32: return launch_segment<int32_t>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    case PSA_I64: return launch_segment<int64_t>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    case PSA_F16: return launch_segment<__half>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    case PSA_BF16: return launch_segment<__hip_bfloat16>(reduce, src, perm, ptr, nseg, D, n_hint, out, s, nseg_dev);
    default:
      psa::set_error("psa_segment_reduce: unsupported dtype");
      return PSA_ERR_UNSUPPORTED;
  }
}

}  // namespace

namespace psa {

int unique_write_packed(const int64_t* sorted_keys, int64_t n, int64_t N, const void* workspace,
                        const int64_t* count, int64_t* ptr_out, int64_t* index_out, hipStream_t s) {
  PSA_REQUIRE(n > 0 && N > 0, "empty input");
  PSA_REQUIRE(sorted_keys && workspace && count && index_out, "NULL pointer");
  const int64_t nb = ceil_div(n, kTile);
  hipLaunchKernelGGL(unique_write_kernel, dim3(static_cast<unsigned>(nb)), dim3(kThreads), 0, s, sorted_keys, n, N,
                     static_cast<const uint32_t*>(workspace), count, ptr_out, index_out,
                     static_cast<int64_t*>(nullptr), 1);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int unique_write_reduce_packed(int reduce, int dtype, const int64_t* sorted_keys, int64_t n, int64_t N,
                               const void* workspace, const int64_t* count, int64_t* index_out, const void* payload,
                               void* value_out, hipStream_t s) {
  PSA_REQUIRE(n > 0 && N > 0, "empty input");
  PSA_REQUIRE(sorted_keys && workspace && count && index_out && payload && value_out, "NULL pointer");
  PSA_REQUIRE(dtype == PSA_F32 || dtype == PSA_I32, "the fused write takes 4-byte values that rode the sort");
  const dim3 grid(static_cast<unsigned>(ceil_div(n, kTile))), block(kThreads);
  const uint32_t* bo = static_cast<const uint32_t*>(workspace);
#define PSA_UWR(T, R)                                                                                             \
  hipLaunchKernelGGL((unique_write_reduce_kernel<T, R>), grid, block, 0, s, sorted_keys, n, N, bo, count, index_out, \
                     static_cast<const T*>(payload), static_cast<T*>(value_out))
#define PSA_UWR_T(T)                           \
  do {                                         \
    if (reduce == PSA_SUM) PSA_UWR(T, R_SUM);  \
    else if (reduce == PSA_MEAN) PSA_UWR(T, R_MEAN); \
    else if (reduce == PSA_MIN) PSA_UWR(T, R_MIN);   \
    else PSA_UWR(T, R_MAX);                    \
  } while (0)
  if (dtype == PSA_F32) PSA_UWR_T(float);
  else PSA_UWR_T(int32_t);
#undef PSA_UWR_T
#undef PSA_UWR
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int unique_count_chain(const int64_t* sorted_keys, int64_t n, void* workspace, int64_t* status,
                       const uint32_t* fault, hipStream_t s) {
  PSA_REQUIRE(n > 0 && sorted_keys && workspace && status, "bad argument");
  const int64_t nb = ceil_div(n, kTile);
  PSA_REQUIRE(nb <= 0x7fffffff, "n too large for one launch");
  uint32_t* bc = static_cast<uint32_t*>(workspace);
  hipLaunchKernelGGL(unique_count_kernel, dim3(static_cast<unsigned>(nb)), dim3(kThreads), 0, s, sorted_keys, n, bc);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, bc, nb, status, 1, fault);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int segment_reduce_dev(int reduce, int dtype, const void* src, const int64_t* perm,
                       const int64_t* ptr, int64_t nseg_bound, const int64_t* nseg_dev,
                       int64_t D, int64_t n_hint, void* out, hipStream_t s) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  if (nseg_bound <= 0 || D <= 0) return PSA_OK;
  PSA_REQUIRE(ptr && out, "NULL pointer");
  return segment_dispatch(reduce, dtype, src, perm, ptr, nseg_bound, D, n_hint, out, s, nseg_dev);
}

}  // namespace psa

extern "C" {

size_t psa_unique_workspace_bytes(int64_t n) { return unique_ws_bytes(n); }

int psa_unique_count_after_sort(const int64_t* sorted_keys, int64_t n, void* workspace, size_t workspace_bytes,
                                const void* sort_workspace, int64_t sort_max_value, int64_t* count_out,
                                psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  PSA_REQUIRE(count_out != nullptr, "count_out is NULL");
  hipStream_t s = psa::as_stream(stream);
  if (n == 0) {
    PSA_ZERO(count_out, 2 * sizeof(int64_t), s);
    return PSA_OK;
  }
  PSA_REQUIRE(sorted_keys != nullptr, "sorted_keys is NULL");
  if (workspace == nullptr || workspace_bytes < unique_ws_bytes(n)) {
    psa::set_error("psa_unique_count_after_sort: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  const int64_t nb = psa::ceil_div(n, kTile);
  PSA_REQUIRE(nb <= 0x7fffffff, "n too large for one launch");
  uint32_t* bc = static_cast<uint32_t*>(workspace);
  hipLaunchKernelGGL(unique_count_kernel, dim3(static_cast<unsigned>(nb)), dim3(kThreads), 0, s, sorted_keys, n, bc);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, bc, nb, count_out, 2,
                     psa::sort_fault_word(sort_workspace, n, sort_max_value));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_unique_count(const int64_t* sorted_keys, int64_t n, void* workspace,
                     size_t workspace_bytes, int64_t* count_out,
                     psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  PSA_REQUIRE(count_out != nullptr, "count_out is NULL");
  hipStream_t s = psa::as_stream(stream);
  if (n == 0) {
    PSA_ZERO(count_out, sizeof(int64_t), s);
    return PSA_OK;
  }
  PSA_REQUIRE(sorted_keys != nullptr, "sorted_keys is NULL");
  if (workspace == nullptr || workspace_bytes < unique_ws_bytes(n)) {
    psa::set_error("psa_unique_count: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  const int64_t nb = psa::ceil_div(n, kTile);
  PSA_REQUIRE(nb <= 0x7fffffff, "n too large for one launch");
  uint32_t* bc = static_cast<uint32_t*>(workspace);
  hipLaunchKernelGGL(unique_count_kernel, dim3(static_cast<unsigned>(nb)),
                     dim3(kThreads), 0, s, sorted_keys, n, bc);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, bc, nb, count_out, 0,
                     static_cast<const uint32_t*>(nullptr));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_unique_write(const int64_t* sorted_keys, int64_t n, int64_t N,
                     const void* workspace, const int64_t* count,
                     int64_t* ptr_out, int64_t* row_out, int64_t* col_out,
                     psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(sorted_keys && workspace && count, "NULL pointer");
  PSA_REQUIRE((row_out == nullptr) == (col_out == nullptr), "row_out/col_out must come together");
  PSA_REQUIRE(row_out == nullptr || N > 0, "N must be positive");
  const int64_t nb = psa::ceil_div(n, kTile);
  hipLaunchKernelGGL(unique_write_kernel, dim3(static_cast<unsigned>(nb)),
                     dim3(kThreads), 0, psa::as_stream(stream), sorted_keys, n, N,
                     static_cast<const uint32_t*>(workspace), count, ptr_out,
                     row_out, col_out, 0);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_unique_write_reduce(int reduce, int dtype, const int64_t* sorted_keys, int64_t n, int64_t N,
                            const void* workspace, const int64_t* count, int64_t* index_out, const void* payload,
                            void* value_out, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  if (n == 0) return PSA_OK;
  if (dtype != PSA_F32 && dtype != PSA_I32) {
    psa::set_error("psa_unique_write_reduce: 4-byte values only (PSA_F32 / PSA_I32); use psa_unique_write + psa_segment_reduce");
    return PSA_ERR_UNSUPPORTED;
  }
  return psa::unique_write_reduce_packed(reduce, dtype, sorted_keys, n, N, workspace, count, index_out, payload, value_out,
                                         psa::as_stream(stream));
}

int psa_segment_reduce(int reduce, int dtype, const void* src,
                       const int64_t* perm, const int64_t* ptr, int64_t nseg,
                       int64_t D, int64_t n_hint, void* out,
                       psa_stream_t stream) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(nseg >= 0 && D >= 0, "negative size");
  if (nseg == 0 || D == 0) return PSA_OK;
  PSA_REQUIRE(ptr && out, "NULL pointer");
  return segment_dispatch(reduce, dtype, src, perm, ptr, nseg, D, n_hint, out, psa::as_stream(stream), nullptr);
}

}  // extern "C"
