// coalesce(index, value, m, n, op) as two C-ABI calls with one host read between
// them (or after both) — paddle_sparse/coalesce.py:25-29 + storage.py:158-171 +
// storage.py:454-486 are one call for the reference's user.
//
//   psa_coalesce_count : keys, sort, run-length count.  Leaves the sorted keys,
//                        the permutation (or the values that rode the sort) and
//                        the run structure in the workspace, {count, flags} in a
//                        device status word pair.
//   psa_coalesce_write : the [2, count] index and the reduced values, sized by the
//                        caller from the count it read — or written into
//                        worst-case buffers before the host has read anything
//                        (the count is taken from the device).
//
// Inputs of at most kSmallTile entries run as ONE workgroup whose radix sort
// never leaves the LDS (keys and element ids of the whole input, 120 KB): every
// pass is rank -> scan -> scatter between barriers, no global round trip.  The
// first one-workgroup kernel (sort.hip, psa_coalesce_small) bounced the keys
// through global memory once per pass and ran a separate histogram sweep: 60 us
// for 10 k entries, as much as a CPU core needs for the whole coalesce.
#include "coalesce_internal.h"
#include "common.h"
#include "radix_util.h"
#include "reduce_util.h"

namespace psa {
}

namespace {

using psa::Acc;
using psa::digit_of;
using psa::kRadix;
using psa::mean_div;
using psa::wave_match;
using psa::wave_rank;

constexpr int kThreads = 256;
constexpr int kSmallThreads = 1024;
constexpr int kSmallItems = 10;
constexpr int kSmallTile = kSmallThreads * kSmallItems;  // 10240 entries: the one-workgroup limit
constexpr int kSmallWaves = kSmallThreads / 64;

enum { R_SUM = 0, R_MEAN = 1, R_MIN = 2, R_MAX = 3 };
enum { F_RANGE = 1, F_UNSORTED = 2, F_SORT_FAULT = 4 };  // status[1] bits

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int bits_for(int64_t max_value) {  // bits needed for keys in [0, max_value)
  int b = 0;
  uint64_t v = max_value > 1 ? static_cast<uint64_t>(max_value - 1) : 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

// ---- one workgroup, LDS-resident -------------------------------------------
// Item i of lane l of wave w is element w * (ITEMS * 64) + i * 64 + l (striped,
// so a wave's items are in input order item after item: the per-wave ranks are
// stable).
__global__ void __launch_bounds__(kSmallThreads)
small_coalesce_sort_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                           int64_t M, int64_t N, int n, int passes,
                           int64_t* __restrict__ out_row, int64_t* __restrict__ out_col,
                           int64_t* __restrict__ ptr, int64_t* __restrict__ perm,
                           int64_t* __restrict__ status,
                           // fused form (psa_coalesce_small_fused): the packed [2, count] index and the
                           // reduced 4-byte scalar values straight from this launch; NULL = two-call form
                           int64_t* __restrict__ index_out, const void* __restrict__ value, int vkind,
                           int red, void* __restrict__ value_out) {
  constexpr int ITEMS = kSmallItems;
  constexpr int WAVES = kSmallWaves;
  __shared__ uint64_t lkeys[kSmallTile];
  __shared__ uint32_t lidx[kSmallTile];
  __shared__ uint32_t wcnt[WAVES][kRadix];
  __shared__ uint32_t wtot[WAVES];
  __shared__ uint32_t lflags;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wbase = wave * (ITEMS * 64);
  if (tid == 0) lflags = 0;

  uint64_t key[ITEMS];
  uint32_t idx[ITEMS];
  bool bad = false;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = wbase + i * 64 + lane;
    idx[i] = static_cast<uint32_t>(q);
    key[i] = ~0ull;
    if (q < n) {
      const int64_t r = row[q], c = col[q];
      bad |= r < 0 || r >= M || c < 0 || c >= N;
      key[i] = static_cast<uint64_t>(r * N + c);
    }
  }
  // input order: is key[q] < key[q - 1] anywhere (storage.py:163)?
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = wbase + i * 64 + lane;
    if (q < n) {
      lkeys[q] = key[i];
      lidx[q] = static_cast<uint32_t>(q);  // stands when passes == 0 (a 1 x 1 matrix)
    }
  }
  __syncthreads();  // lflags zeroed, input keys in the LDS
  bool unsorted = false;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = wbase + i * 64 + lane;
    if (q > 0 && q < n) unsorted |= static_cast<int64_t>(key[i]) < static_cast<int64_t>(lkeys[q - 1]);
  }
  if (__any(bad) && lane == 0) atomicOr(&lflags, static_cast<uint32_t>(F_RANGE));
  if (__any(unsorted) && lane == 0) atomicOr(&lflags, static_cast<uint32_t>(F_UNSORTED));

  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    for (int i = tid; i < WAVES * kRadix; i += kSmallThreads) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const bool valid = wbase + i * 64 + lane < n;
      const unsigned d = digit_of(key[i], shift);
      rank[i] = wave_rank(wcnt[wave], d, valid, wave_match(d, valid));
    }
    __syncthreads();
    // (digit, wave) counts -> scatter bases: exclusive over the waves inside a
    // digit, then over the digits
    uint32_t total = 0;
    if (tid < kRadix) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) total += wcnt[w][tid];
      uint32_t incl = total;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wtot[wave] = incl;
      total = incl - total;  // exclusive inside this wave of digits
    }
    __syncthreads();
    if (tid < kRadix) {
      uint32_t run = total;
      for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = wcnt[w][tid];
        wcnt[w][tid] = run;
        run += c;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (wbase + i * 64 + lane < n) {
        const uint32_t dst = wcnt[wave][digit_of(key[i], shift)] + rank[i];
        lkeys[dst] = key[i];
        lidx[dst] = idx[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int q = wbase + i * 64 + lane;
      if (q < n) {
        key[i] = lkeys[q];
        idx[i] = lidx[q];
      }
    }
  }
  // (passes == 0, a 1 x 1 matrix: the keys written above, all 0, stand in input order)
  // ---- run-length structure --------------------------------------------------
  uint32_t pos[ITEMS];
  uint32_t wave_heads = 0;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = wbase + i * 64 + lane;
    const bool head = q < n && (q == 0 || key[i] != lkeys[q - 1]);
    const unsigned long long m = __ballot(head);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    pos[i] = head ? wave_heads + below : 0xffffffffu;
    wave_heads += static_cast<uint32_t>(__popcll(m));
  }
  __syncthreads();  // wtot is free again
  if (lane == 0) wtot[wave] = wave_heads;
  __syncthreads();
  uint32_t before = 0, count = 0;
  for (int w = 0; w < WAVES; ++w) {
    if (w < wave) before += wtot[w];
    count += wtot[w];
  }
  const bool narrow = (static_cast<uint64_t>(M) * static_cast<uint64_t>(N)) >> 32 == 0;  // 32-bit division
  if (index_out != nullptr) {
    // ---- fused form: run starts go to the LDS (the key array is free now), every head's
    // thread writes its (row, col) into the packed index and reduces its run of values
    __syncthreads();  // all head tests have read lkeys
    uint32_t* lheads = reinterpret_cast<uint32_t*>(lkeys);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (pos[i] != 0xffffffffu) lheads[before + pos[i]] = static_cast<uint32_t>(wbase + i * 64 + lane);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (pos[i] == 0xffffffffu) continue;
      const uint32_t o = before + pos[i];
      int64_t r;
      if (narrow) r = static_cast<uint32_t>(key[i]) / static_cast<uint32_t>(N);
      else r = static_cast<int64_t>(key[i]) / N;
      index_out[o] = r;
      index_out[count + o] = static_cast<int64_t>(key[i]) - r * N;
      if (vkind == 0) continue;
      const uint32_t b = lheads[o], e = o + 1 < count ? lheads[o + 1] : static_cast<uint32_t>(n);
      if (vkind == 1) {
        const float* v = static_cast<const float*>(value);
        float acc = v[lidx[b]];
        for (uint32_t j = b + 1; j < e; ++j) {
          const float x = v[lidx[j]];
          acc = red == R_MIN ? (x < acc ? x : acc) : red == R_MAX ? (x > acc ? x : acc) : acc + x;
        }
        if (red == R_MEAN) acc = acc / static_cast<float>(e - b);
        static_cast<float*>(value_out)[o] = acc;
      } else {
        const int32_t* v = static_cast<const int32_t*>(value);
        int32_t acc = v[lidx[b]];
        for (uint32_t j = b + 1; j < e; ++j) {
          const int32_t x = v[lidx[j]];
          acc = red == R_MIN ? (x < acc ? x : acc) : red == R_MAX ? (x > acc ? x : acc) : acc + x;
        }
        if (red == R_MEAN) acc = mean_div<int32_t>(acc, static_cast<int64_t>(e - b));
        static_cast<int32_t*>(value_out)[o] = acc;
      }
    }
    if (tid == 0) {
      status[0] = count;
      status[1] = lflags;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = wbase + i * 64 + lane;
    if (q < n) {
      perm[q] = static_cast<int64_t>(idx[i]);
      if (pos[i] != 0xffffffffu) {
        const int64_t o = before + pos[i];
        int64_t r;
        if (narrow) r = static_cast<uint32_t>(key[i]) / static_cast<uint32_t>(N);
        else r = static_cast<int64_t>(key[i]) / N;
        out_row[o] = r;
        out_col[o] = static_cast<int64_t>(key[i]) - r * N;
        ptr[o] = q;
      }
    }
  }
  if (tid == 0) {
    ptr[count] = n;
    status[0] = count;
    status[1] = lflags;
  }
}

// Phase 2 of the small path: the packed [2, count] index from the compact
// row' / col' arrays of phase 1, and the segmented reduce of value[perm[.]],
// one thread per output element (s, d).
template <typename T, int RED>
__global__ void __launch_bounds__(kThreads)
small_pack_reduce_kernel(const int64_t* __restrict__ row_c, const int64_t* __restrict__ col_c,
                         const int64_t* __restrict__ ptr, const int64_t* __restrict__ perm,
                         const int64_t* __restrict__ status, const T* __restrict__ value, int64_t D,
                         int64_t n_bound, int64_t* __restrict__ index_out, T* __restrict__ value_out) {
  using A = typename Acc<T>::type;
  const int64_t count = status[0];
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g >= n_bound * D) return;
  const int64_t s = D == 1 ? g : g / D;
  const int64_t d = D == 1 ? 0 : g - s * D;
  if (s >= count) return;
  if (d == 0) {
    index_out[s] = row_c[s];
    index_out[count + s] = col_c[s];
  }
  if (value == nullptr) return;
  const int64_t b = ptr[s], e = ptr[s + 1];
  A acc = Acc<T>::load(value + perm[b] * D + d);
  for (int64_t i = b + 1; i < e; ++i) {
    const A x = Acc<T>::load(value + perm[i] * D + d);
    if (RED == R_MIN) acc = x < acc ? x : acc;
    else if (RED == R_MAX) acc = x > acc ? x : acc;
    else acc = acc + x;
  }
  if (RED == R_MEAN) acc = mean_div<A>(acc, e - b);
  Acc<T>::store(value_out + g, acc);
}

template <typename T>
int launch_small_pack(int reduce, const int64_t* row_c, const int64_t* col_c, const int64_t* ptr,
                      const int64_t* perm, const int64_t* status, const void* value, int64_t D,
                      int64_t n_bound, int64_t* index_out, void* value_out, hipStream_t s) {
  const int64_t blocks = psa::ceil_div(n_bound * D, kThreads);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
  const T* v = static_cast<const T*>(value);
  T* o = static_cast<T*>(value_out);
#define PSA_P(R) hipLaunchKernelGGL((small_pack_reduce_kernel<T, R>), grid, block, 0, s, row_c, col_c, ptr, perm, status, v, D, n_bound, index_out, o)
  if (reduce == PSA_SUM) PSA_P(R_SUM);
  else if (reduce == PSA_MEAN) PSA_P(R_MEAN);
  else if (reduce == PSA_MIN) PSA_P(R_MIN);
  else PSA_P(R_MAX);
#undef PSA_P
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

// ---- large path: keys + range / order flags -----------------------------------
__global__ void __launch_bounds__(kThreads)
chain_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t n, int64_t M,
                  int64_t N, int64_t* __restrict__ keys, int64_t* __restrict__ status) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  unsigned f = 0;
  if (i < n) {
    const int64_t r = row[i], c = col[i];
    const int64_t k = r * N + c;
    keys[i] = k;
    if (r < 0 || r >= M || c < 0 || c >= N) f |= F_RANGE;
    if (i > 0 && k < row[i - 1] * N + col[i - 1]) f |= F_UNSORTED;
  }
  // Plain, idempotent stores into two words of their own: on a random input every
  // wave sees an inversion, and 1.5 M atomic ORs on one address (100 M entries)
  // queue up for ~11 ns each — 17 ms where the sort itself takes 4.
  if (__any(f & F_RANGE) && (threadIdx.x & 63) == 0) status[2] = 1;
  if (__any(f & F_UNSORTED) && (threadIdx.x & 63) == 0) status[3] = 1;
}

// status[1] = flag bits from the two words above (one thread, after the key pass)
__global__ void chain_flags_kernel(int64_t* __restrict__ status) {
  status[1] = (status[2] ? F_RANGE : 0) | (status[3] ? F_UNSORTED : 0);
}

// after the sort: a look-back spin that gave up (never expected) makes the order invalid — the
// caller learns it from the status words it reads anyway instead of getting a wrong permutation
__global__ void chain_sort_fault_kernel(const uint32_t* __restrict__ fault, int64_t* __restrict__ status) {
  if (*fault != 0) status[1] |= F_SORT_FAULT;
}

// ---- the count call of the large path in 7 launches (was 12) -----------------------------------
// Z  chain_zero_kernel       status words + the sort's look-back area, one launch
// K  chain_keys_hist_kernel  keys, range / order flags and the digit histograms of every radix
//                            pass from the one read of (row, col) (no separate histogram launch;
//                            every pass scans its 256 counts itself, sort.hip)
// P  the radix passes        psa::sort_prepared
// U  unique_count_chain      heads per 2048-key block, then the one-block scan, which also writes
//                            the count and folds the flag words and the sort's fault word
// (a one-launch U — "last block scans" behind a release fence, or a decoupled look-back over the
// blocks — measured slower at 1 M entries: 15-22 us against 5 + 5; profiles/r02_coalesce_kernels.txt)
constexpr int kMaxPasses = 8;
enum { S_COUNT = 0, S_FLAGS = 1, S_RANGE = 2, S_UNSORTED = 3 };

__global__ void __launch_bounds__(kThreads)
chain_zero_kernel(uint4* __restrict__ a, size_t na, uint4* __restrict__ b, size_t nb) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) a[i] = z;
    else b[i - na] = z;
  }
}

__global__ void __launch_bounds__(kThreads)
chain_keys_hist_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t n, int64_t M,
                       int64_t N, int64_t* __restrict__ keys, int64_t* __restrict__ status, int passes,
                       uint32_t* __restrict__ ghist) {
  __shared__ uint32_t hist[kMaxPasses][kRadix];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kMaxPasses * kRadix; i += kThreads) (&hist[0][0])[i] = 0;
  __syncthreads();
  unsigned f = 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + tid; i < n; i += stride) {
    const int64_t r = row[i], c = col[i];
    const int64_t k = r * N + c;
    keys[i] = k;
    if (r < 0 || r >= M || c < 0 || c >= N) f |= F_RANGE;
    if (i > 0 && k < row[i - 1] * N + col[i - 1]) f |= F_UNSORTED;
    for (int p = 0; p < passes; ++p) atomicAdd(&hist[p][digit_of(static_cast<uint64_t>(k), 8 * p)], 1u);
  }
  // plain, idempotent stores into two words of their own (see chain_keys_kernel)
  if (__any(f & F_RANGE) && lane == 0) status[S_RANGE] = 1;
  if (__any(f & F_UNSORTED) && lane == 0) status[S_UNSORTED] = 1;
  __syncthreads();
  for (int i = tid; i < passes * kRadix; i += kThreads) {
    const uint32_t c = (&hist[0][0])[i];
    if (c) atomicAdd(ghist + i, c);
  }
}

bool small_path(int64_t n, int64_t M, int64_t N) {
  return n <= kSmallTile && M > 0 && N > 0 && static_cast<double>(M) * static_cast<double>(N) < 9.0e18;
}

// Workspace layout.  Both paths: [status 256 B].  Small: row' | col' | ptr | perm.
// Large: keys | sorted keys | perm (8 n) or payload (4 n) | ptr | sort scratch | unique scratch.
struct ChainWs {
  int64_t* status;
  int64_t* a;  // small: row'   large: keys
  int64_t* b;  // small: col'   large: sorted keys
  int64_t* ptr;
  int64_t* perm;  // large: perm, or the sorted 4-byte payload
  void* sort_ws;
  size_t sort_bytes;
  void* uniq_ws;
  size_t uniq_bytes;
  size_t total;
};

ChainWs carve(void* base, int64_t n, int64_t M, int64_t N) {
  ChainWs w{};
  char* p = static_cast<char*>(base);
  size_t off = 256;
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  auto take = [&](size_t bytes) {
    char* q = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return q;
  };
  w.status = reinterpret_cast<int64_t*>(p);
  w.a = reinterpret_cast<int64_t*>(take(8 * nn));
  w.b = reinterpret_cast<int64_t*>(take(8 * nn));
  w.ptr = reinterpret_cast<int64_t*>(take(8 * (nn + 1)));
  w.perm = reinterpret_cast<int64_t*>(take(8 * nn));
  if (!small_path(n, M, N)) {
    const int64_t max_value = (M > 0 && N > 0 && static_cast<double>(M) * static_cast<double>(N) < 9.0e18) ? M * N : INT64_MAX;
    w.sort_bytes = psa_index_sort_workspace_bytes(n, max_value);
    w.sort_ws = take(w.sort_bytes);
    w.uniq_bytes = psa_unique_workspace_bytes(n);
    w.uniq_ws = take(w.uniq_bytes);
  }
  w.total = off;
  return w;
}

int dtype_bytes(int dtype) {
  switch (dtype) {
    case PSA_F32: case PSA_I32: return 4;
    case PSA_F64: case PSA_I64: return 8;
    case PSA_F16: case PSA_BF16: return 2;
    default: return 0;
  }
}

// 4-byte scalar values ride through the sort as its payload (the reduce then
// reads them as a stream instead of value[perm[i]])
bool rides(const void* value, int dtype, int64_t D) {
  return value != nullptr && D == 1 && dtype_bytes(dtype) == 4;
}

}  // namespace

extern "C" {

size_t psa_coalesce_workspace_bytes(int64_t n, int64_t M, int64_t N) {
  return carve(nullptr, n, M, N).total;
}

int psa_coalesce_count(const int64_t* row, const int64_t* col, const void* value, int dtype, int64_t D,
                       int64_t n, int64_t M, int64_t N, void* workspace, size_t workspace_bytes,
                       psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && M >= 0 && N >= 0 && D >= 0, "negative size");
  PSA_REQUIRE(n < (1ll << 31), "n >= 2^31 not supported by this build");
  PSA_REQUIRE(workspace != nullptr && psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
  const ChainWs w = carve(workspace, n, M, N);
  if (workspace_bytes < w.total) {
    psa::set_error("psa_coalesce_count: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  hipStream_t s = psa::as_stream(stream);
  if (n == 0) {
    PSA_ZERO(w.status, 16, s);
    return PSA_OK;
  }
  PSA_REQUIRE(row && col, "NULL pointer");
  PSA_REQUIRE(M > 0 && N > 0, "entries in an empty matrix");
  PSA_REQUIRE(value == nullptr || dtype_bytes(dtype) > 0, "unsupported dtype");
  if (small_path(n, M, N)) {
    const int passes = (bits_for(M * N) + 7) / 8;
    hipLaunchKernelGGL(small_coalesce_sort_kernel, dim3(1), dim3(kSmallThreads), 0, s, row, col, M, N,
                       static_cast<int>(n), passes, w.a, w.b, w.ptr, w.perm, w.status,
                       static_cast<int64_t*>(nullptr), static_cast<const void*>(nullptr), 0, 0,
                       static_cast<void*>(nullptr));
    PSA_LAUNCH_CHECK();
    return PSA_OK;
  }
  PSA_REQUIRE(static_cast<double>(M) * static_cast<double>(N) < 9.0e18, "M * N does not fit the 63-bit sort key");
  const psa::SortAreas areas = psa::sort_areas(w.sort_ws, n, M * N);
  if (areas.passes == 0 || !psa::sort_takes_prepared()) {
    // a 1 x 1 matrix (nothing to sort) or an A/B variant of the sort selected: the plain sequence
    PSA_ZERO(w.status, 64, s);
    const int64_t blocks = psa::ceil_div(n, kThreads);
    hipLaunchKernelGGL(chain_keys_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, row, col, n,
                       M, N, w.a, w.status);
    hipLaunchKernelGGL(chain_flags_kernel, dim3(1), dim3(1), 0, s, w.status);
    PSA_LAUNCH_CHECK();
    int st;
    if (rides(value, dtype, D))
      st = psa_sort_pairs_u32(w.a, value, n, M * N, w.b, w.perm, w.sort_ws, w.sort_bytes, stream);
    else
      st = psa_index_sort(w.a, n, M * N, w.b, w.perm, w.sort_ws, w.sort_bytes, stream);
    if (st != PSA_OK) return st;
    if (const uint32_t* fault = psa::sort_fault_word(w.sort_ws, n, M * N)) {
      hipLaunchKernelGGL(chain_sort_fault_kernel, dim3(1), dim3(1), 0, s, fault, w.status);
      PSA_LAUNCH_CHECK();
    }
    return psa_unique_count(w.b, n, w.uniq_ws, w.uniq_bytes, w.status, stream);
  }
  const size_t zero_b = areas.zero_bytes / 16;
  const size_t zero_a = 256 / 16;
  const size_t zero_items = zero_a + zero_b;
  const unsigned zero_blocks = static_cast<unsigned>(psa::ceil_div(static_cast<int64_t>(zero_items), kThreads) < 1024
                                                         ? psa::ceil_div(static_cast<int64_t>(zero_items), kThreads) : 1024);
  hipLaunchKernelGGL(chain_zero_kernel, dim3(zero_blocks), dim3(kThreads), 0, s, reinterpret_cast<uint4*>(w.status),
                     zero_a, static_cast<uint4*>(areas.zero_begin), zero_b);
  const int64_t key_blocks = psa::ceil_div(n, kThreads * 8) < 2048 ? psa::ceil_div(n, kThreads * 8) : 2048;
  hipLaunchKernelGGL(chain_keys_hist_kernel, dim3(static_cast<unsigned>(key_blocks)), dim3(kThreads), 0, s, row, col,
                     n, M, N, w.a, w.status, areas.passes, areas.ghist);
  PSA_LAUNCH_CHECK();
  const bool rode = rides(value, dtype, D);
  const int st = psa::sort_prepared(w.a, rode ? static_cast<const uint32_t*>(value) : nullptr, n, M * N, w.b,
                                    rode ? nullptr : w.perm, rode ? reinterpret_cast<uint32_t*>(w.perm) : nullptr,
                                    w.sort_ws, w.sort_bytes, s);
  if (st != PSA_OK) return st;
  return psa::unique_count_chain(w.b, n, w.uniq_ws, w.status, psa::sort_fault_word(w.sort_ws, n, M * N), s);
}

int psa_coalesce_small_max_fused(void) { return kSmallTile; }

int psa_coalesce_small_fused(const int64_t* row, const int64_t* col, const void* value, int dtype, int64_t n,
                             int64_t M, int64_t N, int reduce, int64_t* index_out, void* value_out,
                             int64_t* status, psa_stream_t stream) {
  PSA_REQUIRE(n > 0 && n <= kSmallTile, "n out of range for the one-launch form");
  PSA_REQUIRE(M > 0 && N > 0 && small_path(n, M, N), "matrix shape not served by the one-launch form");
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(row && col && index_out && status, "NULL pointer");
  PSA_REQUIRE(value == nullptr || ((dtype == PSA_F32 || dtype == PSA_I32) && value_out != nullptr),
              "the one-launch form takes fp32 / int32 scalar values (use psa_coalesce_count / _write)");
  const int passes = (bits_for(M * N) + 7) / 8;
  const int vkind = value == nullptr ? 0 : (dtype == PSA_F32 ? 1 : 2);
  hipLaunchKernelGGL(small_coalesce_sort_kernel, dim3(1), dim3(kSmallThreads), 0, psa::as_stream(stream), row, col,
                     M, N, static_cast<int>(n), passes, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr),
                     static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), status, index_out, value, vkind,
                     reduce, value_out);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_make_keys_checked(const int64_t* row, const int64_t* col, int64_t n, int64_t M, int64_t N,
                          int64_t* keys, int64_t* status, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && M >= 0 && N >= 0, "negative size");
  PSA_REQUIRE(status != nullptr, "status is NULL");
  hipStream_t s = psa::as_stream(stream);
  PSA_ZERO(status, 32, s);
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(row && col && keys, "NULL pointer");
  const int64_t blocks = psa::ceil_div(n, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "n too large for one launch");
  hipLaunchKernelGGL(chain_keys_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, row, col, n,
                     M, N, keys, status);
  hipLaunchKernelGGL(chain_flags_kernel, dim3(1), dim3(1), 0, s, status);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_coalesce_write(const void* value, int dtype, int64_t D, int64_t n, int64_t M, int64_t N,
                       int reduce, int64_t count, const void* workspace, int64_t* index_out,
                       void* value_out, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && D >= 0, "negative size");
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  if (n == 0 || count == 0) return PSA_OK;
  PSA_REQUIRE(workspace != nullptr && index_out != nullptr, "NULL pointer");
  PSA_REQUIRE(value == nullptr || value_out != nullptr, "value_out is NULL");
  const ChainWs w = carve(const_cast<void*>(workspace), n, M, N);
  hipStream_t s = psa::as_stream(stream);
  const int64_t bound = count > 0 ? count : n;  // count < 0: not read yet, outputs hold n rows
  if (small_path(n, M, N)) {
    const int64_t d = value ? D : 1;
    if (value != nullptr && D == 0) value = nullptr;
    switch (value ? dtype : PSA_I32) {
      case PSA_F32: return launch_small_pack<float>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      case PSA_F64: return launch_small_pack<double>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      case PSA_I32: return launch_small_pack<int32_t>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      case PSA_I64: return launch_small_pack<int64_t>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      case PSA_F16: return launch_small_pack<__half>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      case PSA_BF16: return launch_small_pack<__hip_bfloat16>(reduce, w.a, w.b, w.ptr, w.perm, w.status, value, d, bound, index_out, value_out, s);
      default:
        psa::set_error("psa_coalesce_write: unsupported dtype");
        return PSA_ERR_UNSUPPORTED;
    }
  }
  const bool rode = rides(value, dtype, D);
  if (rode && (dtype == PSA_F32 || dtype == PSA_I32))  // index and reduced values in one launch
    return psa::unique_write_reduce_packed(reduce, dtype, w.b, n, N, w.uniq_ws, w.status, index_out, w.perm, value_out, s);
  int st = psa::unique_write_packed(w.b, n, N, w.uniq_ws, w.status, w.ptr, index_out, s);
  if (st != PSA_OK || value == nullptr || D == 0) return st;
  return psa::segment_reduce_dev(reduce, dtype, rode ? static_cast<const void*>(w.perm) : value,
                                 rode ? nullptr : w.perm, w.ptr, bound, w.status, D, 0, value_out, s);
}

}  // extern "C"
