// sample_adj on the GPU (the reference's op throws "No CUDA version supported",
// csrc/sample.cpp:13-18; its CPU text is csrc/cpu/sample_cpu.cpp:9-148).
//
// The CPU text is a sequential walk with a hash map that hands out new node
// ids in first-seen order.  Here the same outputs come from data-parallel
// steps over the SELECTION LIST (slot p = out_rowptr[i] + t holds the t-th
// pick of subset row i):
//
//   counts      how many picks each subset row makes             (sample_count)
//   e_raw[p]    the picked edge of slot p                        (sample_select_*)
//   newid[c]    = n for subset nodes (last duplicate wins, as the map does),
//               = -2 - (first slot that picked c) otherwise,
//               one atomicMax per slot                           (relabel_mark)
//   flags[p]    slot p is the first pick of a node outside the subset
//   n_id, keys  a scan of the flags IS the first-seen numbering; keys =
//               i * n_out + new column id, sorted (stable) by the caller
//
// Random picks use a counter-based generator: draw t of subset row i depends on
// (seed, i, t) only, so results do not depend on scheduling and the CPU oracle
// reproduces them bit for bit.  (The reference draws from Paddle's global
// generator, which no other implementation can reproduce.)
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int64_t kUnseen = INT64_MIN;

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// uniform integer in [0, n): high half of a 64 x 64 product (n > 0)
__device__ inline int64_t randint(uint64_t seed, int64_t i, int64_t t, int64_t n) {
  const uint64_t r = mix64(mix64(seed ^ mix64(static_cast<uint64_t>(i))) + static_cast<uint64_t>(t));
  return static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(n)));
}

__device__ inline int64_t picks_of(int64_t deg, int64_t k, int replace) {
  if (k < 0) return deg;                       // sample_cpu.cpp:45-63
  if (replace) return deg > 0 ? k : 0;         // :66-88
  return deg < k ? deg : k;                    // :90-121
}

__global__ void __launch_bounds__(kThreads)
sample_count_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ idx,
                    int64_t S, int64_t k, int replace, int64_t* __restrict__ counts) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= S) return;
  const int64_t n = idx[i];
  counts[i] = picks_of(rowptr[n + 1] - rowptr[n], k, replace);
}

// One thread per slot: every case except "without replacement, deg > k".
__global__ void __launch_bounds__(kThreads)
sample_select_slot_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ idx,
                          const int64_t* __restrict__ out_rowptr,
                          const int64_t* __restrict__ owner, int64_t E, int64_t k,
                          int replace, uint64_t seed, int64_t* __restrict__ e_raw) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (p >= E) return;
  const int64_t i = owner[p];
  const int64_t t = p - out_rowptr[i];
  const int64_t n = idx[i];
  const int64_t start = rowptr[n], deg = rowptr[n + 1] - start;
  if (k < 0 || (!replace && deg <= k)) {
    e_raw[p] = start + t;
  } else if (replace) {
    e_raw[p] = start + randint(seed, i, t, deg);
  }  // else: a Floyd row, written by sample_select_floyd_kernel
}

// Without replacement, deg > k (sample_cpu.cpp:100-106): for j = deg-k .. deg-1
// draw r in [0, j); take r unless already taken, then take j.  One thread per
// row; picks are kept in draw order in the row's own output slots, which double
// as the "already taken" set (k is small: the scan is k^2 / 2 compares).
__global__ void __launch_bounds__(kThreads)
sample_select_floyd_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ idx,
                           const int64_t* __restrict__ out_rowptr, int64_t S, int64_t k,
                           uint64_t seed, int64_t* __restrict__ e_raw) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= S) return;
  const int64_t n = idx[i];
  const int64_t start = rowptr[n], deg = rowptr[n + 1] - start;
  if (deg <= k) return;
  int64_t* mine = e_raw + out_rowptr[i];
  for (int64_t t = 0; t < k; ++t) {
    const int64_t j = deg - k + t;
    int64_t pick = start + (j > 0 ? randint(seed, i, t, j) : 0);
    bool taken = j == 0;
    for (int64_t u = 0; u < t && !taken; ++u) taken = mine[u] == pick;
    if (taken) pick = start + j;
    mine[t] = pick;
  }
}

__global__ void __launch_bounds__(kThreads)
fill_i64_kernel(int64_t* __restrict__ p, int64_t n, int64_t v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) p[i] = v;
}

// threads [0, S): subset nodes; threads [S, S + E): selection slots.
__global__ void __launch_bounds__(kThreads)
relabel_mark_kernel(const int64_t* __restrict__ idx, int64_t S,
                    const int64_t* __restrict__ col, const int64_t* __restrict__ e_raw,
                    int64_t E, long long* __restrict__ newid) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g < S) {
    __hip_atomic_fetch_max(newid + idx[g], static_cast<long long>(g), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  } else if (g < S + E) {
    const int64_t p = g - S;
    __hip_atomic_fetch_max(newid + col[e_raw[p]], static_cast<long long>(-2 - p),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(kThreads)
relabel_flags_kernel(const int64_t* __restrict__ col, const int64_t* __restrict__ e_raw,
                     int64_t E, const int64_t* __restrict__ newid,
                     int64_t* __restrict__ flags) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (p >= E) return;
  flags[p] = newid[col[e_raw[p]]] == -2 - p ? 1 : 0;
}

// threads [0, S): n_id[i] = idx[i]; threads [S, S + E): slot p.
__global__ void __launch_bounds__(kThreads)
relabel_finish_kernel(const int64_t* __restrict__ idx, int64_t S,
                      const int64_t* __restrict__ col, const int64_t* __restrict__ e_raw,
                      int64_t E, const int64_t* __restrict__ newid,
                      const int64_t* __restrict__ rank, const int64_t* __restrict__ owner,
                      int64_t n_out, int64_t* __restrict__ n_id, int64_t* __restrict__ keys) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (g < S) {
    n_id[g] = idx[g];
  } else if (g < S + E) {
    const int64_t p = g - S;
    const int64_t c = col[e_raw[p]];
    const int64_t v = newid[c];
    int64_t id = v;
    if (v < 0) {
      const int64_t first = -2 - v;  // slot of the first pick of c
      id = S + rank[first];
      if (first == p) n_id[id] = c;
    }
    keys[p] = owner[p] * n_out + id;
  }
}

int grid_for(int64_t n, unsigned* blocks) {
  const int64_t b = psa::ceil_div(n > 0 ? n : 1, kThreads);
  if (b > 0x7fffffff) return 0;
  *blocks = static_cast<unsigned>(b);
  return 1;
}

}  // namespace

extern "C" {

int psa_sample_count(const int64_t* rowptr, const int64_t* idx, int64_t S,
                     int64_t num_neighbors, int replace, int64_t* counts,
                     psa_stream_t stream) {
  PSA_REQUIRE(S >= 0, "negative size");
  if (S == 0) return PSA_OK;
  PSA_REQUIRE(rowptr && idx && counts, "NULL pointer");
  unsigned blocks;
  PSA_REQUIRE(grid_for(S, &blocks), "subset too large for one launch");
  hipLaunchKernelGGL(sample_count_kernel, dim3(blocks), dim3(kThreads), 0,
                     psa::as_stream(stream), rowptr, idx, S, num_neighbors, replace, counts);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_sample_select(const int64_t* rowptr, const int64_t* idx, int64_t S,
                      const int64_t* out_rowptr, const int64_t* owner, int64_t E,
                      int64_t num_neighbors, int replace, uint64_t seed,
                      int64_t* e_raw, psa_stream_t stream) {
  PSA_REQUIRE(S >= 0 && E >= 0, "negative size");
  if (S == 0 || E == 0) return PSA_OK;
  PSA_REQUIRE(rowptr && idx && out_rowptr && owner && e_raw, "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  unsigned blocks;
  PSA_REQUIRE(grid_for(E, &blocks), "too many picks for one launch");
  hipLaunchKernelGGL(sample_select_slot_kernel, dim3(blocks), dim3(kThreads), 0, s, rowptr,
                     idx, out_rowptr, owner, E, num_neighbors, replace, seed, e_raw);
  if (num_neighbors >= 0 && !replace) {
    PSA_REQUIRE(grid_for(S, &blocks), "subset too large for one launch");
    hipLaunchKernelGGL(sample_select_floyd_kernel, dim3(blocks), dim3(kThreads), 0, s, rowptr,
                       idx, out_rowptr, S, num_neighbors, seed, e_raw);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_relabel_mark(const int64_t* idx, int64_t S, const int64_t* col,
                     const int64_t* e_raw, int64_t E, int64_t num_nodes,
                     int64_t* newid, int64_t* flags, psa_stream_t stream) {
  PSA_REQUIRE(S >= 0 && E >= 0 && num_nodes >= 0, "negative size");
  hipStream_t s = psa::as_stream(stream);
  unsigned blocks;
  if (num_nodes > 0) {
    PSA_REQUIRE(newid != nullptr, "newid is NULL");
    PSA_REQUIRE(grid_for(num_nodes, &blocks), "num_nodes too large for one launch");
    hipLaunchKernelGGL(fill_i64_kernel, dim3(blocks), dim3(kThreads), 0, s, newid, num_nodes,
                       kUnseen);
  }
  if (S + E == 0) return PSA_OK;
  PSA_REQUIRE(idx || S == 0, "idx is NULL");
  PSA_REQUIRE((col && e_raw && flags) || E == 0, "NULL pointer");
  PSA_REQUIRE(grid_for(S + E, &blocks), "too many picks for one launch");
  hipLaunchKernelGGL(relabel_mark_kernel, dim3(blocks), dim3(kThreads), 0, s, idx, S, col,
                     e_raw, E, reinterpret_cast<long long*>(newid));
  if (E > 0) {
    PSA_REQUIRE(grid_for(E, &blocks), "too many picks for one launch");
    hipLaunchKernelGGL(relabel_flags_kernel, dim3(blocks), dim3(kThreads), 0, s, col, e_raw, E,
                       newid, flags);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_relabel_finish(const int64_t* idx, int64_t S, const int64_t* col,
                       const int64_t* e_raw, int64_t E, const int64_t* newid,
                       const int64_t* rank, const int64_t* owner, int64_t n_out,
                       int64_t* n_id, int64_t* keys, psa_stream_t stream) {
  PSA_REQUIRE(S >= 0 && E >= 0 && n_out >= S, "bad sizes");
  if (S + E == 0) return PSA_OK;
  PSA_REQUIRE(n_id != nullptr && (idx || S == 0), "NULL pointer");
  PSA_REQUIRE((col && e_raw && newid && rank && owner && keys) || E == 0, "NULL pointer");
  unsigned blocks;
  PSA_REQUIRE(grid_for(S + E, &blocks), "too many picks for one launch");
  hipLaunchKernelGGL(relabel_finish_kernel, dim3(blocks), dim3(kThreads), 0,
                     psa::as_stream(stream), idx, S, col, e_raw, E, newid, rank, owner, n_out,
                     n_id, keys);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
