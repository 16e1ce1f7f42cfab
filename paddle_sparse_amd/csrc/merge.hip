// Stable merge of two SORTED key streams (A's entry first on equal keys) — the
// "concatenate, then sort" of paddle_sparse/add.py:30-47, mul.py:57-73 and
// tensor.py:415-451 (to_symmetric) when both halves of the concatenation are
// sorted already, which is what a SparseTensor's (row, col) order guarantees.
//
// Merge path: workgroup t produces the merged positions [2048 t, 2048 (t+1)).
// The split of that range between the two streams (how many of the first d
// merged keys come from A) is found on the cross diagonal d by a search that
// one whole wave runs 64 probes at a time (5 dependent round trips for 20 M
// keys instead of 24).  The two input pieces (2048 keys together) are loaded
// into LDS, every key is ranked against the other piece there (lower bound for
// A's keys, upper bound for B's: that is the stability rule), placed in LDS at
// its rank, and the tile leaves with contiguous, coalesced stores.  Each key is
// read once and written once: 16-24 B per key against 6 radix passes of 24 B.
//
// A first version ranked whole streams against each other (two launches that
// each wrote every other slot of the output): 0.62 ms for 2 x 20 M keys, its
// half-filled lines being written twice; this form exists because of that.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;

// UPPER: first position whose key is > x; else first position whose key is >= x
template <bool UPPER>
__device__ __forceinline__ int bound(const int64_t* arr, int n, int64_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int64_t v = arr[mid];
    if (UPPER ? v <= x : v < x) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Number of A's keys among the first d keys of the stable merge.  p(i) = "a[i]
// is among them" = a[i] <= b[d - i - 1] is true for small i, false for large;
// 64 probes per round cut the range 64-fold.  All 64 lanes of the calling wave
// pass the same arguments.
__device__ __forceinline__ int64_t wave_diagonal(const int64_t* __restrict__ a, int64_t na,
                                                 const int64_t* __restrict__ b, int64_t nb,
                                                 int64_t d) {
  const int lane = threadIdx.x & 63;
  int64_t lo = d > nb ? d - nb : 0;
  int64_t hi = d < na ? d : na;
  while (hi > lo) {
    const int64_t step = (hi - lo + 63) >> 6;
    const int64_t i = lo + (lane + 1) * step - 1;
    const bool pred = i < hi && a[i] <= b[d - i - 1];
    const int c = __popcll(__ballot(pred));  // p is monotone: the first c probes hold
    if (step == 1) return lo + c;
    lo += c * step;
    hi = lo + step - 1 < hi ? lo + step - 1 : hi;
  }
  return lo;
}

__global__ void __launch_bounds__(kThreads)
merge_path_kernel(const int64_t* __restrict__ a, int64_t na, const int64_t* __restrict__ b,
                  int64_t nb, const uint32_t* __restrict__ payload_a,
                  const uint32_t* __restrict__ payload_b, int64_t* __restrict__ merged,
                  int64_t* __restrict__ source, uint32_t* __restrict__ payload_out) {
  __shared__ int64_t split[2];
  __shared__ int64_t buf[2 * kTile];  // in: A's piece then B's piece; out: keys | sources
  const int64_t d0 = static_cast<int64_t>(blockIdx.x) * kTile;
  const int64_t d1 = d0 + kTile < na + nb ? d0 + kTile : na + nb;
  if (threadIdx.x < 128) {  // wave 0: where the tile starts, wave 1: where it ends
    const int w = threadIdx.x >> 6;
    const int64_t i = wave_diagonal(a, na, b, nb, w ? d1 : d0);
    if ((threadIdx.x & 63) == 0) split[w] = i;
  }
  __syncthreads();
  const int64_t i0 = split[0], j0 = d0 - i0;
  const int cnt = static_cast<int>(d1 - d0);
  // sorted inputs give 0 <= split[1] - i0 <= cnt with both pieces inside their
  // arrays; the clamp only keeps UNSORTED input (whose result is unspecified)
  // from reading out of bounds
  int64_t ca64 = split[1] - i0;
  const int64_t ca_min = d1 - i0 - nb > 0 ? d1 - i0 - nb : 0;
  const int64_t ca_max = na - i0 < cnt ? na - i0 : cnt;
  ca64 = ca64 < ca_min ? ca_min : (ca64 > ca_max ? ca_max : ca64);
  const int ca = static_cast<int>(ca64);
  const int cb = cnt - ca;
  int64_t x[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int e = threadIdx.x + k * kThreads;
    x[k] = e < ca ? a[i0 + e] : (e < cnt ? b[j0 + (e - ca)] : 0);
    if (e < cnt) buf[e] = x[k];
  }
  __syncthreads();
  int rank[kItems];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int e = threadIdx.x + k * kThreads;
    rank[k] = e < ca ? e + bound<false>(buf + ca, cb, x[k])
                     : (e < cnt ? (e - ca) + bound<true>(buf, ca, x[k]) : 0);
  }
  __syncthreads();
  int64_t* out_key = buf;
  int64_t* out_src = buf + kTile;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int e = threadIdx.x + k * kThreads;
    if (e < cnt) {
      out_key[rank[k]] = x[k];
      out_src[rank[k]] = e < ca ? i0 + e : na + j0 + (e - ca);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < cnt) {
      const int64_t src = out_src[q];
      merged[d0 + q] = out_key[q];
      if (source) source[d0 + q] = src;
      if (payload_out) {
        // (the range test is for unsorted input only: colliding ranks leave slots unwritten)
        const bool ok = static_cast<uint64_t>(src) < static_cast<uint64_t>(na + nb);
        payload_out[d0 + q] = !ok ? 0u : (src < na ? payload_a[src] : payload_b[src - na]);
      }
    }
  }
}

}  // namespace

extern "C" {

int psa_merge_sorted(const int64_t* a, int64_t na, const int64_t* b, int64_t nb,
                     const void* payload_a, const void* payload_b, int64_t* merged_out,
                     int64_t* source_out, void* payload_out, psa_stream_t stream) {
  PSA_REQUIRE(na >= 0 && nb >= 0, "negative size");
  if (na + nb == 0) return PSA_OK;
  PSA_REQUIRE((na == 0 || a) && (nb == 0 || b) && merged_out, "NULL pointer");
  PSA_REQUIRE(payload_out == nullptr || ((na == 0 || payload_a) && (nb == 0 || payload_b)),
              "payload_out needs both payload arrays");
  PSA_REQUIRE(psa::aligned(payload_a, 4) && psa::aligned(payload_b, 4) && psa::aligned(payload_out, 4),
              "payloads must be 4-byte aligned");
  const int64_t tiles = psa::ceil_div(na + nb, kTile);
  PSA_REQUIRE(tiles < (1ll << 31), "too many keys for one launch");
  hipLaunchKernelGGL(merge_path_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kThreads), 0,
                     psa::as_stream(stream), a, na, b, nb, static_cast<const uint32_t*>(payload_a),
                     static_cast<const uint32_t*>(payload_b), merged_out, source_out,
                     static_cast<uint32_t*>(payload_out));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
