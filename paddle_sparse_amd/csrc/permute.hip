// dst[i] = src[perm[i]] for 4-byte elements through a FIXED permutation, in two passes that
// touch HBM only in streams and runs.
//
// Where it is used: the ways of an nnz-sized float array between CSR and CSC order —
// value[csr2csc] (tensor.py:254-257, transpose.py:19-22) and grad_value's way back from
// the one-pass backward over the CSC view.  Done as `out[i] = src[perm[i]]` these are 20 M
// dependent 4-byte reads on config 3, each its own fabric request and DRAM row activation
// (profiles/r03_pmc_backward.json: 20.3 M requests for 20 M elements, 0.38 ms ~ 53 G/s,
// whatever is kept in flight).  The permutation is structure (cached with csr2csc), so the
// route of every element can be planned once:
//
//   pass 1  a tile of 32 768 consecutive SOURCE elements is loaded into LDS; thread i of the
//           tile takes the element whose destination is i-th smallest among the tile's
//           (plan: sl[p], its index inside the tile) and stores it at its slot of the
//           intermediate array `mid` (plan: gs[p]).  `mid` is ordered by destination BLOCK
//           (32 768 destinations), inside a block by source: consecutive threads write
//           consecutive slots — runs of ~n_tile / n_blocks elements.
//   pass 2  block b's elements are exactly mid[b * 32768 .. (b + 1) * 32768) (a permutation
//           sends as many elements into a block as it has slots); they are dropped into LDS
//           at their destination's low bits (plan: lo[m]) and written out as one stream.
//
// Bytes per element: 4 + 2 + 4 read and 4 written, then 4 + 2 read and 4 written = 24, all
// sequential or in runs, against one 64-byte request for 4 useful bytes.  Plan: 8 bytes per
// element (uint16 + int32 + uint16), built from existing ops (two stable index_sorts, one
// inverse) plus perm_plan_pack_kernel below.
#include "common.h"

namespace {

constexpr int kLog = 15;
constexpr int kTile = 1 << kLog;  // elements per source tile and per destination block (128 KB of LDS)
constexpr int kThreads = 1024;
constexpr int kItems = kTile / kThreads;

// perm_ts: sources ordered by (tile of the source, block of its destination, source);
// gslot[s]: slot of source s in `mid`; perm_mid[m]: source at slot m; dest[s]: destination of s
__global__ void __launch_bounds__(256)
perm_plan_pack_kernel(const int64_t* __restrict__ perm_ts, const int64_t* __restrict__ gslot,
                      const int64_t* __restrict__ perm_mid, const int64_t* __restrict__ dest, int64_t n,
                      uint16_t* __restrict__ sl, int32_t* __restrict__ gs, uint16_t* __restrict__ lo) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= n) return;
  const int64_t s = perm_ts[p];
  sl[p] = static_cast<uint16_t>(s & (kTile - 1));
  gs[p] = static_cast<int32_t>(gslot[s]);
  lo[p] = static_cast<uint16_t>(dest[perm_mid[p]] & (kTile - 1));
}

__global__ void __launch_bounds__(kThreads)
perm_pass1_kernel(const uint32_t* __restrict__ src, const uint16_t* __restrict__ sl,
                  const int32_t* __restrict__ gs, int64_t n, uint32_t* __restrict__ mid) {
  __shared__ uint32_t buf[kTile];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const int tid = threadIdx.x;
  uint32_t l[kItems];
  int32_t g[kItems];
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int64_t p = base + j * kThreads + tid;
    const bool ok = p < n;
    buf[j * kThreads + tid] = ok ? __builtin_nontemporal_load(src + p) : 0u;
    l[j] = ok ? sl[p] : 0u;
    g[j] = ok ? gs[p] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kItems; ++j)
    if (g[j] >= 0) mid[g[j]] = buf[l[j]];
}

__global__ void __launch_bounds__(kThreads)
perm_pass2_kernel(const uint32_t* __restrict__ mid, const uint16_t* __restrict__ lo, int64_t n,
                  uint32_t* __restrict__ dst) {
  __shared__ uint32_t buf[kTile];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile;
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int64_t m = base + j * kThreads + tid;
    if (m < n) buf[lo[m]] = mid[m];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int64_t i = base + j * kThreads + tid;
    if (i < n) __builtin_nontemporal_store(buf[j * kThreads + tid], dst + i);
  }
}

}  // namespace

extern "C" {

int64_t psa_permute_tile(void) { return kTile; }

int psa_permute_plan_pack(const int64_t* perm_ts, const int64_t* gslot, const int64_t* perm_mid,
                          const int64_t* dest, int64_t n, void* sl, void* gs, void* lo, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && n < (1ll << 31), "n must be below 2^31");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(perm_ts && gslot && perm_mid && dest && sl && gs && lo, "NULL pointer");
  hipLaunchKernelGGL(perm_plan_pack_kernel, dim3(static_cast<unsigned>(psa::ceil_div(n, 256))), dim3(256), 0,
                     psa::as_stream(stream), perm_ts, gslot, perm_mid, dest, n, static_cast<uint16_t*>(sl),
                     static_cast<int32_t*>(gs), static_cast<uint16_t*>(lo));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_permute_apply_u32(const void* src, const void* sl, const void* gs, const void* lo, int64_t n, void* mid,
                          void* dst, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0 && n < (1ll << 31), "n must be below 2^31");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(src && sl && gs && lo && mid && dst, "NULL pointer");
  PSA_REQUIRE(src != dst && mid != dst && mid != src, "src, mid and dst must be three different arrays");
  PSA_REQUIRE(psa::aligned(src, 4) && psa::aligned(mid, 4) && psa::aligned(dst, 4), "4-byte alignment");
  hipStream_t s = psa::as_stream(stream);
  const dim3 grid(static_cast<unsigned>(psa::ceil_div(n, kTile)));
  hipLaunchKernelGGL(perm_pass1_kernel, grid, dim3(kThreads), 0, s, static_cast<const uint32_t*>(src),
                     static_cast<const uint16_t*>(sl), static_cast<const int32_t*>(gs), n, static_cast<uint32_t*>(mid));
  hipLaunchKernelGGL(perm_pass2_kernel, grid, dim3(kThreads), 0, s, static_cast<const uint32_t*>(mid),
                     static_cast<const uint16_t*>(lo), n, static_cast<uint32_t*>(dst));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
