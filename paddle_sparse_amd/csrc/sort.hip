// index_sort: stable LSD radix sort of int64 keys -> (sorted keys, permutation).
//
// Replaces the reference's index_sort seam (paddle_sparse/utils.py:14-23, a
// paddle argsort).  `max_value` (already part of that signature) bounds the
// key width, so only ceil(log2(max_value)/8) 8-bit passes run: 6 for the
// 2^24 x 2^24 matrices of BASELINE config 5 instead of 8.
//
// Per pass (classic histogram / scan / scatter with a fixed grid, no
// inter-workgroup hand-off inside a launch):
//   1. radix_hist_kernel    every block counts the digits of its contiguous
//                           key range (LDS histogram per wave)   -> counts
//   2. radix_scan_kernel    one wave per digit: exclusive scan of counts in
//                           (digit, block) order                 -> offsets
//   3. radix_scatter_kernel every block walks its range tile by tile; ranks
//                           keys inside the tile with wave-wide digit
//                           matching (stable), reorders the tile in LDS so
//                           that global stores are contiguous runs per digit,
//                           and carries its 256 running offsets in LDS.
// HBM traffic per pass: 8n (hist) + 12n read + 12n written (key64 + idx32).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 8;                    // keys per thread per tile
constexpr int kTile = kThreads * kItems;     // 2048 keys per tile
constexpr int kRadix = 256;
constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ unsigned digit_of(uint64_t key, int shift) {
  return static_cast<unsigned>(key >> shift) & (kRadix - 1);
}

// ---- 1. histogram ---------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                  int tiles_per_block, int num_blocks,
                  uint32_t* __restrict__ counts,
                  uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t hist[kWaves][kRadix];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  for (int i = tid; i < kWaves * kRadix; i += kThreads)
    (&hist[0][0])[i] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTile;
  int64_t end = begin + static_cast<int64_t>(tiles_per_block) * kTile;
  end = end < n ? end : n;
  for (int64_t i = begin + tid; i < end; i += kThreads) {
    const unsigned d = digit_of(keys[i], shift);
    // Degenerate digits (all keys of the wave equal, e.g. high bytes of small
    // keys) would serialise 64 LDS atomics on one address.
    const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
    const unsigned long long same = __ballot(d == d0);
    const unsigned long long active = __ballot(1);
    if (same == active) {
      if ((tid & 63) == (__ffsll(static_cast<long long>(active)) - 1))
        atomicAdd(&hist[wave][d0], static_cast<uint32_t>(__popcll(active)));
    } else {
      atomicAdd(&hist[wave][d], 1u);
    }
  }
  __syncthreads();
  for (int d = tid; d < kRadix; d += kThreads) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) c += hist[w][d];
    counts[static_cast<size_t>(d) * num_blocks + blockIdx.x] = c;
    if (c) atomicAdd(&digit_total[d], c);
  }
}

// ---- 2. scan of counts in (digit, block) order -----------------------------
// One wave per digit.  base(d) = sum of digit_total[d' < d]; then an exclusive
// scan along the digit's row of num_blocks counters.
__global__ void __launch_bounds__(kThreads)
radix_scan_kernel(uint32_t* __restrict__ counts,
                  const uint32_t* __restrict__ digit_total, int num_blocks) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * kWaves + (threadIdx.x >> 6);
  uint32_t part = 0;
#pragma unroll
  for (int j = 0; j < kRadix / 64; ++j) {
    const int dd = j * 64 + lane;
    part += dd < d ? digit_total[dd] : 0u;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
  uint32_t carry = part;
  uint32_t* row = counts + static_cast<size_t>(d) * num_blocks;
  for (int b0 = 0; b0 < num_blocks; b0 += 64) {
    const int b = b0 + lane;
    const uint32_t c = b < num_blocks ? row[b] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (b < num_blocks) row[b] = carry + incl - c;
    carry += __shfl(incl, 63);
  }
}

// ---- 3. scatter ------------------------------------------------------------
// PASS0: payload is the element's own index (nothing to read).
// LAST : payload is written as int64 to perm_out (the API dtype).
template <bool PASS0, bool LAST>
__global__ void __launch_bounds__(kThreads)
radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                     const uint32_t* __restrict__ idx_in,
                     uint64_t* __restrict__ keys_out,  // may be null if LAST
                     uint32_t* __restrict__ idx_out,   // !LAST
                     int64_t* __restrict__ perm_out,   // LAST
                     int64_t n, int shift, int tiles_per_block, int num_blocks,
                     const uint32_t* __restrict__ offsets) {
  __shared__ uint64_t skey[kTile];
  __shared__ uint32_t sidx[kTile];
  __shared__ uint32_t wcnt[kWaves][kRadix];  // per-wave digit counts of a tile
  __shared__ uint32_t gbase[kRadix];         // running global offset per digit
  __shared__ int32_t gofs[kRadix];           // gbase - tile_digit_start
  __shared__ uint32_t wave_tot[kWaves];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  gbase[tid] = offsets[static_cast<size_t>(tid) * num_blocks + blockIdx.x];

  const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTile;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int64_t tile_begin = blk_begin + static_cast<int64_t>(t) * kTile;
    if (tile_begin >= n) break;  // block-uniform
    const int tile_n = (n - tile_begin) < kTile ? static_cast<int>(n - tile_begin) : kTile;

#pragma unroll
    for (int w = 0; w < kWaves; ++w) wcnt[w][tid] = 0;
    __syncthreads();

    // wave-striped inside a contiguous chunk per wave: item i of lane l is
    // tile element  wave*kItems*64 + i*64 + l  (index order == (wave,i,lane))
    uint64_t key[kItems];
    uint32_t idx[kItems];
    uint32_t rank[kItems];
    const int wbase = wave * (kItems * 64);
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      const int p = wbase + i * 64 + lane;
      const bool valid = p < tile_n;
      key[i] = valid ? keys_in[tile_begin + p] : ~0ull;
      if (PASS0) {
        idx[i] = static_cast<uint32_t>(tile_begin + p);
      } else {
        idx[i] = valid ? idx_in[tile_begin + p] : 0u;
      }
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      const int p = wbase + i * 64 + lane;
      const bool valid = p < tile_n;
      const unsigned d = digit_of(key[i], shift);
      // lanes of this wave holding the same digit
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
      }
      const uint32_t old = wcnt[wave][d];
      rank[i] = old + static_cast<uint32_t>(__popcll(peers & lt_mask));
      if (valid && (peers & lt_mask) == 0ull)  // lowest lane of the peer set
        wcnt[wave][d] = old + static_cast<uint32_t>(__popcll(peers));
    }
    __syncthreads();

    // thread d: offsets of digit d across waves, tile totals, digit scan
    const unsigned d = tid;
    uint32_t c[kWaves];
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      c[w] = wcnt[w][d];
      wcnt[w][d] = tot;  // exclusive offset of wave w inside digit d's run
      tot += c[w];
    }
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wprefix = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wprefix += w < wave ? wave_tot[w] : 0u;
    const uint32_t dstart = wprefix + incl - tot;  // tile position of digit d's run
    const uint32_t gb = gbase[d];
    gofs[d] = static_cast<int32_t>(gb) - static_cast<int32_t>(dstart);
    gbase[d] = gb + tot;
    // reuse wcnt[0] slot? no: keep dstart in registers via a second table
    // (wave offsets already hold per-wave exclusive counts); add dstart now.
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wcnt[w][d] += dstart;
    __syncthreads();

#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      const int p = wbase + i * 64 + lane;
      if (p < tile_n) {
        const unsigned dd = digit_of(key[i], shift);
        const uint32_t pos = wcnt[wave][dd] + rank[i];
        skey[pos] = key[i];
        sidx[pos] = idx[i];
      }
    }
    __syncthreads();

    for (int p = tid; p < tile_n; p += kThreads) {
      const uint64_t k = skey[p];
      const unsigned dd = digit_of(k, shift);
      const int64_t dst = static_cast<int64_t>(gofs[dd]) + p;
      if (LAST) {
        if (keys_out) keys_out[dst] = k;
        perm_out[dst] = static_cast<int64_t>(sidx[p]);
      } else {
        keys_out[dst] = k;
        idx_out[dst] = sidx[p];
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kThreads)
iota_kernel(int64_t* __restrict__ out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) out[i] = i;
}

struct SortPlan {
  int passes;
  int64_t num_tiles;
  int tiles_per_block;
  int num_blocks;
  size_t keys_bytes, idx_bytes, counts_bytes, total_bytes;
};

int bits_for(int64_t max_value) {
  // keys lie in [0, max_value); max_value <= 1 -> every key is 0
  if (max_value <= 1) return 0;
  uint64_t m = static_cast<uint64_t>(max_value - 1);
  int bits = 0;
  while (m) {
    ++bits;
    m >>= 1;
  }
  return bits;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

SortPlan make_plan(int64_t n, int64_t max_value) {
  SortPlan p;
  p.passes = (bits_for(max_value) + 7) / 8;
  p.num_tiles = psa::ceil_div(n > 0 ? n : 1, kTile);
  p.tiles_per_block = static_cast<int>(psa::ceil_div(p.num_tiles, kMaxBlocks));
  p.num_blocks = static_cast<int>(psa::ceil_div(p.num_tiles, p.tiles_per_block));
  p.keys_bytes = align_up(sizeof(uint64_t) * static_cast<size_t>(n), 256);
  p.idx_bytes = align_up(sizeof(uint32_t) * static_cast<size_t>(n), 256);
  p.counts_bytes =
      align_up(sizeof(uint32_t) * kRadix * (static_cast<size_t>(p.num_blocks) + 8), 256);
  p.total_bytes = 2 * p.keys_bytes + 2 * p.idx_bytes + p.counts_bytes;
  return p;
}

}  // namespace

extern "C" {

size_t psa_index_sort_workspace_bytes(int64_t n, int64_t max_value) {
  if (n <= 0) return 0;
  return make_plan(n, max_value).total_bytes;
}

int psa_index_sort(const int64_t* keys, int64_t n, int64_t max_value,
                   int64_t* sorted_out, int64_t* perm_out, void* workspace,
                   size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(keys != nullptr && perm_out != nullptr, "keys/perm_out is NULL");
  PSA_REQUIRE(max_value >= 0, "max_value must be >= 0");
  if (n >= (1ll << 31)) {
    psa::set_error("psa_index_sort: n >= 2^31 not supported by this build");
    return PSA_ERR_UNSUPPORTED;
  }
  hipStream_t s = psa::as_stream(stream);
  const SortPlan p = make_plan(n, max_value);
  if (p.passes == 0) {  // all keys equal: the stable permutation is identity
    hipLaunchKernelGGL(iota_kernel, dim3(static_cast<unsigned>(psa::ceil_div(n, kThreads))),
                       dim3(kThreads), 0, s, perm_out, n);
    PSA_LAUNCH_CHECK();
    if (sorted_out)
      PSA_HIP(hipMemcpyAsync(sorted_out, keys, sizeof(int64_t) * n,
                             hipMemcpyDeviceToDevice, s));
    return PSA_OK;
  }
  if (workspace == nullptr || workspace_bytes < p.total_bytes) {
    psa::set_error("psa_index_sort: workspace too small (need " +
                   std::to_string(p.total_bytes) + " bytes)");
    return PSA_ERR_WORKSPACE;
  }
  PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  uint64_t* kbuf[2] = {reinterpret_cast<uint64_t*>(ws),
                       reinterpret_cast<uint64_t*>(ws + p.keys_bytes)};
  uint32_t* ibuf[2] = {reinterpret_cast<uint32_t*>(ws + 2 * p.keys_bytes),
                       reinterpret_cast<uint32_t*>(ws + 2 * p.keys_bytes + p.idx_bytes)};
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + 2 * p.keys_bytes + 2 * p.idx_bytes);
  uint32_t* digit_total = counts + static_cast<size_t>(kRadix) * p.num_blocks;

  const uint64_t* kin = reinterpret_cast<const uint64_t*>(keys);
  const uint32_t* iin = nullptr;
  const dim3 grid(static_cast<unsigned>(p.num_blocks)), block(kThreads);
  for (int pass = 0; pass < p.passes; ++pass) {
    const int shift = 8 * pass;
    const bool last = pass == p.passes - 1;
    PSA_HIP(hipMemsetAsync(digit_total, 0, sizeof(uint32_t) * kRadix, s));
    hipLaunchKernelGGL(radix_hist_kernel, grid, block, 0, s, kin, n, shift,
                       p.tiles_per_block, p.num_blocks, counts, digit_total);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(kRadix / kWaves), block, 0, s,
                       counts, digit_total, p.num_blocks);
    uint64_t* kout = last ? reinterpret_cast<uint64_t*>(sorted_out) : kbuf[pass & 1];
    uint32_t* iout = last ? nullptr : ibuf[pass & 1];
#define PSA_SCATTER(P0, L)                                                      \
  hipLaunchKernelGGL((radix_scatter_kernel<P0, L>), grid, block, 0, s, kin, iin, \
                     kout, iout, perm_out, n, shift, p.tiles_per_block,          \
                     p.num_blocks, counts)
    if (pass == 0 && last) PSA_SCATTER(true, true);
    else if (pass == 0) PSA_SCATTER(true, false);
    else if (last) PSA_SCATTER(false, true);
    else PSA_SCATTER(false, false);
#undef PSA_SCATTER
    PSA_LAUNCH_CHECK();
    kin = kout;
    iin = iout;
  }
  return PSA_OK;
}

}  // extern "C"
