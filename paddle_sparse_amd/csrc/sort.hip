// index_sort: stable LSD radix sort of int64 keys -> (sorted keys, permutation).
//
// Replaces the reference's index_sort seam (paddle_sparse/utils.py:14-23, a
// paddle argsort).  `max_value` (already part of that signature) bounds the
// key width, so only ceil(log2(max_value)/8) 8-bit passes run: 6 for the
// 2^24 x 2^24 matrices of BASELINE config 5 instead of 8.
//
// PRODUCTION PATH (variant 0): single-sweep passes, see the block comment above
// os_hist_kernel — one key read builds every pass's histogram, each pass is one
// launch with decoupled look-back.  100 M 48-bit keys: 4.1 ms (rocPRIM
// onesweep through torch.sort: 8.2 ms).
//
// A/B FAMILY (variants 1-4, 7), three launches per pass (histogram / scan /
// scatter with a fixed grid, no inter-workgroup hand-off inside a launch):
//   1. radix_hist_kernel    every block counts the digits of its contiguous
//                           key range (16-B loads, LDS histogram per wave)
//   2. radix_scan_kernel    one wave per digit: exclusive scan of the counts
//                           in (digit, block) order
//   3. radix_scatter_*      every block re-walks its range tile by tile; keys
//                           get their stable rank inside the tile from
//                           wave-wide digit matching (8 ballots), the block's
//                           256 running offsets live in LDS.
//      wide (variant 7) : 1024 threads x 8 keys = 8192-key tile, reordered in
//                         LDS so every store instruction writes contiguous
//                         runs (32 keys per digit on average); ONE resident
//                         block per CU, so the 512 open output lines of all
//                         blocks of an XCD fit its 4 MiB L2 and partial lines
//                         complete there.  Measured 5.9 ms for 100 M 48-bit
//                         keys (rocPRIM onesweep via torch.sort: 8.0 ms).
//      variants 1-4 kept for A/B (tools/archive/sort_bench.py): 256-thread staged
//      tile 7.6 ms, direct per-lane stores 8.9 ms, 512-thread tile 6.6 ms.
// HBM traffic per pass: 8n (hist) + 12n read + 12n written (key64 + idx32).
#include "coalesce_internal.h"
#include "common.h"
#include "radix_util.h"

namespace {

using psa::digit_of;
using psa::kRadix;
using psa::wave_match;
using psa::wave_rank;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxBlocks = 1024;  // default cap on the fixed grid (tunable, see make_plan)
constexpr int kTileKeys = 8192;  // granule of the block ranges = production scatter tile
constexpr int kHistStep = 2048;  // keys per block iteration of the histogram kernel

int g_sort_variant = 0;

// Count one key into the wave's LDS histogram.  A wave whose keys all share
// the digit (sorted or small keys) would serialise 64 LDS atomics on one
// address, so that case adds the lane count once.
__device__ __forceinline__ void hist_add(uint32_t* whist, unsigned d, bool valid, int lane) {
  const unsigned long long active = __ballot(valid);
  if (active == 0ull) return;
  const int first = __ffsll(static_cast<long long>(active)) - 1;
  const unsigned d0 = __shfl(d, first);
  if (__ballot(valid && d != d0) == 0ull) {
    if (lane == first) atomicAdd(whist + d0, static_cast<uint32_t>(__popcll(active)));
  } else if (valid) {
    atomicAdd(whist + d, 1u);
  }
}

// ---- 1. histogram ---------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                  int tiles_per_block, int num_blocks,
                  uint32_t* __restrict__ counts,
                  uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t hist[kWaves][kRadix];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < kWaves * kRadix; i += kThreads) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTileKeys;
  int64_t end = begin + static_cast<int64_t>(tiles_per_block) * kTileKeys;
  end = end < n ? end : n;
  uint32_t* whist = hist[wave];
  const bool vec_ok = (reinterpret_cast<uintptr_t>(keys) & 15) == 0;
  if (vec_ok) {
    // 4 x 16-B loads in flight per lane: 2048 keys per block iteration
    const ulonglong2* kv = reinterpret_cast<const ulonglong2*>(keys);
    for (int64_t base = begin; base < end; base += kHistStep) {
      ulonglong2 v[4];
      int64_t idx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        idx[j] = base + 2 * (j * kThreads + tid);
        v[j] = idx[j] + 1 < end ? kv[idx[j] >> 1]
                                : make_ulonglong2(idx[j] < end ? keys[idx[j]] : 0ull, 0ull);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hist_add(whist, digit_of(v[j].x, shift), idx[j] < end, lane);
        hist_add(whist, digit_of(v[j].y, shift), idx[j] + 1 < end, lane);
      }
    }
  } else {
    for (int64_t base = begin; base < end; base += kThreads) {
      const int64_t i = base + tid;
      const bool valid = i < end;
      hist_add(whist, digit_of(valid ? keys[i] : 0ull, shift), valid, lane);
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

// ---- 3a. scatter, direct stores ---------------------------------------------
// PASS0: payload is the element's own index (nothing to read).
// LAST : payload is written as int64 to perm_out (the API dtype).
template <bool PASS0, bool LAST, int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS)
radix_scatter_direct_kernel(const uint64_t* __restrict__ keys_in,
                            const uint32_t* __restrict__ idx_in,
                            uint64_t* __restrict__ keys_out,  // may be null if LAST
                            uint32_t* __restrict__ idx_out,   // !LAST
                            int64_t* __restrict__ perm_out,   // LAST
                            int64_t n, int shift, int tiles_per_block,
                            int num_blocks, const uint32_t* __restrict__ offsets) {
  constexpr int TILE = THREADS * ITEMS;
  constexpr int WAVES = THREADS / 64;
  __shared__ uint32_t wcnt[2][WAVES][kRadix];  // double-buffered per-wave counts
  __shared__ uint32_t gbase[kRadix];            // running global offset per digit

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (tid < kRadix) gbase[tid] = offsets[static_cast<size_t>(tid) * num_blocks + blockIdx.x];
  for (int i = tid; i < WAVES * kRadix; i += THREADS) (&wcnt[0][0][0])[i] = 0;
  __syncthreads();

  const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTileKeys;
  int64_t blk_end = blk_begin + static_cast<int64_t>(tiles_per_block) * kTileKeys;
  blk_end = blk_end < n ? blk_end : n;
  int buf = 0;
  for (int64_t tile_begin = blk_begin; tile_begin < blk_end; tile_begin += TILE, buf ^= 1) {
    const int64_t rem = blk_end - tile_begin;
    const int tile_n = rem < TILE ? static_cast<int>(rem) : TILE;
    // wave-striped inside a contiguous chunk per wave: item i of lane l is
    // tile element  wave*ITEMS*64 + i*64 + l  (index order == (wave, i, lane))
    uint64_t key[ITEMS];
    uint32_t idx[ITEMS];
    uint32_t rank[ITEMS];
    const int wbase = wave * (ITEMS * 64);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = wbase + i * 64 + lane;
      const bool valid = p < tile_n;
      key[i] = valid ? keys_in[tile_begin + p] : ~0ull;
      if (PASS0) idx[i] = static_cast<uint32_t>(tile_begin + p);
      else idx[i] = valid ? idx_in[tile_begin + p] : 0u;
    }
    uint32_t* mine = wcnt[buf][wave];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const bool valid = wbase + i * 64 + lane < tile_n;
      const unsigned d = digit_of(key[i], shift);
      rank[i] = wave_rank(mine, d, valid, wave_match(d, valid));
    }
    __syncthreads();
    if (tid < kRadix) {  // thread d: global start of every wave's run of digit d; advance gbase
      uint32_t run = gbase[tid];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = wcnt[buf][w][tid];
        wcnt[buf][w][tid] = run;
        run += c;
        wcnt[buf ^ 1][w][tid] = 0;  // next tile's counters (idle since the last barrier pair)
      }
      gbase[tid] = run;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (wbase + i * 64 + lane < tile_n) {
        const int64_t dst = static_cast<int64_t>(mine[digit_of(key[i], shift)]) + rank[i];
        if (LAST) {
          if (keys_out) keys_out[dst] = key[i];
          perm_out[dst] = static_cast<int64_t>(idx[i]);
        } else {
          keys_out[dst] = key[i];
          idx_out[dst] = idx[i];
        }
      }
    }
  }
}

// ---- 3b. scatter, tile reordered in LDS first --------------------------------
template <bool PASS0, bool LAST>
__global__ void __launch_bounds__(kThreads)
radix_scatter_staged_kernel(const uint64_t* __restrict__ keys_in,
                            const uint32_t* __restrict__ idx_in,
                            uint64_t* __restrict__ keys_out,
                            uint32_t* __restrict__ idx_out,
                            int64_t* __restrict__ perm_out, int64_t n, int shift,
                            int tiles_per_block, int num_blocks,
                            const uint32_t* __restrict__ offsets) {
  constexpr int ITEMS = 8;
  constexpr int TILE = kThreads * ITEMS;
  __shared__ uint64_t skey[TILE];
  __shared__ uint32_t sidx[TILE];
  __shared__ uint32_t wcnt[kWaves][kRadix];
  __shared__ uint32_t gbase[kRadix];
  __shared__ int32_t gofs[kRadix];  // gbase - tile_digit_start
  __shared__ uint32_t wave_tot[kWaves];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  gbase[tid] = offsets[static_cast<size_t>(tid) * num_blocks + blockIdx.x];

  const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTileKeys;
  int64_t blk_end = blk_begin + static_cast<int64_t>(tiles_per_block) * kTileKeys;
  blk_end = blk_end < n ? blk_end : n;
  for (int64_t tile_begin = blk_begin; tile_begin < blk_end; tile_begin += TILE) {
    const int tile_n = (blk_end - tile_begin) < TILE ? static_cast<int>(blk_end - tile_begin) : TILE;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    uint64_t key[ITEMS];
    uint32_t idx[ITEMS];
    uint32_t rank[ITEMS];
    const int wbase = wave * (ITEMS * 64);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = wbase + i * 64 + lane;
      const bool valid = p < tile_n;
      key[i] = valid ? keys_in[tile_begin + p] : ~0ull;
      if (PASS0) idx[i] = static_cast<uint32_t>(tile_begin + p);
      else idx[i] = valid ? idx_in[tile_begin + p] : 0u;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      // match + rank item by item: the 9 ballots of one item die before the
      // next item's start (all items at once spills SGPRs)
      const bool valid = wbase + i * 64 + lane < tile_n;
      const unsigned d = digit_of(key[i], shift);
      rank[i] = wave_rank(wcnt[wave], d, valid, wave_match(d, valid));
    }
    __syncthreads();
    const unsigned d = tid;
    uint32_t c[kWaves];
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      c[w] = wcnt[w][d];
      wcnt[w][d] = tot;
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
    const uint32_t dstart = wprefix + incl - tot;
    const uint32_t gb = gbase[d];
    gofs[d] = static_cast<int32_t>(gb) - static_cast<int32_t>(dstart);
    gbase[d] = gb + tot;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wcnt[w][d] += dstart;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = wbase + i * 64 + lane;
      if (p < tile_n) {
        const uint32_t pos = wcnt[wave][digit_of(key[i], shift)] + rank[i];
        skey[pos] = key[i];
        sidx[pos] = idx[i];
      }
    }
    __syncthreads();
    for (int p = tid; p < tile_n; p += kThreads) {
      const uint64_t k = skey[p];
      const int64_t dst = static_cast<int64_t>(gofs[digit_of(k, shift)]) + p;
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

// ---- 3c. staged scatter, wide workgroup (bigger tile, longer store runs) --------
// Same algorithm as 3b with THREADS x ITEMS keys per tile.  A tile of T keys
// gives runs of T/256 keys per digit, i.e. T/32 bytes of keys per contiguous
// store run, and one resident block per CU keeps the 512 open output lines of
// every block inside the XCD's L2 until the next tile completes them.
// TWO_ROUND: keys and indices go through ONE LDS buffer one after the other
// (each thread keeps the destinations of its output slots in registers), which
// halves... (1/3 less) LDS per key and lets a 16K-key tile fit in 160 KB.
template <bool PASS0, bool LAST, int THREADS, int ITEMS, bool TWO_ROUND>
__global__ void __launch_bounds__(THREADS)
radix_scatter_wide_kernel(const uint64_t* __restrict__ keys_in,
                          const uint32_t* __restrict__ idx_in,
                          uint64_t* __restrict__ keys_out,
                          uint32_t* __restrict__ idx_out,
                          int64_t* __restrict__ perm_out, int64_t n, int shift,
                          int tiles_per_block, int num_blocks,
                          const uint32_t* __restrict__ offsets) {
  constexpr int WAVES = THREADS / 64;
  constexpr int TILE = THREADS * ITEMS;
  __shared__ uint64_t skey[TILE];
  __shared__ uint32_t sidx_store[TWO_ROUND ? 1 : TILE];
  __shared__ uint32_t wcnt[WAVES][kRadix];
  __shared__ uint32_t gbase[kRadix];
  __shared__ int32_t gofs[kRadix];
  __shared__ uint32_t wave_tot[4];
  uint32_t* sidx = TWO_ROUND ? reinterpret_cast<uint32_t*>(skey) : sidx_store;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid < kRadix) gbase[tid] = offsets[static_cast<size_t>(tid) * num_blocks + blockIdx.x];

  const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * tiles_per_block * kTileKeys;
  int64_t blk_end = blk_begin + static_cast<int64_t>(tiles_per_block) * kTileKeys;
  blk_end = blk_end < n ? blk_end : n;
  const int wbase = wave * (ITEMS * 64);
  // The next tile's keys/indices are requested before the current tile's
  // write-out, so their HBM latency overlaps the stores (one resident block
  // per CU has no other workgroup to hide it behind).
  uint64_t nkey[ITEMS];
  uint32_t nidx[ITEMS];
  auto request_tile = [&](int64_t tb) {
    const int64_t r = blk_end - tb;
    const int tn = r < TILE ? static_cast<int>(r) : TILE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = wbase + i * 64 + lane;
      const bool valid = p < tn;
      nkey[i] = valid ? keys_in[tb + p] : ~0ull;
      if (PASS0) nidx[i] = static_cast<uint32_t>(tb + p);
      else nidx[i] = valid ? idx_in[tb + p] : 0u;
    }
  };
  if (blk_begin < blk_end) request_tile(blk_begin);
  for (int64_t tile_begin = blk_begin; tile_begin < blk_end; tile_begin += TILE) {
    const int64_t rem = blk_end - tile_begin;
    const int tile_n = rem < TILE ? static_cast<int>(rem) : TILE;
    for (int i = tid; i < WAVES * kRadix; i += THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    uint64_t key[ITEMS];
    uint32_t idx[ITEMS];
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      key[i] = nkey[i];
      idx[i] = nidx[i];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      // match + rank item by item: the 9 ballots of one item die before the
      // next item's start (all items at once spills SGPRs)
      const bool valid = wbase + i * 64 + lane < tile_n;
      const unsigned d = digit_of(key[i], shift);
      rank[i] = wave_rank(wcnt[wave], d, valid, wave_match(d, valid));
    }
    __syncthreads();
    uint32_t tot = 0, incl = 0;
    if (tid < kRadix) {  // thread d: exclusive wave offsets of digit d, tile total
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = wcnt[w][tid];
        wcnt[w][tid] = tot;
        tot += c;
      }
      incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wave_tot[wave] = incl;
    }
    __syncthreads();
    if (tid < kRadix) {
      uint32_t wprefix = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) wprefix += w < wave ? wave_tot[w] : 0u;
      const uint32_t dstart = wprefix + incl - tot;
      const uint32_t gb = gbase[tid];
      gofs[tid] = static_cast<int32_t>(gb) - static_cast<int32_t>(dstart);
      gbase[tid] = gb + tot;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) wcnt[w][tid] += dstart;
    }
    __syncthreads();
    // tile position of every key (kept in rank[] from here on)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (wbase + i * 64 + lane < tile_n) {
        rank[i] += wcnt[wave][digit_of(key[i], shift)];
        skey[rank[i]] = key[i];
        if (!TWO_ROUND) sidx[rank[i]] = idx[i];
      }
    }
    __syncthreads();
    if (tile_begin + TILE < blk_end) request_tile(tile_begin + TILE);
    if (!TWO_ROUND) {
      for (int p = tid; p < tile_n; p += THREADS) {
        const uint64_t k = skey[p];
        const int64_t dst = static_cast<int64_t>(gofs[digit_of(k, shift)]) + p;
        if (LAST) {
          if (keys_out) keys_out[dst] = k;
          perm_out[dst] = static_cast<int64_t>(sidx[p]);
        } else {
          keys_out[dst] = k;
          idx_out[dst] = sidx[p];
        }
      }
      __syncthreads();
    } else {
      uint32_t dst[ITEMS];  // global slot of tile position tid + j*THREADS
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int p = tid + j * THREADS;
        if (p < tile_n) {
          const uint64_t k = skey[p];
          dst[j] = static_cast<uint32_t>(gofs[digit_of(k, shift)] + p);
          if (!LAST || keys_out) keys_out[dst[j]] = k;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (wbase + i * 64 + lane < tile_n) sidx[rank[i]] = idx[i];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int p = tid + j * THREADS;
        if (p < tile_n) {
          if (LAST) perm_out[dst[j]] = static_cast<int64_t>(sidx[p]);
          else idx_out[dst[j]] = sidx[p];
        }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(kThreads)
iota_kernel(int64_t* __restrict__ out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) out[i] = i;
}

// ===========================================================================
// Single-sweep passes (variant 5): the per-pass histogram kernel disappears.
//   os_hist_kernel   ONE read of the keys builds the 256-bin histograms of all
//                    passes (each pass turns its 256 counts into digit bases itself).
//   os_pass_kernel   tiles are handed out in order by an atomic ticket; a tile
//                    publishes its 256 digit counts, then walks back over its
//                    predecessors' published words until it meets an inclusive
//                    prefix (decoupled look-back).
// Inter-workgroup words follow the guide's granule rule (MI355X_MICROARCH.md
// "Valid forms", cdna_hip_programming.md Guideline 16 R2): ONE naturally
// aligned 8-byte word {pass-tagged flag, value} per (tile, digit), written by
// one relaxed agent-scope atomic store (sc1) and polled with relaxed
// agent-scope atomic loads — flag and payload cannot be seen torn or stale
// apart, so no fence is needed.  Tickets come from an atomic counter, so every
// predecessor of a running tile is itself running: waits always end.  Spins
// are bounded anyway (err word) so a bug cannot hang the GPU.
// ===========================================================================
constexpr int kOsThreads = 1024;
constexpr int kOsItems = 8;
constexpr int kOsTile = kOsThreads * kOsItems;  // 8192 keys
constexpr int kOsMaxPasses = 8;
constexpr uint32_t kOsSpinLimit = 1u << 22;
uint32_t g_os_spin_limit = kOsSpinLimit;  // test hook: psa_sort_set_spin_limit

__device__ __forceinline__ uint64_t os_pack(uint32_t flag, uint32_t value) {
  return (static_cast<uint64_t>(flag) << 32) | value;
}

__global__ void __launch_bounds__(kThreads)
os_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int passes, int first_bit,
               uint32_t* __restrict__ ghist /*[kOsMaxPasses][256]*/) {
  __shared__ uint32_t hist[kOsMaxPasses][kRadix];
  const int tid = threadIdx.x;
  for (int i = tid; i < kOsMaxPasses * kRadix; i += kThreads) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + tid; i < n; i += stride) {
    const uint64_t k = keys[i];
    for (int p = 0; p < passes; ++p) atomicAdd(&hist[p][digit_of(k, first_bit + 8 * p)], 1u);
  }
  __syncthreads();
  for (int i = tid; i < passes * kRadix; i += kThreads) {
    const uint32_t c = (&hist[0][0])[i];
    if (c) atomicAdd(ghist + i, c);
  }
}

template <bool PASS0, bool LAST, int THREADS, int ITEMS>
__global__ void __launch_bounds__(THREADS)
os_pass_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
               uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
               int64_t* __restrict__ perm_out, int64_t n, int shift, int pass,
               const uint32_t* __restrict__ ghist /*this pass: [256] digit counts of the whole input*/,
               uint64_t* __restrict__ status /*[num_tiles][256]*/,
               uint32_t* __restrict__ ticket, uint32_t* __restrict__ err, uint32_t spin_limit) {
  constexpr int WAVES = THREADS / 64;
  constexpr int TILE = THREADS * ITEMS;
  __shared__ uint64_t skey[TILE];
  __shared__ uint32_t wcnt[WAVES][kRadix];
  __shared__ int32_t gofs[kRadix];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t hist_tot[4];
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_fault;
  uint32_t* sidx = reinterpret_cast<uint32_t*>(skey);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t flag_agg = 2u * pass + 1u;   // pass-tagged: words of earlier
  const uint32_t flag_incl = 2u * pass + 2u;  // passes read as "not ready"

  if (tid == 0) {
    s_tile = atomicAdd(ticket, 1u);
    s_fault = 0u;
  }
  for (int i = tid; i < WAVES * kRadix; i += THREADS) (&wcnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const int64_t tile_begin = static_cast<int64_t>(tile) * TILE;
  if (tile_begin >= n) return;  // block-uniform
  const int tile_n = (n - tile_begin) < TILE ? static_cast<int>(n - tile_begin) : TILE;

  uint64_t key[ITEMS];
  uint32_t idx[ITEMS];
  uint32_t rank[ITEMS];
  const int wbase = wave * (ITEMS * 64);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int p = wbase + i * 64 + lane;
    const bool valid = p < tile_n;
    key[i] = valid ? keys_in[tile_begin + p] : ~0ull;
    if (PASS0) idx[i] = static_cast<uint32_t>(tile_begin + p);
    else idx[i] = valid ? idx_in[tile_begin + p] : 0u;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const bool valid = wbase + i * 64 + lane < tile_n;
    const unsigned d = digit_of(key[i], shift);
    rank[i] = wave_rank(wcnt[wave], d, valid, wave_match(d, valid));
  }
  __syncthreads();
  uint32_t tot = 0, incl = 0;
  uint32_t hcount = 0, hincl = 0;  // digit `tid` over the whole input, and its inclusive scan
  uint64_t* my_status = status + static_cast<size_t>(tile) * kRadix + tid;
  if (tid < kRadix) {
    hcount = ghist[tid];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = wcnt[w][tid];
      wcnt[w][tid] = tot;
      tot += c;
    }
    // publish this tile's count of digit `tid` as early as possible
    __hip_atomic_store(my_status, os_pack(tile == 0 ? flag_incl : flag_agg, tot),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    incl = tot;
    hincl = hcount;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      const uint32_t ho = __shfl_up(hincl, off);
      if (lane >= off) {
        incl += o;
        hincl += ho;
      }
    }
    if (lane == 63) {
      wave_tot[wave] = incl;
      hist_tot[wave] = hincl;
    }
  }
  __syncthreads();
  uint32_t dstart = 0, dbase = 0;  // first slot of digit `tid` inside the tile / in the output
  if (tid < kRadix) {
    uint32_t wprefix = 0, hprefix = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      wprefix += w < wave ? wave_tot[w] : 0u;
      hprefix += w < wave ? hist_tot[w] : 0u;
    }
    dstart = wprefix + incl - tot;
    dbase = hprefix + hincl - hcount;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) wcnt[w][tid] += dstart;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    if (wbase + i * 64 + lane < tile_n) {
      rank[i] += wcnt[wave][digit_of(key[i], shift)];
      skey[rank[i]] = key[i];
    }
  }
  if (tid < kRadix) {
    // decoupled look-back: sum predecessors' counts of digit `tid` until one of
    // them already carries its inclusive prefix
    uint32_t excl = 0;
    if (tile > 0) {
      int64_t t = static_cast<int64_t>(tile) - 1;
      uint32_t spins = 0;
      while (t >= 0) {
        const uint64_t w = __hip_atomic_load(status + static_cast<size_t>(t) * kRadix + tid,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t f = static_cast<uint32_t>(w >> 32);
        if (f == flag_incl) {
          excl += static_cast<uint32_t>(w);
          break;
        }
        if (f == flag_agg) {
          excl += static_cast<uint32_t>(w);
          --t;
          continue;
        }
        if (++spins > spin_limit) {  // never expected; keeps a bug from hanging the GPU
          // the fault word is set and made visible (agent scope) BEFORE the flag_incl store below: a
          // tile that takes this tile's (wrong) prefix finds it set when it looks
          __hip_atomic_store(err, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          __threadfence();
          s_fault = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __hip_atomic_store(my_status, os_pack(flag_incl, excl + tot), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    gofs[tid] = static_cast<int32_t>(dbase + excl) - static_cast<int32_t>(dstart);
    // A sort never returns garbage (storage.py:164-169): once any tile of any pass gave up,
    // the LAST pass stores -1 over its share of the outputs instead of a wrong order, so a
    // caller without a host read behind the sort cannot mistake the result for one either.
    // Every OTHER pass stops storing as well: a tile that gave up scatters to under-counted
    // (in-range) positions and leaves slots of its output unwritten, so the NEXT pass would read
    // keys that do not match the histograms its positions come from and could store past the
    // end of the buffers (seen as a memory fault with PSA_POISON_WORKSPACE=1 and the spin limit
    // at 0).  Within the faulting pass every position is an under-count of a true one: in range.
    if (tid == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_fault = 1u;
  }
  __syncthreads();
  if (s_fault != 0u) {  // block-uniform
    if (LAST) {
      for (int p = tid; p < tile_n; p += THREADS) {
        if (perm_out) perm_out[tile_begin + p] = -1;
        if (keys_out) keys_out[tile_begin + p] = ~0ull;
      }
    }
    return;
  }
  uint32_t dst[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int p = tid + j * THREADS;
    if (p < tile_n) {
      const uint64_t k = skey[p];
      dst[j] = static_cast<uint32_t>(gofs[digit_of(k, shift)] + p);
      if (!LAST || keys_out) keys_out[dst[j]] = k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (wbase + i * 64 + lane < tile_n) sidx[rank[i]] = idx[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int p = tid + j * THREADS;
    if (p < tile_n) {
      if (LAST) perm_out[dst[j]] = static_cast<int64_t>(sidx[p]);
      else idx_out[dst[j]] = sidx[p];
    }
  }
}

// Keys per tile of a single-sweep pass.  8192 keeps the store runs long once the input
// fills the chip; a mid-size input (a few hundred thousand keys) is one wave of workgroups
// either way and finishes sooner with less work per workgroup.
constexpr int64_t kOsSmallBelow = int64_t{1} << 20;  // the A/B shapes (variants 8-11) apply below this
constexpr int64_t kOsMidBelow = int64_t{1} << 19;    // production: 1024 x 4 below, 512 x 16 from here
constexpr int64_t kOsHugeFrom = int64_t{1} << 25;    // ... and 1024 x 16 from here
constexpr int kOsSmallTile = 2048;  // smallest tile any configuration uses (sizes the status words)

size_t os_status_bytes(int64_t n);

struct SortPlan {
  int passes;
  int64_t num_tiles;
  int tiles_per_block;
  int num_blocks;
  size_t keys_bytes, idx_bytes, counts_bytes, os_bytes, total_bytes;
};

bool single_sweep_variant(int variant) {
  return variant == 0 || variant == 5 || variant == 6 || (variant >= 8 && variant <= 11);
}

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

// look-back words of one pass: [tiles][256] x 8 bytes, sized for the smallest tile n may get
size_t os_status_bytes(int64_t n) {
  const int64_t tiles = psa::ceil_div(n > 0 ? n : 1, n < kOsSmallBelow ? kOsSmallTile : kOsTile);
  return align_up(static_cast<size_t>(tiles) * kRadix * sizeof(uint64_t), 256);
}

SortPlan make_plan(int64_t n, int64_t max_value) {
  SortPlan p;
  p.passes = (bits_for(max_value) + 7) / 8;
  p.num_tiles = psa::ceil_div(n > 0 ? n : 1, kTileKeys);
  // even tiles_per_block so a 4096-key scatter tile never straddles two blocks
  // test hook: variant / 16 (if non-zero) overrides the block cap in units of 256
  const int max_blocks = (g_sort_variant >> 4) > 0 ? (g_sort_variant >> 4) * 256 : kMaxBlocks;
  p.tiles_per_block = static_cast<int>(psa::ceil_div(p.num_tiles, max_blocks));
  if (p.tiles_per_block > 1 && (p.tiles_per_block & 1)) ++p.tiles_per_block;
  p.num_blocks = static_cast<int>(psa::ceil_div(p.num_tiles, p.tiles_per_block));
  p.keys_bytes = align_up(sizeof(uint64_t) * static_cast<size_t>(n), 256);
  p.idx_bytes = align_up(sizeof(uint32_t) * static_cast<size_t>(n), 256);
  p.counts_bytes =
      align_up(sizeof(uint32_t) * kRadix * (static_cast<size_t>(p.num_blocks) + 8), 256);
  // single-sweep area: status words, histograms / bases of all passes, tickets, err
  p.os_bytes = os_status_bytes(n) +
               align_up(2 * kOsMaxPasses * kRadix * sizeof(uint32_t), 256) + 256;
  p.total_bytes = 2 * p.keys_bytes + 2 * p.idx_bytes + p.counts_bytes + p.os_bytes;
  return p;
}

template <bool P0, bool L>
void launch_scatter(int variant, dim3 grid, hipStream_t s, const uint64_t* kin,
                    const uint32_t* iin, uint64_t* kout, uint32_t* iout,
                    int64_t* perm_out, int64_t n, int shift, const SortPlan& p,
                    const uint32_t* counts) {
#define PSA_SCATTER(KERNEL, THREADS)                                                   \
  hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), 0, s, kin, iin, kout, iout, perm_out, \
                     n, shift, p.tiles_per_block, p.num_blocks, counts)
  switch (variant) {
    case 1: PSA_SCATTER((radix_scatter_staged_kernel<P0, L>), kThreads); break;
    case 2: PSA_SCATTER((radix_scatter_direct_kernel<P0, L, 8, 256>), 256); break;
    case 3: PSA_SCATTER((radix_scatter_wide_kernel<P0, L, 512, 8, false>), 512); break;
    case 4: PSA_SCATTER((radix_scatter_wide_kernel<P0, L, 1024, 8, false>), 1024); break;
    default: PSA_SCATTER((radix_scatter_wide_kernel<P0, L, 1024, 8, true>), 1024); break;
  }
#undef PSA_SCATTER
}

// ===========================================================================
// Small inputs: the whole "sort by (row, col) + run-length structure" of a
// coalesce in ONE launch of ONE workgroup.
//
// At 10 k entries (BASELINE config 1) the multi-launch chain is pure latency:
// a dozen launches of ~5-10 us each (histogram, scan, three passes, run
// counts, block scan, write, ...) plus host reads, 135 us in all — slower than
// a tight single-threaded CPU loop (85-100 us).  One workgroup of 1024 threads
// does every step here with barriers instead of launches: keys are formed on
// the fly in pass 0, each radix pass is {digit histogram, 256-bin scan, stable
// ranks by wave-wide digit matching, scatter} over global ping-pong buffers
// (L2-resident at this size; __syncthreads orders the block's global writes
// and reads), and the last stage turns the sorted keys into (row, col), run
// starts and the run count.  Same stable permutation, bit for bit.
// ===========================================================================
constexpr int kSmallThreads = 1024;
constexpr int kSmallItems = 10;
constexpr int kSmallTile = kSmallThreads * kSmallItems;  // 10240 keys per tile step

__global__ void __launch_bounds__(kSmallThreads)
small_sort_unique_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                         int64_t mul, int64_t n, int passes,
                         uint64_t* kbuf0, uint64_t* kbuf1, uint32_t* ibuf0, uint32_t* ibuf1,
                         int64_t* __restrict__ out_row, int64_t* __restrict__ out_col,
                         int64_t* __restrict__ ptr, int64_t* __restrict__ perm,
                         int64_t* __restrict__ count_out) {
  // Every global access below is one of kSmallItems unrolled, independent loads
  // or stores per thread: a single workgroup has no other way to hide the
  // ~1 us memory latency (a first version with `for (i = tid; i < n; i +=
  // 1024)` loops spent 100 us on 10 k keys, one round trip per iteration).
  constexpr int WAVES = kSmallThreads / 64;
  constexpr int ITEMS = kSmallItems;
  __shared__ uint32_t wcnt[2][WAVES][kRadix];
  __shared__ uint32_t gbase[kRadix];
  __shared__ uint32_t wtot[WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wbase = wave * (ITEMS * 64);  // item i of lane l: tile element wbase + i*64 + l
  const bool single = n <= kSmallTile;

  const uint64_t* kin = kbuf0;  // pass 0 reads (row, col) instead
  const uint32_t* iin = ibuf0;
  uint64_t key[ITEMS];
  uint32_t idx[ITEMS];
  auto load_tile = [&](int p, int64_t tile_begin, int tile_n) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int q = wbase + i * 64 + lane;
      const bool valid = q < tile_n;
      if (p == 0) {
        key[i] = valid ? static_cast<uint64_t>(row[tile_begin + q] * mul + col[tile_begin + q]) : ~0ull;
        idx[i] = static_cast<uint32_t>(tile_begin + q);
      } else {
        key[i] = valid ? kin[tile_begin + q] : ~0ull;
        idx[i] = valid ? iin[tile_begin + q] : 0u;
      }
    }
  };

  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    uint64_t* ko = (p & 1) ? kbuf1 : kbuf0;
    uint32_t* io = (p & 1) ? ibuf1 : ibuf0;
    if (tid < kRadix) gbase[tid] = 0;
    for (int i = tid; i < 2 * WAVES * kRadix; i += kSmallThreads) (&wcnt[0][0][0])[i] = 0;
    __syncthreads();
    // ---- digit histogram of the whole array -> exclusive bases --------------
    for (int64_t tile_begin = 0; tile_begin < n; tile_begin += kSmallTile) {
      const int64_t rem = n - tile_begin;
      const int tile_n = rem < kSmallTile ? static_cast<int>(rem) : kSmallTile;
      load_tile(p, tile_begin, tile_n);
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (wbase + i * 64 + lane < tile_n) atomicAdd(&gbase[digit_of(key[i], shift)], 1u);
    }
    __syncthreads();
    if (tid < kRadix) {  // 256-bin exclusive scan: 4 waves, shuffle scan + wave totals
      const uint32_t c = gbase[tid];
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wtot[wave] = incl;
      gbase[tid] = incl - c;  // exclusive inside the wave; wave offset added below
    }
    __syncthreads();
    if (tid < kRadix) {
      uint32_t add = 0;
      for (int w = 0; w < wave; ++w) add += wtot[w];
      gbase[tid] += add;
    }
    __syncthreads();
    // ---- stable scatter, tile by tile (as radix_scatter_direct_kernel) ---------
    int buf = 0;
    for (int64_t tile_begin = 0; tile_begin < n; tile_begin += kSmallTile, buf ^= 1) {
      const int64_t rem = n - tile_begin;
      const int tile_n = rem < kSmallTile ? static_cast<int>(rem) : kSmallTile;
      if (!single) load_tile(p, tile_begin, tile_n);  // a single tile is still in registers
      uint32_t rank[ITEMS];
      uint32_t* mine = wcnt[buf][wave];
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        const bool valid = wbase + i * 64 + lane < tile_n;
        const unsigned d = digit_of(key[i], shift);
        rank[i] = wave_rank(mine, d, valid, wave_match(d, valid));
      }
      __syncthreads();
      if (tid < kRadix) {
        uint32_t run = gbase[tid];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
          const uint32_t c = wcnt[buf][w][tid];
          wcnt[buf][w][tid] = run;
          run += c;
          wcnt[buf ^ 1][w][tid] = 0;
        }
        gbase[tid] = run;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        if (wbase + i * 64 + lane < tile_n) {
          const uint32_t dst = mine[digit_of(key[i], shift)] + rank[i];
          ko[dst] = key[i];
          io[dst] = idx[i];
        }
      }
    }
    __syncthreads();  // this pass's writes are visible to the whole block
    kin = ko;
    iin = io;
  }
  // ---- run-length structure of the sorted keys, in the same striped order -----
  uint32_t heads_before = 0;  // distinct keys in earlier tiles
  for (int64_t tile_begin = 0; tile_begin < n; tile_begin += kSmallTile) {
    const int64_t rem = n - tile_begin;
    const int tile_n = rem < kSmallTile ? static_cast<int>(rem) : kSmallTile;
    uint64_t prev[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int q = wbase + i * 64 + lane;
      const bool valid = q < tile_n;
      const int64_t g = tile_begin + q;
      key[i] = valid ? kin[g] : 0ull;
      idx[i] = valid ? iin[g] : 0u;
      prev[i] = (valid && g > 0) ? kin[g - 1] : ~0ull;
    }
    // position of every head inside its wave: ballots, item after item
    uint32_t pos[ITEMS];
    uint32_t wave_heads = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int q = wbase + i * 64 + lane;
      const bool head = q < tile_n && (tile_begin + q == 0 || key[i] != prev[i]);
      const unsigned long long m = __ballot(head);
      const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                       __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      pos[i] = head ? wave_heads + below : 0xffffffffu;
      wave_heads += static_cast<uint32_t>(__popcll(m));
    }
    __syncthreads();  // wtot is free again
    if (lane == 0) wtot[wave] = wave_heads;
    __syncthreads();
    uint32_t before = heads_before, total = 0;
    for (int w = 0; w < WAVES; ++w) {
      if (w < wave) before += wtot[w];
      total += wtot[w];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int q = wbase + i * 64 + lane;
      if (q < tile_n) {
        const int64_t g = tile_begin + q;
        perm[g] = static_cast<int64_t>(idx[i]);
        if (pos[i] != 0xffffffffu) {
          const int64_t o = before + pos[i];
          const int64_t r = static_cast<int64_t>(key[i]) / mul;
          out_row[o] = r;
          out_col[o] = static_cast<int64_t>(key[i]) - r * mul;
          ptr[o] = g;
        }
      }
    }
    heads_before += total;
  }
  if (tid == 0) {
    ptr[heads_before] = n;
    *count_out = heads_before;
  }
}

}  // namespace

extern "C" {

size_t psa_coalesce_small_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return 2 * align_up(sizeof(uint64_t) * static_cast<size_t>(n), 256) +
         2 * align_up(sizeof(uint32_t) * static_cast<size_t>(n), 256);
}

int64_t psa_coalesce_small_max(void) { return 4 * kSmallTile; }

int psa_coalesce_small(const int64_t* row, const int64_t* col, int64_t n, int64_t M, int64_t N,
                       int64_t* out_row, int64_t* out_col, int64_t* ptr, int64_t* perm,
                       int64_t* count_out, void* workspace, size_t workspace_bytes,
                       psa_stream_t stream) {
  PSA_REQUIRE(n > 0 && n <= psa_coalesce_small_max(), "n out of range for the one-workgroup path");
  PSA_REQUIRE(M > 0 && N > 0, "empty matrix");
  PSA_REQUIRE(row && col && out_row && out_col && ptr && perm && count_out, "NULL pointer");
  if (workspace == nullptr || workspace_bytes < psa_coalesce_small_workspace_bytes(n)) {
    psa::set_error("psa_coalesce_small: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
  int passes = (bits_for(M * N) + 7) / 8;
  if (passes < 1) passes = 1;
  char* p = static_cast<char*>(workspace);
  const size_t kb = align_up(sizeof(uint64_t) * static_cast<size_t>(n), 256);
  const size_t ib = align_up(sizeof(uint32_t) * static_cast<size_t>(n), 256);
  uint64_t* k0 = reinterpret_cast<uint64_t*>(p);
  uint64_t* k1 = reinterpret_cast<uint64_t*>(p + kb);
  uint32_t* i0 = reinterpret_cast<uint32_t*>(p + 2 * kb);
  uint32_t* i1 = reinterpret_cast<uint32_t*>(p + 2 * kb + ib);
  hipLaunchKernelGGL(small_sort_unique_kernel, dim3(1), dim3(kSmallThreads), 0,
                     psa::as_stream(stream), row, col, N, n, passes, k0, k1, i0, i1, out_row,
                     out_col, ptr, perm, count_out);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_sort_set_variant(int variant) {
  const int prev = g_sort_variant;
  g_sort_variant = variant;
  return prev;
}

int64_t psa_sort_set_spin_limit(int64_t limit) {
  const int64_t prev = g_os_spin_limit;
  g_os_spin_limit = limit < 0 ? kOsSpinLimit : static_cast<uint32_t>(limit > 0xffffffffll ? 0xffffffffll : limit);
  return prev;
}

size_t psa_index_sort_workspace_bytes(int64_t n, int64_t max_value) {
  if (n <= 0) return 0;
  return make_plan(n, max_value).total_bytes;
}

// Shared driver.  perm mode: pay_in == NULL (payload = element index),
// perm_out int64.  pairs mode: pay_in / pay_out carry a caller-defined 32-bit
// payload (e.g. the fp32 value of a COO entry) and perm_out is NULL.
static int sort_impl(const char* who, const int64_t* keys, const uint32_t* pay_in,
                     int64_t n, int64_t max_value, int64_t* sorted_out,
                     int64_t* perm_out, uint32_t* pay_out, void* workspace,
                     size_t workspace_bytes, hipStream_t s, int first_bit = 0, bool prepared = false) {
  // prepared: the caller has zeroed the single-sweep area and filled the histograms and digit
  // bases of every pass already (psa::sort_areas; the coalesce chain does both in its key pass)
  // first_bit > 0: order by the bit field (key >> first_bit) < max_value only;
  // the bits below ride along inside the key (stable, like any other payload)
  const SortPlan p = make_plan(n, max_value);
  if (p.passes == 0) {  // all keys equal: the stable order is the input order
    if (perm_out) {
      hipLaunchKernelGGL(iota_kernel, dim3(static_cast<unsigned>(psa::ceil_div(n, kThreads))),
                         dim3(kThreads), 0, s, perm_out, n);
      PSA_LAUNCH_CHECK();
    } else {
      PSA_HIP(hipMemcpyAsync(pay_out, pay_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, s));
    }
    if (sorted_out)
      PSA_HIP(hipMemcpyAsync(sorted_out, keys, sizeof(int64_t) * n,
                             hipMemcpyDeviceToDevice, s));
    return PSA_OK;
  }
  if (workspace == nullptr || workspace_bytes < p.total_bytes) {
    psa::set_error(std::string(who) + ": workspace too small (need " +
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
  const uint32_t* iin = pay_in;
  const dim3 grid(static_cast<unsigned>(p.num_blocks)), block(kThreads);
  const int variant = g_sort_variant & 15;

  // 0 (production) and 5: single-sweep passes with decoupled look-back
  // (0: 512 threads x 16 keys, two blocks per CU; 5: 1024 x 8, one block);
  // 1-4, 7: the three-launch-per-pass family, kept for A/B.
  PSA_REQUIRE(!prepared || variant == 0, "prepared histograms need the single-sweep passes");
  if (single_sweep_variant(variant)) {
    char* os = ws + 2 * p.keys_bytes + 2 * p.idx_bytes + p.counts_bytes;
    // tile shape: 0 production rule, 5 / 8-11 forced for A/B (tools/sort_tiles.py)
    int os_threads = 512, os_items = 16;
    if (variant == 5) os_threads = 1024, os_items = 8;
    if (variant == 6) os_threads = 1024, os_items = 16;  // A/B: 16 384-key tiles, one workgroup per CU
    if (variant == 0 && n < kOsMidBelow) os_threads = 1024, os_items = 4;  // profiles/r02_sort_tiles.txt
    // very large inputs: 16 384-key tiles (one workgroup per CU) — 64-key store runs per digit; 100 M 48-bit keys
    // 3.75 -> 3.44 ms, no gain at 20 M (profiles/r02_sort_tiles.txt)
    if (variant == 0 && n >= kOsHugeFrom) os_threads = 1024, os_items = 16;
    if (n < kOsSmallBelow) {
      if (variant == 8) os_threads = 512, os_items = 4;
      if (variant == 9) os_threads = 512, os_items = 8;
      if (variant == 10) os_threads = 256, os_items = 8;
      if (variant == 11) os_threads = 1024, os_items = 4;
    }
    const int os_tile = os_threads * os_items;
    const size_t os_tiles = static_cast<size_t>(psa::ceil_div(n, os_tile));
    const size_t status_bytes = os_status_bytes(n);
    uint64_t* status = reinterpret_cast<uint64_t*>(os);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(os + status_bytes);
    uint32_t* tickets = ghist + 2 * kOsMaxPasses * kRadix;  // [kOsMaxPasses] + err word
    uint32_t* err = tickets + kOsMaxPasses;
    if (!prepared) {
      PSA_ZERO(os, p.os_bytes, s);
      const int hist_blocks = static_cast<int>(psa::ceil_div(n, kThreads * 16) < 2048
                                                   ? psa::ceil_div(n, kThreads * 16) : 2048);
      hipLaunchKernelGGL(os_hist_kernel, dim3(hist_blocks), block, 0, s, kin, n, p.passes, first_bit, ghist);
    }
    const dim3 os_grid(static_cast<unsigned>(os_tiles)), os_block(os_threads);
    for (int pass = 0; pass < p.passes; ++pass) {
      const int shift = first_bit + 8 * pass;
      const bool last = pass == p.passes - 1;
      const bool iota_payload = pass == 0 && pay_in == nullptr;
      const bool widen = last && perm_out != nullptr;
      uint64_t* kout = last ? reinterpret_cast<uint64_t*>(sorted_out) : kbuf[pass & 1];
      uint32_t* iout = last ? pay_out : ibuf[pass & 1];
#define PSA_OS1(P0, L, T, I)                                                                  \
  hipLaunchKernelGGL((os_pass_kernel<P0, L, T, I>), os_grid, os_block, 0, s, kin, iin, kout, \
                     iout, perm_out, n, shift, pass, ghist + pass * kRadix, status,          \
                     tickets + pass, err, g_os_spin_limit)
#define PSA_OS(P0, L)                                                   \
  do {                                                                  \
    if (os_threads == 512 && os_items == 16) PSA_OS1(P0, L, 512, 16);   \
    else if (os_threads == 1024 && os_items == 8) PSA_OS1(P0, L, 1024, 8); \
    else if (os_threads == 1024 && os_items == 16) PSA_OS1(P0, L, 1024, 16); \
    else if (os_threads == 512 && os_items == 4) PSA_OS1(P0, L, 512, 4);  \
    else if (os_threads == 512 && os_items == 8) PSA_OS1(P0, L, 512, 8);  \
    else if (os_threads == 256 && os_items == 8) PSA_OS1(P0, L, 256, 8);  \
    else PSA_OS1(P0, L, 1024, 4);                                       \
  } while (0)
      if (iota_payload && widen) PSA_OS(true, true);
      else if (iota_payload) PSA_OS(true, false);
      else if (widen) PSA_OS(false, true);
      else PSA_OS(false, false);
#undef PSA_OS
#undef PSA_OS1
      PSA_LAUNCH_CHECK();
      kin = kout;
      iin = iout;
    }
    return PSA_OK;
  }

  for (int pass = 0; pass < p.passes; ++pass) {
    const int shift = first_bit + 8 * pass;
    const bool last = pass == p.passes - 1;
    const bool iota_payload = pass == 0 && pay_in == nullptr;
    const bool widen = last && perm_out != nullptr;  // int64 permutation output
    PSA_ZERO(digit_total, sizeof(uint32_t) * kRadix, s);
    hipLaunchKernelGGL(radix_hist_kernel, grid, block, 0, s, kin, n, shift,
                       p.tiles_per_block, p.num_blocks, counts, digit_total);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(kRadix / kWaves), block, 0, s,
                       counts, digit_total, p.num_blocks);
    uint64_t* kout = last ? reinterpret_cast<uint64_t*>(sorted_out) : kbuf[pass & 1];
    uint32_t* iout = last ? pay_out : ibuf[pass & 1];
    if (iota_payload && widen)
      launch_scatter<true, true>(variant, grid, s, kin, iin, kout, iout, perm_out, n, shift, p, counts);
    else if (iota_payload)
      launch_scatter<true, false>(variant, grid, s, kin, iin, kout, iout, perm_out, n, shift, p, counts);
    else if (widen)
      launch_scatter<false, true>(variant, grid, s, kin, iin, kout, iout, perm_out, n, shift, p, counts);
    else
      launch_scatter<false, false>(variant, grid, s, kin, iin, kout, iout, perm_out, n, shift, p, counts);
    PSA_LAUNCH_CHECK();
    kin = kout;
    iin = iout;
  }
  return PSA_OK;
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
  return sort_impl("psa_index_sort", keys, nullptr, n, max_value, sorted_out, perm_out,
                   nullptr, workspace, workspace_bytes, psa::as_stream(stream));
}

}  // extern "C"

namespace psa {
// The words of `workspace` a caller fills to run the passes without the sort's own
// zero / histogram / scan launches (sort_prepared below).  zero_begin[0 .. zero_bytes) must be
// zero before anything else touches it; ghist[p * 256 + d] counts the keys whose digit p is d
// (every pass scans its 256 counts itself).  passes == 0: nothing to prepare.
SortAreas sort_areas(void* workspace, int64_t n, int64_t max_value) {
  SortAreas a{};
  if (n <= 0 || workspace == nullptr) return a;
  const SortPlan p = make_plan(n, max_value);
  a.passes = p.passes;
  char* os = static_cast<char*>(workspace) + 2 * p.keys_bytes + 2 * p.idx_bytes + p.counts_bytes;
  a.zero_begin = os;
  a.zero_bytes = p.os_bytes;
  a.ghist = reinterpret_cast<uint32_t*>(os + os_status_bytes(n));
  return a;
}

bool sort_takes_prepared() { return (g_sort_variant & 15) == 0; }

int sort_prepared(const int64_t* keys, const uint32_t* pay_in, int64_t n, int64_t max_value,
                  int64_t* sorted_out, int64_t* perm_out, uint32_t* pay_out, void* workspace,
                  size_t workspace_bytes, hipStream_t s) {
  return sort_impl("sort_prepared", keys, pay_in, n, max_value, sorted_out, perm_out, pay_out, workspace,
                   workspace_bytes, s, 0, true);
}

// Device address of the sort's look-back diagnostic word inside `workspace` (NULL when the
// sort had nothing to do): non-zero after a bounded spin of a pass gave up, i.e. the order is
// invalid.  Callers that read a count from the device anyway fold it into that read (chain.hip).
const uint32_t* sort_fault_word(const void* workspace, int64_t n, int64_t max_value) {
  if (n <= 0 || workspace == nullptr) return nullptr;
  const SortPlan p = make_plan(n, max_value);
  if (p.passes == 0) return nullptr;
  // only the single-sweep passes zero and use the word; behind an A/B variant (1-4, 7) it is
  // untouched scratch (0xff under PSA_POISON_WORKSPACE) and those passes cannot fault
  if (!single_sweep_variant(g_sort_variant & 15)) return nullptr;
  const char* os = static_cast<const char*>(workspace) + 2 * p.keys_bytes + 2 * p.idx_bytes +
                   p.counts_bytes;
  return reinterpret_cast<const uint32_t*>(os + os_status_bytes(n)) + 2 * kOsMaxPasses * kRadix + kOsMaxPasses;
}
}  // namespace psa

extern "C" {

int psa_index_sort_status(const void* workspace, int64_t n, int64_t max_value,
                          psa_stream_t stream) {
  const uint32_t* err = psa::sort_fault_word(workspace, n, max_value);
  if (err == nullptr) return 0;
  uint32_t host = 0;
  hipStream_t s = psa::as_stream(stream);
  if (hipMemcpyAsync(&host, err, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    psa::set_error("psa_index_sort_status: copy failed");
    return -1;
  }
  return static_cast<int>(host);
}

int psa_sort_pairs_u32(const int64_t* keys, const void* payload, int64_t n,
                       int64_t max_value, int64_t* sorted_out, void* payload_out,
                       void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(keys && payload && sorted_out && payload_out, "NULL pointer");
  PSA_REQUIRE(max_value >= 0, "max_value must be >= 0");
  PSA_REQUIRE(psa::aligned(payload, 4) && psa::aligned(payload_out, 4), "payload must be 4-byte aligned");
  if (n >= (1ll << 31)) {
    psa::set_error("psa_sort_pairs_u32: n >= 2^31 not supported by this build");
    return PSA_ERR_UNSUPPORTED;
  }
  return sort_impl("psa_sort_pairs_u32", keys, static_cast<const uint32_t*>(payload), n,
                   max_value, sorted_out, nullptr, static_cast<uint32_t*>(payload_out),
                   workspace, workspace_bytes, psa::as_stream(stream));
}

int psa_sort_pairs_u32_field(const int64_t* keys, const void* payload, int64_t n, int first_bit,
                             int64_t max_value, int64_t* sorted_out, void* payload_out,
                             void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(n >= 0, "negative size");
  if (n == 0) return PSA_OK;
  PSA_REQUIRE(keys && payload && sorted_out && payload_out, "NULL pointer");
  PSA_REQUIRE(max_value >= 0 && first_bit >= 0 && first_bit < 64, "bad bit field");
  PSA_REQUIRE(first_bit + bits_for(max_value) <= 64, "bit field runs past bit 63");
  PSA_REQUIRE(psa::aligned(payload, 4) && psa::aligned(payload_out, 4), "payload must be 4-byte aligned");
  if (n >= (1ll << 31)) {
    psa::set_error("psa_sort_pairs_u32_field: n >= 2^31 not supported by this build");
    return PSA_ERR_UNSUPPORTED;
  }
  return sort_impl("psa_sort_pairs_u32_field", keys, static_cast<const uint32_t*>(payload), n,
                   max_value, sorted_out, nullptr, static_cast<uint32_t*>(payload_out),
                   workspace, workspace_bytes, psa::as_stream(stream), first_bit);
}

}  // extern "C"
