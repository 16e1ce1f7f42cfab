// Work list of long CSR rows, shared by the SpMM forward and backward kernels.
//
// One wave pulls only a few GB/s (8 gathers in flight), so a 40 000-edge row
// would keep its wave busy for milliseconds while the rest of the chip idles.
// Rows longer than kLongRow are therefore NOT processed by their row wave: it
// records (row, first chunk) with ONE 64-bit atomic {rows << 32 | chunks}, so
// the list order equals the chunk order and a chunk id can be mapped back to
// its row by a binary search over first_chunk.  A second launch gives every
// kLongChunk-edge chunk its own wave.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace psa {

constexpr int kLongRow = 128;    // rows with more edges take the chunked path
constexpr int kLongChunk = 128;  // edges per chunk wave
// chunk / combine launches: few, fat workgroups (512 x 1024 threads = 8192
// waves, the whole chip) — when no row is long these launches are pure
// dispatch overhead, which grows with the number of workgroups, not threads
constexpr int kLongBlocks = 512;
constexpr int kLongThreads = 1024;

struct LongEntry {
  int64_t row;
  uint32_t first_chunk;
  uint32_t num_chunks;
};

inline int64_t max_long_rows(int64_t nnz) { return nnz / (kLongRow + 1) + 1; }
inline int64_t max_long_chunks(int64_t nnz) { return nnz / kLongChunk + max_long_rows(nnz) + 1; }
inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
// bytes of {counter (256-B slot), list}
inline size_t long_list_bytes(int64_t nnz) {
  return 256 + align256(sizeof(LongEntry) * static_cast<size_t>(max_long_rows(nnz)));
}

// Lane 0 of a row wave hands a long row to the chunk kernel.
__device__ __forceinline__ void push_long_row(unsigned long long* ctr, LongEntry* list,
                                              int64_t row, int64_t deg) {
  const uint32_t chunks = static_cast<uint32_t>((deg + kLongChunk - 1) / kLongChunk);
  const unsigned long long old = atomicAdd(ctr, (1ull << 32) | chunks);
  LongEntry e;
  e.row = row;
  e.first_chunk = static_cast<uint32_t>(old & 0xffffffffull);
  e.num_chunks = chunks;
  list[old >> 32] = e;
}

// Entry owning chunk c: the largest slot with first_chunk <= c.
__device__ __forceinline__ LongEntry find_long_entry(const LongEntry* list, int nrows, uint32_t c) {
  int lo = 0, hi = nrows - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (list[mid].first_chunk <= c) lo = mid;
    else hi = mid - 1;
  }
  return list[lo];
}

// ---- list pre-pass (for the kernels that run chunk waves and row waves as two roles of ONE launch) ----
constexpr int kFindThreads = 256;
constexpr int kFindWaves = kFindThreads / 64;
constexpr int kFindIters = 4;   // x 256 rows per workgroup of find_long_rows_kernel

namespace {  // one copy per translation unit (the library is built without relocatable device code)

// Builds the long-row list {row, first chunk, chunks}, list order = chunk order =
// row order inside a workgroup.  ONE atomic per workgroup of 1024 rows reserves
// {long rows, chunks} for all of them: atomics on the single counter complete
// one after the other (~3 ns each), and a power-law graph has tens of thousands
// of long rows (R-MAT scale 21: 63 us with one atomic per row or per wave).
// Pass 1 counts per (trip, wave), lane 0 of the block scans those 64 pairs and
// reserves; pass 2 recomputes the (L2-hot) degrees and writes the entries.
__global__ void __launch_bounds__(kFindThreads)
find_long_rows_kernel(const int64_t* __restrict__ rowptr, int64_t M,
                      unsigned long long* __restrict__ ctr, LongEntry* __restrict__ list) {
  __shared__ uint32_t s_rows[kFindIters][kFindWaves];
  __shared__ uint32_t s_chunks[kFindIters][kFindWaves];
  __shared__ uint32_t s_any;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * (kFindThreads * kFindIters);
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  bool any = false;
  for (int it = 0; it < kFindIters; ++it) {
    const int64_t r = base + it * kFindThreads + threadIdx.x;
    const int64_t deg = r < M ? rowptr[r + 1] - rowptr[r] : 0;
    const bool is_long = deg > kLongRow;
    uint32_t chunks = is_long ? static_cast<uint32_t>((deg + kLongChunk - 1) / kLongChunk) : 0u;
    const unsigned long long mask = __ballot(is_long);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) chunks += __shfl_xor(chunks, off);
    if (lane == 0) {
      s_rows[it][wave] = static_cast<uint32_t>(__popcll(mask));
      s_chunks[it][wave] = chunks;
    }
    any |= mask != 0;
  }
  if (any && lane == 0) s_any = 1;
  __syncthreads();
  if (s_any == 0) return;  // block-uniform: no atomic at all for blocks of short rows
  if (threadIdx.x == 0) {
    uint32_t nrows = 0, nchunks = 0;
    for (int it = 0; it < kFindIters; ++it) {
      for (int w = 0; w < kFindWaves; ++w) {  // exclusive prefixes in (trip, wave) = row order
        const uint32_t a = s_rows[it][w], b = s_chunks[it][w];
        s_rows[it][w] = nrows;
        s_chunks[it][w] = nchunks;
        nrows += a;
        nchunks += b;
      }
    }
    const unsigned long long old = atomicAdd(ctr, (static_cast<unsigned long long>(nrows) << 32) | nchunks);
    for (int it = 0; it < kFindIters; ++it) {
      for (int w = 0; w < kFindWaves; ++w) {
        s_rows[it][w] += static_cast<uint32_t>(old >> 32);
        s_chunks[it][w] += static_cast<uint32_t>(old & 0xffffffffull);
      }
    }
  }
  __syncthreads();
  for (int it = 0; it < kFindIters; ++it) {
    const int64_t r = base + it * kFindThreads + threadIdx.x;
    const int64_t deg = r < M ? rowptr[r + 1] - rowptr[r] : 0;
    const bool is_long = deg > kLongRow;
    const unsigned long long mask = __ballot(is_long);
    if (mask == 0) continue;  // wave-uniform
    const uint32_t chunks = is_long ? static_cast<uint32_t>((deg + kLongChunk - 1) / kLongChunk) : 0u;
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (is_long) {
      LongEntry e;
      e.row = r;
      e.first_chunk = s_chunks[it][wave] + incl - chunks;
      e.num_chunks = chunks;
      list[s_rows[it][wave] + __popcll(mask & ((1ull << lane) - 1ull))] = e;
    }
  }
}


}  // namespace

}  // namespace psa
