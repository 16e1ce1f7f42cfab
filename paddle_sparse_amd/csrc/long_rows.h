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

}  // namespace psa
