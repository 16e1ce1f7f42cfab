// Entry points of coalesce.hip used by the two-call coalesce chain (chain.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace psa {

// psa_unique_write with the distinct (row, col) pairs packed as the [2, count]
// index of the functional API: row' at index_out[0 .. count), col' at
// index_out[count .. 2 count), count read on the device.
int unique_write_packed(const int64_t* sorted_keys, int64_t n, int64_t N, const void* workspace,
                        const int64_t* count, int64_t* ptr_out, int64_t* index_out, hipStream_t s);

// unique_write_packed and segment_reduce_dev in ONE launch for 4-byte values (PSA_F32 / PSA_I32) that rode the sort as its
// payload: the thread that
// writes a distinct pair reduces its run of `payload` (sorted order) into value_out; no ptr array.
int unique_write_reduce_packed(int reduce, int dtype, const int64_t* sorted_keys, int64_t n, int64_t N,
                               const void* workspace, const int64_t* count, int64_t* index_out, const void* payload,
                               void* value_out, hipStream_t s);

// psa_segment_reduce whose segment count lives on the device: the grid is sized
// for nseg_bound segments, segments at or beyond *nseg_dev are skipped.
int segment_reduce_dev(int reduce, int dtype, const void* src, const int64_t* perm,
                       const int64_t* ptr, int64_t nseg_bound, const int64_t* nseg_dev,
                       int64_t D, int64_t n_hint, void* out, hipStream_t s);

// psa_unique_count for the coalesce chain: the block scan also writes the chain's status
// words — status[0] = count, status[1] = flag bits from status[2] (range), status[3]
// (inversion) and *fault (bit 2).
int unique_count_chain(const int64_t* sorted_keys, int64_t n, void* workspace, int64_t* status,
                       const uint32_t* fault, hipStream_t s);

// ---- sort.hip: the radix passes on histograms the caller has built ----------------------------
struct SortAreas {
  void* zero_begin;   // [zero_bytes) must be zeroed first (look-back words, tickets, histograms)
  size_t zero_bytes;
  uint32_t* ghist;    // [passes][256] digit counts, filled by the caller
  int passes;         // 8-bit passes for keys below max_value (0: all keys equal, nothing to do)
};
SortAreas sort_areas(void* workspace, int64_t n, int64_t max_value);
// perm mode (pay_in NULL, perm_out int64) or pairs mode (pay_in / pay_out 32-bit payload), as
// psa_index_sort / psa_sort_pairs_u32, minus their zero / histogram / scan launches
int sort_prepared(const int64_t* keys, const uint32_t* pay_in, int64_t n, int64_t max_value,
                  int64_t* sorted_out, int64_t* perm_out, uint32_t* pay_out, void* workspace,
                  size_t workspace_bytes, hipStream_t s);
bool sort_takes_prepared();  // false while an A/B variant of the sort is selected (psa_sort_set_variant)
const uint32_t* sort_fault_word(const void* workspace, int64_t n, int64_t max_value);

}  // namespace psa
