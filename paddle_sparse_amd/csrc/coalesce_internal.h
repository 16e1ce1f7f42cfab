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

// psa_segment_reduce whose segment count lives on the device: the grid is sized
// for nseg_bound segments, segments at or beyond *nseg_dev are skipped.
int segment_reduce_dev(int reduce, int dtype, const void* src, const int64_t* perm,
                       const int64_t* ptr, int64_t nseg_bound, const int64_t* nseg_dev,
                       int64_t D, int64_t n_hint, void* out, hipStream_t s);

}  // namespace psa
