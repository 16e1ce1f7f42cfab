// Edge-balanced SpMM forward (spmm_eb.hip): host entry used by psa_spmm's dispatch.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace psa {

// Shapes the edge-balanced kernels serve: rows of whole 16-byte units (K % 4 == 0 for fp32,
// K % 8 == 0 for fp16 / bf16 operands), sizes whose edge and row ids fit 31 bits.
// half: 0 fp32, 1 fp16, 2 bf16 (mat, hot_rows and out; values, sums and partials are fp32).
bool eb_supported(int64_t M, int64_t K, int64_t nnz, int half = 0);

// Scratch: [row ids built from rowptr when the caller has none] + two partial
// slots per edge range (+ their winners for min/max with arg tracking).
size_t eb_workspace_bytes(bool minmax, int64_t K, int64_t nnz, int half = 0);

// Masked sum (grad of the dense operand of spmm_min / max over the CSC view): a term counts for column k
// only where words[c, k] — c the edge's gathered row — equals the edge's tag.  Entries of `width` bytes.
struct EbMask {
  const void* words;      // [rows of mat, K]
  const void* hot_words;  // the rows of `hot_rows`, compact (NULL without a hot copy)
  const void* tags;       // [nnz], in edge order
  int width;              // 1 or 2
};

// red: 0 sum, 1 min, 2 max (the R_* ids of spmm.hip).  row may be NULL: it is
// then derived from rowptr into the workspace (one ptr2ind launch).
int launch_spmm_eb(int red, int mean, const int64_t* rowptr, const int64_t* row,
                   const int64_t* col, const float* val, const void* mat, void* out, int64_t ldo,
                   int64_t* arg_out, uint8_t* arg_bytes, int arg_width, int64_t M, int64_t N, int64_t K,
                   int64_t nnz, const void* hot_rows, int64_t num_hot, void* workspace, size_t workspace_bytes,
                   bool nt_gather, int range_len_override, int dbg, hipStream_t s, int half = 0,
                   const EbMask* mask = nullptr);

}  // namespace psa
