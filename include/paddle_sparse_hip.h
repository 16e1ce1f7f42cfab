/*
 * paddle_sparse_hip.h — C-ABI of the MI355X (gfx950) sparse-op core.
 *
 * This is the drop-in boundary: the entry points a Paddle custom-op shim
 * (PD_BUILD_OP) binds in place of the reference's
 * csrc/cpu + csrc/cuda kernels.  Plain pointers and sizes only; no framework
 * types.  All pointers are DEVICE pointers (hipMalloc'd or framework-owned
 * HBM) unless a parameter says "host".  Every call enqueues work on `stream`
 * (a hipStream_t passed as void*) and returns without synchronising, exactly
 * like the reference's CUDA launchers (csrc/cuda/convert_cuda.cu:34-38).
 * Outputs are caller-allocated (the shim allocates them with the framework's
 * allocator, as the reference does with paddle::empty).
 *
 * Return value: PSA_OK (0) or a psa_status error; psa_last_error() gives the
 * message the shim should PD_THROW (reference errors are C++ exceptions,
 * csrc/cpu/utils.h:6-9).
 *
 * Index dtype is int64 everywhere (reference asserts it:
 * paddle_sparse/storage.py:61,94,101).  `reduce` strings of the Python API
 * ("sum"/"add"/"mean"/"min"/"max", paddle_sparse/testing.py:10) map to
 * psa_reduce.
 */
#ifndef PADDLE_SPARSE_HIP_H_
#define PADDLE_SPARSE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSA_ABI_VERSION 1

typedef void* psa_stream_t; /* hipStream_t */

typedef enum psa_status {
  PSA_OK = 0,
  PSA_ERR_INVALID_ARG = 1, /* bad pointer / size / enum */
  PSA_ERR_HIP = 2,         /* a HIP runtime call failed */
  PSA_ERR_WORKSPACE = 3,   /* workspace too small */
  PSA_ERR_UNSUPPORTED = 4  /* dtype / shape not supported by this build */
} psa_status;

typedef enum psa_reduce {
  PSA_SUM = 0, /* "sum" and "add" */
  PSA_MEAN = 1,
  PSA_MIN = 2,
  PSA_MAX = 3
} psa_reduce;

/* Value dtypes accepted by the dtype-generic entry points (gather, segment
 * reduce, coalesce).  SpMM is fp32 (the north-star dtype). */
typedef enum psa_dtype {
  PSA_F32 = 0,
  PSA_F64 = 1,
  PSA_I32 = 2,
  PSA_I64 = 3,
  PSA_F16 = 4,
  PSA_BF16 = 5
} psa_dtype;

/* ---- library ---------------------------------------------------------- */

/* Message of the last failing call on this host thread ("" if none). */
const char* psa_last_error(void);

/* PSA_ABI_VERSION the library was built with. */
int psa_abi_version(void);

/* Replaces sparse_cuda_version (csrc/version.cpp:14-22).  Always -1 on the
 * HIP build so that paddle_sparse/__init__.py:18-32 skips its CUDA-major
 * check. */
int64_t psa_sparse_cuda_version(void);

/* ---- index conversion (bit-exact) ------------------------------------- */

/* Replaces ind2ptr (csrc/convert.cpp:13-23 -> csrc/cuda/convert_cuda.cu:6-40,
 * CPU text csrc/cpu/convert_cpu.cpp:6-30).
 * ind: int64[numel], sorted ascending, values in [0, M).  out: int64[M+1],
 * out[r] = #{e : ind[e] < r}.  numel == 0 -> out is all zeros. */
int psa_ind2ptr(const int64_t* ind, int64_t numel, int64_t M, int64_t* out,
                psa_stream_t stream);

/* Replaces ptr2ind (csrc/convert.cpp:46-56 -> csrc/cuda/convert_cuda.cu:42-68,
 * CPU text csrc/cpu/convert_cpu.cpp:32-48).
 * ptr: int64[M+1], non-decreasing, ptr[0] >= 0, ptr[M] <= E.  out: int64[E];
 * out[e] = r for ptr[r] <= e < ptr[r+1].  Entries outside [ptr[0], ptr[M])
 * are left untouched (as in the reference). */
int psa_ptr2ind(const int64_t* ptr, int64_t M, int64_t E, int64_t* out,
                psa_stream_t stream);

/* ---- SpMM (CSR x dense), fp32 ------------------------------------------ */

/* out[i,:] = REDUCE_{e in [rowptr[i], rowptr[i+1])} w_e * mat[col[e], :]
 * with w_e = value[e], or 1 when value == NULL.
 *   PSA_SUM : plain sum.
 *   PSA_MEAN: sum / max(deg_i, 1).
 *   PSA_MIN/PSA_MAX: element-wise over k; arg_out[i,k] = e of the first
 *     winner in edge order; empty row -> out = 0, arg_out = nnz (sentinel).
 * Not in the reference tree (README.md:47-50 "Support later"); semantics are
 * upstream pytorch_sparse spmm (README.md:267-306 holds the API and one KAT).
 *
 * rowptr int64[M+1], col int64[nnz] (values in [0, N)), value f32[nnz] or
 * NULL, mat f32[N,K] row-major, out f32[M,K], arg_out int64[M,K] (MIN/MAX;
 * ignored for SUM/MEAN).  arg_out may be NULL for MIN/MAX too: it is then not
 * stored — 2 GB fewer writes at M = 2 M, K = 128; with arg_bytes NULL as well
 * the winners are not even tracked: 2.15 -> 1.60 ms, the sum's time — which is
 * what a caller wants that needs `out` only (inference), or whose backward
 * reads arg_bytes alone (complete when no row has more than 128 entries).
 * nnz is passed explicitly because rowptr lives on the device.
 *
 * arg_bytes: uint8[M,K] or NULL (MIN/MAX only, needs K % 4 == 0): arg_out once
 * more as the winner's index INSIDE its row, one byte per element (index mod
 * 128; bit 7 marks rows of more than 128 edges, where equal bytes name a
 * candidate that the backward checks against arg_out) — the form
 * psa_spmm_minmax_bw_csc reads.  The
 * forward has it in registers, so writing it here (0.26 GB at M = 2 M, K = 128)
 * saves the backward a pass that re-reads all of arg_out (2 GB).  With
 * arg_out == NULL the bytes must come from the kernel itself (K % 4 == 0 and
 * K <= 256); other K tiles return PSA_ERR_INVALID_ARG then.
 *
 * workspace (psa_spmm_workspace_bytes(reduce, K, nnz) bytes, 16-byte aligned)
 * enables the long-row path: rows with more than 128 edges are split into
 * 128-edge chunks reduced by separate wavefronts and folded in chunk order
 * (deterministic).  workspace == NULL keeps every row on one wavefront — same
 * results, but a power-law graph then waits for its longest row. */
size_t psa_spmm_workspace_bytes(int reduce, int64_t K, int64_t nnz);
int psa_spmm(int reduce, const int64_t* rowptr, const int64_t* col,
             const float* value, const float* mat, int64_t M, int64_t N,
             int64_t K, int64_t nnz, float* out, int64_t* arg_out,
             uint8_t* arg_bytes, void* workspace, size_t workspace_bytes,
             psa_stream_t stream);

/* psa_spmm with the COO row ids beside the CSR pointer: row int64[nnz] is what
 * SparseStorage.row() holds (paddle_sparse/storage.py:195-206), row[e] = r for
 * rowptr[r] <= e < rowptr[r+1]; NULL = not at hand (derived from rowptr into
 * the workspace when a kernel wants it — one psa_ptr2ind pass).  The
 * edge-balanced kernels read it: their unit of work is a range of consecutive
 * edges whatever rows they belong to, each lane group keeps the running
 * reduction of the current row in registers and stores a row when the id
 * changes; rows that cross a range boundary are folded in range order by a
 * second launch (deterministic), rows without edges are zero-filled from
 * rowptr by extra workgroups of the same launch.  That removes the per-row
 * rowptr -> col -> gather dependency chain which bounds the one-wave-per-row
 * kernels on power-law graphs.  Same results contract as psa_spmm (sums are
 * taken in edge order inside a range, so sum / mean may differ from the
 * row-wave kernels in the last bits; min / max / arg_out are identical).
 * ldo: floats between consecutive rows of `out` (0 = K): out may be a K-column
 * slice of a wider row-major matrix, so a product computed slice by slice
 * (the feature-sliced multi-GPU exchange) lands in place with no concatenation.
 * arg_out / arg_bytes, when given, stay dense [M, K].
 * hot_rows / num_hot (PSA_SPMM_EDGE_RANGES only; NULL / 0 = off): a compact
 * copy f32[num_hot, K] of the most referenced rows of mat.  Column ids in
 * [N, N + num_hot) then name row id - N of that copy instead of a row of mat.
 * Hub columns of a power-law graph draw a large share of all gathers to a few
 * rows of mat; where those rows sit at addresses that fall on few memory
 * channels (R-MAT as generated: hubs at ids with few bits set) the gather runs
 * at 3.7 TB/s instead of 5; from a compact copy — consecutive addresses, a few
 * MB that stay cache resident — R-MAT scale 21 goes 2.03 -> 1.43 ms.  The caller
 * keeps, per matrix, the list of hot columns and the redirected col array
 * (paddle_sparse_amd/storage.py::_hot_columns), and per call gathers the hot
 * rows of the current mat with psa_gather_rows (13 us for 65 536 rows).
 * arg_width: bytes per entry of arg_bytes, 1 (as psa_spmm) or 2 — (index in the
 * row) & 0xffff, exact up to 65 535 entries per row; see psa_spmm_minmax_bw_csc. */
typedef enum psa_spmm_algo {
  PSA_SPMM_AUTO = 0,        /* today: PSA_SPMM_ROW_WAVES */
  PSA_SPMM_ROW_WAVES = 1,   /* one wavefront per CSR row (+ the long-row chunks) */
  PSA_SPMM_EDGE_RANGES = 2  /* edge-balanced; shapes it does not serve (K % 4 != 0,
                               no workspace, ids beyond 31 bits) fall back to 1 */
} psa_spmm_algo;
int psa_spmm_coo(int reduce, const int64_t* rowptr, const int64_t* row,
                 const int64_t* col, const float* value, const float* mat,
                 const float* hot_rows, int64_t num_hot,
                 int64_t M, int64_t N, int64_t K, int64_t nnz, float* out,
                 int64_t ldo, int64_t* arg_out, void* arg_bytes, int arg_width,
                 int algo, void* workspace, size_t workspace_bytes,
                 psa_stream_t stream);

/* psa_spmm with half-width dense operands: mat and out are fp16 (PSA_F16) or
 * bf16 (PSA_BF16) [N, K] / [M, K] row-major, every product and sum is fp32 and
 * the result is rounded to the 2-byte type once, on store (round to nearest
 * even).  The reference parametrises its tests over float16 / bfloat16 / float32
 * / float64 values (paddle_sparse/testing.py:12-21).  value: fp32[nnz]
 * (value_dtype = PSA_F32), or `dtype`[nnz] (value_dtype = dtype), or NULL.
 * arg_out int64[M, K] or NULL (MIN / MAX).  The gathered rows are half as wide:
 * 2 M x 2 M, 20 M entries, K = 128 moves 6.4 GB instead of 11.5 GB.
 * Needs K % 8 == 0 and 16-byte aligned mat / out (PSA_ERR_UNSUPPORTED
 * otherwise: callers widen to fp32 and use psa_spmm).  One wavefront per row,
 * rows of any length on their own wave (no long-row scratch, no workspace).
 * N IS A CONTRACT, not a hint: it must be the number of rows of mat and bound every
 * column id (0 <= col[e] < N).  When N * K * 2 < 2^32 the kernels address mat with
 * 32-bit byte offsets built from the low 32 bits of col; a stale or short N gives
 * wrong gathers without any error (index values are never validated, as in the
 * reference: csrc/cpu/convert_cpu.cpp:19-27). */
int psa_spmm_half(int reduce, int dtype, const int64_t* rowptr,
                  const int64_t* col, const void* value, int value_dtype,
                  const void* mat, int64_t M, int64_t N, int64_t K, int64_t nnz,
                  void* out, int64_t* arg_out, psa_stream_t stream);

/* psa_spmm_half with the COO row ids, the kernel family and the hub-row copy of
 * psa_spmm_coo: on a power-law graph the one-wave-per-row kernel above waits for
 * its longest row (R-MAT scale 21, bf16, K = 128: 2.3 ms); the edge-range kernels
 * are the same code for 16-byte lanes of 8 two-byte elements.  value: fp32[nnz]
 * or NULL.  hot_rows: [num_hot, K] in mat's type.  algo != PSA_SPMM_EDGE_RANGES,
 * or a shape the edge ranges do not serve (K % 8 != 0, no workspace): forwards to
 * psa_spmm_half (hot_rows must then be NULL).  workspace:
 * psa_spmm_half_workspace_bytes(reduce, K, nnz) bytes, 16-byte aligned.  Sums are
 * taken in edge order inside a range, rows that cross ranges are folded from fp32
 * partials: sum / mean may differ from psa_spmm_half in the last bit of the
 * 2-byte result; min / max / arg_out are identical.  arg_bytes / arg_width (MIN / MAX): the
 * row-local form of arg_out as in psa_spmm_coo ([M, K] entries of 1 or 2 bytes, or NULL) — what the
 * half-width masked pass over the CSC view reads (psa_spmm_half_minmax_bw_csc), so that min / max
 * training on a power-law matrix needs neither an int64 arg_out nor fp32 copies of the operands. */
size_t psa_spmm_half_workspace_bytes(int reduce, int64_t K, int64_t nnz);
int psa_spmm_half_coo(int reduce, int dtype, const int64_t* rowptr,
                      const int64_t* row, const int64_t* col, const float* value,
                      const void* mat, const void* hot_rows, int64_t num_hot,
                      int64_t M, int64_t N, int64_t K, int64_t nnz, void* out,
                      int64_t* arg_out, void* arg_bytes, int arg_width, int algo,
                      void* workspace, size_t workspace_bytes, psa_stream_t stream);

/* sum / mean backward with BOTH gradients in one pass over the CSC view, half-width dense
 * operands (the fp32 form is psa_spmm_sum_bw_csc): mat f16/bf16 [N, K] (the forward's dense
 * operand), grad f16/bf16 [M, K], grad_mat f16/bf16 [N, K] out (fp32 sums, one rounding),
 * grad_value_csc f32[nnz] out in CSC order or NULL.  weight_csc: f32[nnz] = value[csr2csc] (CSC
 * order: psa_permute_apply_u32 / psa_transpose_weights) or NULL (weights 1); row_scale f32[M] or
 * NULL multiplies both gradients per entry (mean: 1 / max(deg(r), 1)).  One wave per column; with a
 * workspace of psa_spmm_half_bw_csc_workspace_bytes(K, nnz) bytes, columns above 128 entries (the hub
 * rows of a power-law matrix) are cut into 128-entry chunks with a wave each, whose fp32 partials are
 * added in chunk order (workspace NULL: every column on its one wave, whatever its length).
 * K % 8 == 0, K <= 512 (the dot <mat[c, :], grad[r, :]> needs the whole row in one tile), else
 * PSA_ERR_UNSUPPORTED.  The dtype list the reference parametrises over: paddle_sparse/testing.py:12-21. */
size_t psa_spmm_half_bw_csc_workspace_bytes(int64_t K, int64_t nnz);

/* psa_spmm_half that also leaves the row-local form of arg_out behind (min / max; arg_bytes: [M, K]
 * entries of arg_width bytes as in psa_spmm_coo, or NULL; arg_out and arg_bytes are both optional). */
int psa_spmm_half_arg(int reduce, int dtype, const int64_t* rowptr, const int64_t* col, const void* value,
                      int value_dtype, const void* mat, int64_t M, int64_t N, int64_t K, int64_t nnz, void* out,
                      int64_t* arg_out, void* arg_bytes, int arg_width, psa_stream_t stream);

/* min / max backward over the CSC view with half-width dense operands, both gradients in one pass (the
 * fp32 form is psa_spmm_minmax_bw_csc in its exact arg_bytes forms): an entry's term counts for column k
 * only where arg_bytes[r, k] equals the entry's tag (psa_csc_edge_tags, same width).  arg_bytes must be
 * exact (one byte: no row above 128 entries; two: none above 65 535).  weight_csc: f32[nnz] value[csr2csc]
 * or NULL; grad_value_csc f32[nnz] in CSC order or NULL; grad_mat in grad's dtype.  K % 8 == 0, K <= 512.
 * workspace: as for psa_spmm_half_sum_bw_csc (long columns in chunks), or NULL. */
int psa_spmm_half_minmax_bw_csc(int dtype, const int64_t* colptr, const int64_t* row_csc, const void* tag,
                                const float* weight_csc, const void* mat, const void* grad, const void* arg_bytes,
                                int arg_width, int64_t M, int64_t N, int64_t K, int64_t nnz, float* grad_value_csc,
                                void* grad_mat, void* workspace, size_t workspace_bytes, psa_stream_t stream);

int psa_spmm_half_sum_bw_csc(int dtype, const int64_t* colptr, const int64_t* row_csc, const float* weight_csc,
                             const float* row_scale, const void* mat, const void* grad, int64_t M, int64_t N,
                             int64_t K, int64_t nnz, float* grad_value_csc, void* grad_mat, void* workspace,
                             size_t workspace_bytes, psa_stream_t stream);

/* Test/bench hook: 0 = default (one row per wave), 1 = several rows per wave for
 * K <= 128, 2 = one row per wave with 8 gather steps in flight, 3 = default
 * without the XCD mixing of the row blocks, 4 = default with 64-bit addressing of the
 * dense operands forced (the form operands of 4 GiB and more take).  Returns the previous value. */
int psa_spmm_half_set_variant(int variant);

/* Row-length statistics of a CSR pointer, for choosing psa_spmm_algo once per
 * matrix (the caller reads the four words back and keeps the answer with the
 * matrix, like the reference keeps rowcount: paddle_sparse/storage.py:373-381).
 * stats int64[4] (device) = { rows without entries, rows with 1 or 2 entries,
 * rows with more than 128 entries, longest row }.  One wave per row is the
 * faster forward while most rows hold a handful of entries or more (uniform
 * 2 M x 2 M, 20 M entries, K = 128: 1.67 ms against 2.02 ms); edge ranges win
 * once empty and 1-2-entry rows dominate (R-MAT degrees, hub ids spread over the
 * address space: 1.51 ms against 1.89 ms). */
int psa_csr_row_stats(const int64_t* rowptr, int64_t M, int64_t* stats,
                      psa_stream_t stream);

/* ---- SpMM backward (fp32) ------------------------------------------------ */

/* grad wrt the sparse values for sum/mean (upstream spmm_value_bw):
 * out[e] = sum_k mat[col[e],k] * grad[row(e),k], divided by max(deg(row(e)),1)
 * for PSA_MEAN.  row(e) is implied by rowptr (one wave per CSR row), so the
 * upstream `row` argument is not needed.  out: f32[nnz].  workspace
 * (psa_spmm_value_bw_workspace_bytes(nnz) bytes, or NULL) holds the work list
 * of the long-row path, as for psa_spmm. */
size_t psa_spmm_value_bw_workspace_bytes(int64_t nnz);
int psa_spmm_value_bw(int reduce, const int64_t* rowptr, const int64_t* col,
                      const float* mat, const float* grad, int64_t M, int64_t K,
                      int64_t nnz, float* out, void* workspace,
                      size_t workspace_bytes, psa_stream_t stream);

/* CSC-ordered edge weights for grad wrt the dense operand (sum/mean):
 * out[j] = (value ? value[csr2csc[j]] : 1) / (mean ? max(deg(row_csc[j]),1) : 1)
 * with row_csc = row[csr2csc].  gB = psa_spmm(PSA_SUM, colptr, row_csc, out,
 * gOut) — upstream torch_sparse/matmul.py spmm_sum/spmm_mean backward. */
int psa_transpose_weights(const float* value, const int64_t* csr2csc,
                          const int64_t* row_csc, const int64_t* rowptr,
                          int64_t nnz, int mean, float* out,
                          psa_stream_t stream);

/* min/max backward through arg_out (entries == nnz are masked):
 *   grad_value[arg] += mat[col[arg],k] * grad[i,k]   (f32[nnz], or NULL)
 *   grad_mat[col[arg],k] += w_arg * grad[i,k]        (f32[N,K], or NULL)
 * Both outputs are zero-filled by the call; accumulation uses float atomics
 * (summation order is not fixed; results agree to fp32 rounding). */
int psa_spmm_minmax_bw(const int64_t* col, const float* value, const float* mat,
                       const float* grad, const int64_t* arg_out, int64_t M,
                       int64_t N, int64_t K, int64_t nnz, float* grad_value,
                       float* grad_mat, psa_stream_t stream);

/* min/max backward WITHOUT atomics, in one pass over the CSC view of the
 * matrix (the caches storage.py:425-434 / tensor.py:254-257 already keep for
 * the sum/mean backward).  For column c and each stored entry e = (r, c):
 *   grad_mat[c, k]  += w_e * grad[r, k]        where arg_out[r, k] == e
 *   grad_value_csc[j] = sum of mat[c, k] * grad[r, k] over those k   (j = CSC
 *                       position of e, i.e. e = csr2csc[j])
 * Scattered float atomics run at ~20-50 G adds/s on this chip whatever their
 * shape (256 M of them at M = 2 M, K = 128: 5 ms); this pass is a gather like
 * the forward, every output element is written once and sums run in CSC edge
 * order (reproducible bit for bit).
 *
 * tag: uint8[nnz] from psa_csc_edge_tags (depends on the sparsity structure
 * only: cache it with csr2csc).  value: f32[nnz] in CSR order, or NULL.
 * arg_bytes: what psa_spmm left behind (see there), or NULL — the call then
 * derives it from arg_out itself in a first pass.  arg_out may be NULL when
 * arg_bytes is given and no row has more than 128 entries (entries of longer
 * rows need the exact test against arg_out; without it they count as no hit).
 * grad_value_csc: f32[nnz] or NULL (then mat may be NULL too); it is written in
 * CSC order, contiguously — psa_gather_rows(grad_value_csc, csc2csr, nnz, 4, ..)
 * puts it into the CSR order the API returns (a 4-byte scatter from inside the
 * pass costs more than that gather).
 * workspace: psa_spmm_minmax_bw_csc_workspace_bytes(M, K, nnz) bytes, 16-byte
 * aligned (arg_out compressed to its row-local form + long-column scratch).
 * Returns PSA_ERR_UNSUPPORTED unless K % 4 == 0 and K <= 256 (callers then
 * use psa_spmm_minmax_bw).
 *
 * arg_width (1 or 2): bytes per entry of arg_bytes and of tag — the width
 * psa_spmm_coo wrote and psa_csc_edge_tags was asked for.  Width 2 holds
 * (index in the row) & 0xffff: exact for rows of up to 65 535 entries (0xffff, like 0xff in the one-byte form, says "no winner": empty row, or no product beat the init), so on a
 * power-law graph (R-MAT scale 21: longest row 41 677) the forward stores
 * 2 bytes per element instead of 8 + 1 and this pass needs no arg_out.
 *
 * hot_grad f32[num_hot, K] / hot_bytes [num_hot, K] entries / num_hot (NULL,
 * NULL, 0 = off): compact copies of the most referenced rows of grad and of
 * arg_bytes.  Ids in [M, M + num_hot) in row_csc then name row id - M of the
 * copies (the hub rows of a power-law graph, as hot_rows of psa_spmm_coo;
 * R-MAT scale 21 as generated: the pass runs 1.9 ms faster with the hubs'
 * rows relabelled away from their crowded addresses).  Needs the two-byte
 * arg_bytes (arg_width 2, arg_out NULL): a matrix whose rows all fit the
 * one-byte form has no hub rows to speak of. */
int psa_csc_edge_tags(const int64_t* rowptr, const int64_t* row_csc,
                      const int64_t* csr2csc, int64_t nnz, void* tag,
                      int width, psa_stream_t stream);
size_t psa_spmm_minmax_bw_csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz);
int psa_spmm_minmax_bw_csc(const int64_t* rowptr, const int64_t* colptr,
                           const int64_t* row_csc, const int64_t* csr2csc,
                           const void* tag, const float* value,
                           const float* mat, const float* grad,
                           const int64_t* arg_out, const void* arg_bytes,
                           int arg_width, const float* hot_grad,
                           const void* hot_bytes, int64_t num_hot,
                           int64_t M, int64_t N,
                           int64_t K, int64_t nnz, float* grad_value_csc,
                           float* grad_mat, void* workspace,
                           size_t workspace_bytes, psa_stream_t stream);

/* Gradient wrt the dense operand of spmm_min / spmm_max for a FIXED adjacency on a
 * power-law matrix: the edge-range kernels of psa_spmm_coo over the CSC view, with
 * the term of an entry counted only where the forward's row-local arg_out names it
 *   grad_mat[c, k] = sum over entries e = (r, c) with arg_bytes[r, k] == tag[e] of
 *                    weight_csc[e] * grad[r, k]
 * (two gathers per entry: the row of grad and its row of arg_bytes).  No
 * grad_value: callers that train the edge values use psa_spmm_minmax_bw_csc.
 * col_csc: int64[nnz], the column of every CSC-ordered entry (the COO row ids of
 * the view), or NULL (derived from colptr into the workspace).  row_csc, tag,
 * arg_bytes / arg_width, hot_grad / hot_bytes / num_hot: as psa_spmm_minmax_bw_csc;
 * arg_bytes must be exact (one byte: no row above 128 entries; two bytes: none above
 * 65 535).  weight_csc: f32[nnz] = value[csr2csc] (psa_transpose_weights) or NULL
 * (weights 1).  Sums run in entry order inside a range and range by range for
 * columns that cross ranges: deterministic, last bits may differ from
 * psa_spmm_minmax_bw_csc.  R-MAT scale 21, K = 128: 2.7 -> 2.0 ms.
 * workspace: psa_spmm_minmax_bw_eb_workspace_bytes(K, nnz) bytes, 16-byte aligned.
 * PSA_ERR_UNSUPPORTED unless K % 4 == 0 and every id fits 31 bits. */
size_t psa_spmm_minmax_bw_eb_workspace_bytes(int64_t K, int64_t nnz);
int psa_spmm_minmax_bw_eb(const int64_t* colptr, const int64_t* col_csc,
                          const int64_t* row_csc, const void* tag,
                          const float* weight_csc, const float* grad,
                          const void* arg_bytes, int arg_width,
                          const float* hot_grad, const void* hot_bytes,
                          int64_t num_hot, int64_t M, int64_t N, int64_t K,
                          int64_t nnz, float* grad_mat, void* workspace,
                          size_t workspace_bytes, psa_stream_t stream);

/* sum backward with BOTH gradients in one pass over the CSC view (trainable
 * edge values): for column c and each stored entry e = (r, c)
 *   grad_mat[c, :] += value[e] * grad[r, :]      (value NULL: weights 1)
 *   grad_value_csc[j] = <mat[c, :], grad[r, :]>   (j = CSC position of e)
 * The gathered grad row serves both, mat[c, :] is the column's own row, and
 * value is read through csr2csc — so the separate psa_spmm_value_bw (a second
 * full gather, of mat rows) and psa_transpose_weights passes are not needed.
 * csr2csc = NULL: `value` is in CSC order already (value[csr2csc], e.g. from
 * psa_permute_apply_u32: 0.12 ms at 20 M entries) and is read as a stream — the
 * nnz dependent 4-byte reads value[csr2csc[j]] were 18 % of this pass's memory
 * requests (profiles/r03_pmc_backward.json).  psa_spmm_minmax_bw_csc takes the
 * same convention when arg_out is NULL (the exact arg_bytes forms).
 * row_scale: f32[M] or NULL; with it both gradients carry the factor
 * row_scale[r] (mean: 1 / max(deg(r), 1), folded in per edge instead of a
 * pre-scaling pass over grad).
 * grad_value_csc: f32[nnz] in CSC order or NULL (CSR order: psa_gather_rows
 * through csc2csr, as above).  hot_grad f32[num_hot, K] / num_hot (NULL / 0 =
 * off): compact copy of the most referenced rows of grad; ids in
 * [M, M + num_hot) in row_csc name its rows (row_scale then has M + num_hot
 * entries, the copies' scales at the end).  M = rows of grad.  workspace:
 * psa_spmm_sum_bw_csc_workspace_bytes(K, nnz) bytes, 16-byte aligned.
 * PSA_ERR_UNSUPPORTED unless K % 4 == 0 and K <= 256. */
size_t psa_spmm_sum_bw_csc_workspace_bytes(int64_t K, int64_t nnz);
int psa_spmm_sum_bw_csc(const int64_t* colptr, const int64_t* row_csc,
                        const int64_t* csr2csc, const float* value,
                        const float* row_scale, const float* mat,
                        const float* grad, const float* hot_grad,
                        int64_t num_hot, int64_t M, int64_t N,
                        int64_t K, int64_t nnz, float* grad_value_csc,
                        float* grad_mat, void* workspace,
                        size_t workspace_bytes, psa_stream_t stream);

/* Test/bench hook: choose the SpMM kernel variant for subsequent psa_spmm
 * calls of this process (0 = auto).  Returns the previous value.  Variants
 * compute identical results up to fp32 summation order; listed in DESIGN.md. */
int psa_spmm_set_variant(int variant);

/* ---- index_sort: stable LSD radix sort (bit-exact permutation) ----------- */

/* Replaces index_sort (paddle_sparse/utils.py:14-23: `inputs.argsort()`),
 * whose call sites are the constructor sort (storage.py:164-169), csr2csc
 * (storage.py:430-432) and csc2csr (storage.py:444-445).
 * keys: int64[n], values in [0, max_value) — max_value is the hint the
 * reference signature already carries (M*N); it selects the number of 8-bit
 * passes.  perm_out: int64[n], the STABLE sorting permutation
 * (== numpy argsort(kind="stable")); sorted_out: int64[n] or NULL.
 * keys is not modified.  workspace: psa_index_sort_workspace_bytes(n,
 * max_value) bytes of device memory, 16-byte aligned.  n < 2^31. */
size_t psa_index_sort_workspace_bytes(int64_t n, int64_t max_value);
int psa_index_sort(const int64_t* keys, int64_t n, int64_t max_value,
                   int64_t* sorted_out, int64_t* perm_out, void* workspace,
                   size_t workspace_bytes, psa_stream_t stream);

/* Diagnostic (synchronises the stream): 0 if the last psa_index_sort /
 * psa_sort_pairs_u32 that used `workspace` (same n, max_value) finished every
 * inter-workgroup wait normally, 1 if a bounded spin of the look-back gave up
 * (never expected; results are then invalid), -1 if the flag could not be read.
 * A sort never hands back a wrong order unmarked (the reference's argsort cannot,
 * storage.py:164-169): once a wait gave up, the last pass stores -1 over its share
 * of perm_out / sorted_out instead of positions.  Callers with no host read behind
 * the sort (the SparseStorage constructor, csr2csc) call this and raise. */
int psa_index_sort_status(const void* workspace, int64_t n, int64_t max_value,
                          psa_stream_t stream);

/* Same sort carrying a caller-defined 4-byte payload per key instead of the
 * permutation: payload_out[i] = payload[perm[i]] (any 4-byte dtype: fp32 /
 * int32 values of a COO matrix).  Lets coalesce (storage.py:164-169 + :471)
 * keep `value[perm]` a sequential stream instead of a random gather.
 * sorted_out is required.  Workspace as psa_index_sort. */
int psa_sort_pairs_u32(const int64_t* keys, const void* payload, int64_t n,
                       int64_t max_value, int64_t* sorted_out, void* payload_out,
                       void* workspace, size_t workspace_bytes, psa_stream_t stream);

/* psa_sort_pairs_u32 on a BIT FIELD of the keys: stable order by
 * (keys[i] >> first_bit), which must be < max_value; the bits below first_bit
 * are not looked at and travel inside the key.  With keys packed as
 * (hi << 32) | lo this orders by `hi` alone in ceil(log2(max_value) / 8) passes
 * while `lo` comes along for free — e.g. products grouped by output column
 * (lo) brought into (row, col) order by ONE sort on the row field.  Workspace:
 * psa_index_sort_workspace_bytes(n, max_value). */
int psa_sort_pairs_u32_field(const int64_t* keys, const void* payload, int64_t n,
                             int first_bit, int64_t max_value,
                             int64_t* sorted_out, void* payload_out,
                             void* workspace, size_t workspace_bytes,
                             psa_stream_t stream);

/* Stable merge of two SORTED key arrays: merged_out[na + nb] is what a stable
 * sort of the concatenation [a; b] would give (a's entry first on equal keys)
 * — the "cat, then argsort" of add.py:30-47, mul.py:57-73 and tensor.py:415-451
 * when both halves are sorted already, in one streaming launch (merge path)
 * instead of ceil(bits / 8) radix passes.  source_out (or NULL) receives the index into
 * the concatenation every merged key came from (i for a[i], na + j for b[j]) —
 * the permutation psa_index_sort would return; payload_out (or NULL; then
 * payload_a / payload_b are ignored) receives the 4-byte payloads in merged
 * order.  Unsorted inputs give an unspecified (memory-safe) order. */
int psa_merge_sorted(const int64_t* a, int64_t na, const int64_t* b, int64_t nb,
                     const void* payload_a, const void* payload_b,
                     int64_t* merged_out, int64_t* source_out, void* payload_out,
                     psa_stream_t stream);

/* Small inputs (n <= psa_coalesce_small_max(), 40960): the sort by (row, col)
 * AND the run-length structure of a coalesce (storage.py:158-171 + 455-470) in
 * one launch of one workgroup — at BASELINE config 1 (10 k entries) the
 * multi-launch chain is pure launch latency.  Outputs are sized for the worst
 * case: out_row / out_col / perm int64[n], ptr int64[n + 1]; on return
 * count_out[0] (device) = number of distinct (row, col), out_row/out_col[0:count]
 * the sorted distinct pairs, ptr[0:count + 1] the run starts in sorted order,
 * perm the stable sorting permutation (same bits as psa_index_sort).
 * workspace: psa_coalesce_small_workspace_bytes(n) bytes, 16-byte aligned. */
int64_t psa_coalesce_small_max(void);
size_t psa_coalesce_small_workspace_bytes(int64_t n);
int psa_coalesce_small(const int64_t* row, const int64_t* col, int64_t n,
                       int64_t M, int64_t N, int64_t* out_row, int64_t* out_col,
                       int64_t* ptr, int64_t* perm, int64_t* count_out,
                       void* workspace, size_t workspace_bytes,
                       psa_stream_t stream);

/* ---- coalesce(index, value, m, n, op) in two calls -------------------------
 * What the reference's user gets from ONE call (paddle_sparse/coalesce.py:25-29
 * = SparseStorage(is_sorted=False) storage.py:158-171 + storage.coalesce
 * storage.py:454-486), with the one host read the dynamic output size needs:
 *
 *   psa_coalesce_count(...)          keys row * N + col, stable sort, run count
 *   host reads status[0] (= count)   <- the only synchronisation
 *   psa_coalesce_write(...)          [2, count] index + reduced values
 *
 * status = the first two int64 words of the workspace: status[0] = number of
 * distinct (row, col) pairs; status[1] = flags, bit 0: some row/col lies outside
 * [0, M) x [0, N) (the reference asserts this, storage.py:78-91: raise, the
 * outputs are unspecified then), bit 1: the input was not sorted by (row, col),
 * bit 2: an inter-workgroup wait of the sort gave up (never expected; the outputs
 * are invalid — the same word psa_index_sort_status reads).
 * psa_coalesce_write takes the count from the device, so a caller may also
 * allocate worst-case outputs (n rows), enqueue both calls back to back and read
 * the status afterwards (count < 0 in the call = "not read yet").
 *
 * value: [n, D] of `dtype` (psa_dtype) or NULL; fp32 / int32 scalars (D = 1) ride
 * through the sort as its payload.  index_out: int64[2 * rows] with rows >= count,
 * filled as the contiguous [2, count] index (row' then col'); value_out [rows, D].
 * n <= 10240 runs as one workgroup whose radix sort stays in the LDS (two
 * launches in all); larger inputs use psa_index_sort / psa_sort_pairs_u32,
 * psa_unique_* and psa_segment_reduce.  transpose(index, value, m, n)
 * (paddle_sparse/transpose.py:41-65) is the same two calls with row/col and M/N
 * swapped.  workspace: psa_coalesce_workspace_bytes(n, M, N) bytes, 16-byte
 * aligned, untouched between the two calls. */
size_t psa_coalesce_workspace_bytes(int64_t n, int64_t M, int64_t N);
int psa_coalesce_count(const int64_t* row, const int64_t* col, const void* value,
                       int dtype, int64_t D, int64_t n, int64_t M, int64_t N,
                       void* workspace, size_t workspace_bytes,
                       psa_stream_t stream);
int psa_coalesce_write(const void* value, int dtype, int64_t D, int64_t n,
                       int64_t M, int64_t N, int reduce, int64_t count,
                       const void* workspace, int64_t* index_out,
                       void* value_out, psa_stream_t stream);

/* The whole coalesce in ONE launch, for n <= psa_coalesce_small_max_fused() (10 240)
 * entries with no values or fp32 / int32 scalar values (BASELINE config 1): the
 * one-workgroup LDS sort of the chain above also writes the contiguous [2, count]
 * index into index_out (int64[2 n] capacity) and reduces every run of values into
 * value_out ([n] capacity).  status int64[2] (device) = {count, flags} as above;
 * the caller reads it once after the launch.  No workspace. */
int psa_coalesce_small_max_fused(void);
int psa_coalesce_small_fused(const int64_t* row, const int64_t* col,
                             const void* value, int dtype, int64_t n, int64_t M,
                             int64_t N, int reduce, int64_t* index_out,
                             void* value_out, int64_t* status,
                             psa_stream_t stream);

/* psa_make_keys for (row, col) with the reference's range assertions folded in:
 * keys[i] = row[i] * N + col[i]; status int64[4] (device): status[1] gets bit 0
 * when some row/col lies outside [0, M) x [0, N), bit 1 when keys[i] < keys[i-1]
 * for some i (storage.py:163); status[0] is zeroed, status[2..3] are scratch. */
int psa_make_keys_checked(const int64_t* row, const int64_t* col, int64_t n,
                          int64_t M, int64_t N, int64_t* keys, int64_t* status,
                          psa_stream_t stream);

/* ---- a 4-byte array through a FIXED permutation, planned ---------------------- */

/* dst[i] = src[perm[i]] for 4-byte elements, n < 2^31, as two streaming passes over a plan
 * built once per permutation (csrc/permute.hip) instead of n dependent 4-byte reads: the
 * value[csr2csc] gathers of tensor.py:254-257 / transpose.py:19-22 and the way of grad_value
 * from CSC back to CSR order.  T = psa_permute_tile() (32 768).
 * Plan (structure only, 8 bytes per element), with dest = the INVERSE of perm (dest[s] = where
 * source s goes), tile(s) = s / T, block(s) = dest[s] / T:
 *   perm_ts  = stable sorting permutation of the keys tile(s) * ceil(n / T) + block(s)
 *   perm_mid = stable sorting permutation of block(s);  gslot = its inverse
 *   psa_permute_plan_pack(perm_ts, gslot, perm_mid, dest, n, sl, gs, lo):
 *     sl: uint16[n], gs: int32[n], lo: uint16[n]
 * psa_permute_apply_u32: mid = scratch of n 4-byte elements; src, mid, dst distinct. */
int64_t psa_permute_tile(void);
int psa_permute_plan_pack(const int64_t* perm_ts, const int64_t* gslot, const int64_t* perm_mid,
                          const int64_t* dest, int64_t n, void* sl, void* gs, void* lo, psa_stream_t stream);
int psa_permute_apply_u32(const void* src, const void* sl, const void* gs, const void* lo, int64_t n, void* mid,
                          void* dst, psa_stream_t stream);

/* Test/bench hook: scatter kernel variant of psa_index_sort for this process
 * (0 = production: single-sweep passes with decoupled look-back, 512 threads x
 * 16 keys; 5 = the same with 1024 x 8; 1-4, 7 = histogram / scan / scatter
 * per pass with 2048-key LDS tile, direct per-lane stores, 4096- / 8192-key
 * tiles).  Returns the previous value.  All variants produce the same bits. */
int psa_sort_set_variant(int variant);

/* Test hook: polls a look-back wait of the single-sweep passes may spend on a word that is
 * not ready before it gives up and raises the fault word (production: 2^22; 0 makes every
 * wait that does not succeed at once give up; negative restores the default).  Returns the
 * previous value. */
int64_t psa_sort_set_spin_limit(int64_t limit);

/* keys[i] = a[i] * mul + b[i]  (storage.py:159-162 key = row*N + col,
 * storage.py:430 key = M*col + row).  If unsorted_flag != NULL, *unsorted_flag
 * (device int32, caller-zeroed) is set to 1 when some keys[i] < keys[i-1]
 * (the reference's sortedness test, storage.py:163). */
int psa_make_keys(const int64_t* a, const int64_t* b, int64_t mul, int64_t n,
                  int64_t* keys, int32_t* unsorted_flag, psa_stream_t stream);

/* The inverse of psa_make_keys on a (sorted) key stream: hi[i] = keys[i] / div,
 * lo[i] = keys[i] % div; either output may be NULL.  Replaces the row[perm] /
 * col[perm] gathers after a sort (storage.py:166-168, tensor.py:254-257,
 * transpose.py:14-16) with one sequential pass: the sorted keys already hold
 * both indices. */
int psa_split_keys(const int64_t* keys, int64_t n, int64_t div, int64_t* hi,
                   int64_t* lo, psa_stream_t stream);

/* out[i, :] = src[perm[i], :] for rows of row_bytes bytes (any dtype):
 * the `x[perm]` gathers of storage.py:166-169, transpose.py:14-22,
 * tensor.py:252-257. */
int psa_gather_rows(const void* src, const int64_t* perm, int64_t n,
                    int64_t row_bytes, void* out, psa_stream_t stream);

/* psa_gather_rows on a column window of the source rows: out[i, :] = the
 * width_bytes bytes at offset_bytes of row perm[i] of src, whose rows are
 * src_row_bytes apart; out is dense [n, width_bytes].  Packs the rows (and the
 * feature slice) of the dense operand another rank asked for into its send
 * buffer (row-partitioned SpMM, halo exchange: paddle_sparse_amd/distributed.py). */
int psa_gather_rows_window(const void* src, int64_t src_row_bytes,
                           int64_t offset_bytes, int64_t width_bytes,
                           const int64_t* perm, int64_t n, void* out,
                           psa_stream_t stream);

/* inv[perm[i]] = i.  Same result as the reference's second sort in csc2csr
 * (storage.py:444-445 sorts a permutation to invert it), in one pass. */
int psa_invert_permutation(const int64_t* perm, int64_t n, int64_t* inv,
                           psa_stream_t stream);

/* out[b] = #{i : index[i] == b} for b in [0, size): colcount
 * (storage.py:414-418, scatter_add of ones).  out: int64[size]. */
int psa_bincount(const int64_t* index, int64_t n, int64_t size, int64_t* out,
                 psa_stream_t stream);

/* ptr_out[0] = 0, ptr_out[i+1] = counts[0] + ... + counts[i]: colptr from
 * colcount (storage.py:397-398, zeros + cumsum).  ptr_out: int64[n+1]. */
size_t psa_count2ptr_workspace_bytes(int64_t n);
int psa_count2ptr(const int64_t* counts, int64_t n, int64_t* ptr_out,
                  void* workspace, size_t workspace_bytes, psa_stream_t stream);

/* ---- coalesce on sorted keys (storage.py:454-486) ------------------------ */

/* Phase 1: *count_out (device int64) = number of distinct values in
 * sorted_keys[n] (the reference's mask.sum(); mask.all() <=> count == n).
 * workspace: psa_unique_workspace_bytes(n) bytes; it carries per-block
 * offsets into phase 2 and must stay untouched in between. */
size_t psa_unique_workspace_bytes(int64_t n);
int psa_unique_count(const int64_t* sorted_keys, int64_t n, void* workspace,
                     size_t workspace_bytes, int64_t* count_out,
                     psa_stream_t stream);
/* psa_unique_count on the output of psa_index_sort / psa_sort_pairs_u32, folding in
 * the sort's look-back diagnostic: count_out int64[2] (device) = {count, fault},
 * fault != 0 when a bounded inter-workgroup wait of a radix pass gave up (never
 * expected; the order is then invalid).  sort_workspace / sort_max_value: what the
 * sort was called with (its workspace must still be alive); n is the same n.
 * The caller reads both words in its one host read of the count. */
int psa_unique_count_after_sort(const int64_t* sorted_keys, int64_t n,
                                void* workspace, size_t workspace_bytes,
                                const void* sort_workspace, int64_t sort_max_value,
                                int64_t* count_out, psa_stream_t stream);

/* Phase 2 (after the caller has read the count and sized the outputs):
 * ptr_out int64[count+1] = start of every run of equal keys, ptr_out[count]
 * = n (the reference's `ptr`, storage.py:467-470); row_out/col_out
 * int64[count] = key / N, key % N of each distinct key (the reference's
 * row[mask], col[mask], storage.py:462-463).  Any of ptr_out or the
 * (row_out, col_out) pair may be NULL.  count: the device scalar written by
 * phase 1. */
int psa_unique_write(const int64_t* sorted_keys, int64_t n, int64_t N,
                     const void* workspace, const int64_t* count,
                     int64_t* ptr_out, int64_t* row_out, int64_t* col_out,
                     psa_stream_t stream);

/* psa_unique_write (packed index, no ptr) and psa_segment_reduce in ONE launch, for 4-byte values
 * (PSA_F32 / PSA_I32) that are in SORTED order already — they rode the sort as its payload
 * (psa_sort_pairs_u32) or the input was sorted: index_out int64[2 * count] = the distinct rows, then the
 * distinct columns (the [2, nnz'] index of coalesce.py:29); value_out[count] = the reduction of every run
 * of equal keys, taken sequentially in run order (psa_segment_reduce's order and bits).  The thread that
 * writes a pair reduces its run: meant for inputs whose runs are short (count * 32 > n); a heavily
 * duplicated input is better served by psa_unique_write + psa_segment_reduce, whose reducer then takes a
 * wave per run.  Other dtypes: PSA_ERR_UNSUPPORTED. */
int psa_unique_write_reduce(int reduce, int dtype, const int64_t* sorted_keys, int64_t n, int64_t N,
                            const void* workspace, const int64_t* count, int64_t* index_out,
                            const void* payload, void* value_out, psa_stream_t stream);

/* out[s, :] = REDUCE_{i in [ptr[s], ptr[s+1])} src[perm ? perm[i] : i, :]
 * for src rows of D elements of `dtype` (psa_dtype).  Stands in for
 * paddle_scatter.segment_csr at storage.py:471 (with perm = the sort
 * permutation, so value[perm] is never materialised), reduce.py:51 and
 * tensor.py:437.  Semantics = pytorch_scatter segment_csr: empty segment ->
 * 0, mean = sum / count (floor division for integer dtypes), min/max values
 * only.  Sums run in segment order.  n_hint = ptr[nseg] if known (selects the
 * wave-per-segment kernel for long segments), else 0. */
int psa_segment_reduce(int reduce, int dtype, const void* src,
                       const int64_t* perm, const int64_t* ptr, int64_t nseg,
                       int64_t D, int64_t n_hint, void* out,
                       psa_stream_t stream);

/* out[index[i], :] = REDUCE over i of src[i, :]; src [n, D] of `dtype`
 * (f32/f64/i32/i64), out [dim_size, D].  Stands in for
 * paddle_scatter.scatter at reduce.py:40-42 (reduce over dim 0, index = col).
 * Rows no index points at are 0; mean = sum / count (floor for integers).
 * workspace (psa_scatter_workspace_bytes(dim_size) bytes) is needed for
 * mean/min/max.  Float sums use atomics: order not fixed. */
size_t psa_scatter_workspace_bytes(int64_t dim_size);
int psa_scatter_reduce(int reduce, int dtype, const void* src,
                       const int64_t* index, int64_t n, int64_t D,
                       int64_t dim_size, void* out, void* workspace,
                       size_t workspace_bytes, psa_stream_t stream);

/* ---- spspmm: C = A @ B with both operands sparse (README.md:308-353) -------
 * The reference documents `spspmm(indexA, valueA, indexB, valueB, m, k, n)` but
 * ships no kernel for it.  Here it is expand / sort / compress on the path's
 * own kernels: psa_spspmm_count -> psa_count2ptr (product offsets) ->
 * psa_ptr2ind (owner A entry of every product) -> psa_spspmm_expand ->
 * psa_sort_pairs_u32 | psa_index_sort -> psa_unique_* -> psa_segment_reduce.
 *
 * counts[e] = entries stored in B's row colA[e] (A entries in storage order). */
int psa_spspmm_count(const int64_t* colA, int64_t nnzA, const int64_t* rowptrB,
                     int64_t* counts, psa_stream_t stream);

/* For product p in [0, total): e = owner[p], q = rowptrB[colA[e]] + p -
 * offsets[e]; keys[p] = rowA[e] * n + colB[q]; vals[p] = valA[e] * valB[q]
 * (a NULL value array counts as ones; vals may be NULL).  With n < 0 the key
 * is packed instead: keys[p] = (colB[q] << 32) | rowA[e] — the form used when
 * the walk runs over the CSC views of both operands (outer = entries of B by
 * column, inner = a column of A), so that products come out grouped by output
 * column and ONE stable sort on the row field (psa_sort_pairs_u32_field, half
 * the passes of a sort on row * n + col) finishes the order.  dtype: f32, f64,
 * i32 or i64.  Products of one C entry keep the order in which a sequential
 * row-by-row product meets them, so a stable sort + psa_segment_reduce's
 * in-order kernel (runs averaging < 32 products) reproduces that sum bit for
 * bit. */
int psa_spspmm_expand(int dtype, const int64_t* rowA, const int64_t* colA,
                      const void* valA, const int64_t* rowptrB, const int64_t* colB,
                      const void* valB, const int64_t* offsets, const int64_t* owner,
                      int64_t total, int64_t n, int64_t* keys, void* vals,
                      psa_stream_t stream);

/* ---- sample_adj on the GPU ------------------------------------------------
 * Replaces sample_adj (csrc/sample.cpp:8-24, which throws for GPU tensors;
 * CPU text csrc/cpu/sample_cpu.cpp:9-148).  The op is a chain, because its
 * output sizes are data dependent (two host reads: E and the new-node count):
 *
 *   psa_sample_count  -> psa_count2ptr (out_rowptr; E = out_rowptr[S])
 *   psa_ptr2ind(out_rowptr) -> owner;  psa_sample_select -> e_raw[E]
 *   psa_relabel_mark -> flags;  psa_count2ptr(flags) -> rank; new = rank[E]
 *   psa_relabel_finish -> n_id[S + new], keys[E] = i * n_out + new column id
 *   psa_index_sort(keys) -> out_col = keys % n_out, out_e_id = e_raw[perm]
 *
 * Conventions where the CPU text leaves the order open: picks of one row are
 * taken in draw order (sample_cpu.cpp:108 iterates an unordered_set), and
 * equal new column ids inside a row keep that order (the reference's std::sort
 * at :134-140 is not stable).  num_neighbors < 0 (no sampling) involves
 * neither, apart from multi-edges.
 *
 * Random draws are counter based: draw t of subset row i is a function of
 * (seed, i, t) alone (oracle/sample_oracle.c restates it), in place of the
 * reference's framework-global generator.
 *
 * counts[i] = picks of subset row i: deg if num_neighbors < 0; num_neighbors if
 * replace and deg > 0; min(deg, num_neighbors) otherwise. */
int psa_sample_count(const int64_t* rowptr, const int64_t* idx, int64_t S,
                     int64_t num_neighbors, int replace, int64_t* counts,
                     psa_stream_t stream);

/* e_raw[p] = edge picked by slot p (p = out_rowptr[i] + t, i = owner[p]). */
int psa_sample_select(const int64_t* rowptr, const int64_t* idx, int64_t S,
                      const int64_t* out_rowptr, const int64_t* owner, int64_t E,
                      int64_t num_neighbors, int replace, uint64_t seed,
                      int64_t* e_raw, psa_stream_t stream);

/* newid: int64[num_nodes] scratch (filled by the call).  Afterwards newid[c] =
 * position of c in idx (the last one for duplicates, as the CPU map does) for
 * subset nodes, -2 - (first slot that picked c) for other picked nodes.
 * flags[p] = 1 when slot p is that first pick, else 0 (int64[E]). */
int psa_relabel_mark(const int64_t* idx, int64_t S, const int64_t* col,
                     const int64_t* e_raw, int64_t E, int64_t num_nodes,
                     int64_t* newid, int64_t* flags, psa_stream_t stream);

/* rank = exclusive scan of flags (psa_count2ptr); n_out = S + rank[E].
 * n_id[0:S] = idx, n_id[S + rank[p]] = node first picked at slot p;
 * keys[p] = owner[p] * n_out + (new id of slot p's node). */
int psa_relabel_finish(const int64_t* idx, int64_t S, const int64_t* col,
                       const int64_t* e_raw, int64_t E, const int64_t* newid,
                       const int64_t* rank, const int64_t* owner, int64_t n_out,
                       int64_t* n_id, int64_t* keys, psa_stream_t stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* PADDLE_SPARSE_HIP_H_ */
