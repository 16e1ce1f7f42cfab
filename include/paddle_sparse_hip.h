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
 * NULL, mat f32[N,K] row-major, out f32[M,K], arg_out int64[M,K] (required
 * for MIN/MAX, ignored otherwise; may be NULL for SUM/MEAN).
 * nnz is passed explicitly because rowptr lives on the device. */
int psa_spmm(int reduce, const int64_t* rowptr, const int64_t* col,
             const float* value, const float* mat, int64_t M, int64_t N,
             int64_t K, int64_t nnz, float* out, int64_t* arg_out,
             psa_stream_t stream);

/* Test/bench hook: choose the SpMM kernel variant for subsequent psa_spmm
 * calls of this process (0 = auto).  Returns the previous value.  Variants
 * compute identical results up to fp32 summation order; listed in DESIGN.md. */
int psa_spmm_set_variant(int variant);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* PADDLE_SPARSE_HIP_H_ */
