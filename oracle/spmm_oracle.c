/*
 * TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.
 *
 * SpMM (CSR x dense) forward and backward on the CPU.
 *
 * The reference tree has NO SpMM (/root/reference/README.md:47-50 lists
 * spmm/matmul under "Support later"); it declares upstream
 * rusty1s/pytorch_sparse as its base (README.md:8) and documents the API and
 * one known answer (README.md:267-306).  This file restates the published
 * upstream algorithm (pytorch_sparse csrc/cpu/spmm_cpu.cpp + reducer.h,
 * not vendored in /root/reference):
 *   for each row i, for each edge e in [rowptr[i], rowptr[i+1]) IN ORDER,
 *   for each k: update(acc[k], w_e * mat[col[e], k]) ; then write(acc).
 *   init : sum/mean 0, min +FLT_MAX, max -FLT_MAX (numeric_limits lowest)
 *   update: sum/mean acc += x ; min if (x < acc) {acc = x; arg = e} ; max ">"
 *   write: sum acc ; mean acc / max(deg, 1) ; min/max deg > 0 ? acc : 0,
 *          arg stays at the sentinel nnz for empty rows.
 * Pinned by README.md:293-305 (tests/golden/reference_kats.json) and
 * cross-checked against torch-CPU torch.sparse.mm(csr, dense, reduce) in
 * tests/test_oracle.py; beyond those: "parity unpinned" (no reference tests).
 *
 * fp32 arithmetic in the upstream order; products and sums are kept as
 * separate roundings (-ffp-contract=off in the build recipe).
 */
#include <float.h>
#include <stdint.h>
#include <string.h>

enum { ORACLE_SUM = 0, ORACLE_MEAN = 1, ORACLE_MIN = 2, ORACLE_MAX = 3 };

static void spmm_rows(int reduce, const int64_t* rowptr, const int64_t* col,
                      const float* value, const float* mat, int64_t r_begin,
                      int64_t r_end, int64_t K, int64_t nnz, float* out,
                      int64_t* arg_out) {
  for (int64_t i = r_begin; i < r_end; ++i) {
    const int64_t s = rowptr[i], e_end = rowptr[i + 1];
    float* o = out + i * K;
    int64_t* a = arg_out ? arg_out + i * K : 0;
    for (int64_t k = 0; k < K; ++k) {
      o[k] = reduce == ORACLE_MIN ? FLT_MAX
                                  : (reduce == ORACLE_MAX ? -FLT_MAX : 0.f);
      if (a) a[k] = nnz;
    }
    for (int64_t e = s; e < e_end; ++e) {
      const float w = value ? value[e] : 1.f;
      const float* b = mat + col[e] * K;
      if (reduce == ORACLE_SUM || reduce == ORACLE_MEAN) {
        for (int64_t k = 0; k < K; ++k) o[k] += w * b[k];
      } else if (reduce == ORACLE_MIN) {
        for (int64_t k = 0; k < K; ++k) {
          const float x = w * b[k];
          if (x < o[k]) {
            o[k] = x;
            a[k] = e;
          }
        }
      } else {
        for (int64_t k = 0; k < K; ++k) {
          const float x = w * b[k];
          if (x > o[k]) {
            o[k] = x;
            a[k] = e;
          }
        }
      }
    }
    const int64_t deg = e_end - s;
    if (reduce == ORACLE_MEAN) {
      const float d = (float)(deg > 0 ? deg : 1);
      for (int64_t k = 0; k < K; ++k) o[k] = o[k] / d;
    } else if ((reduce == ORACLE_MIN || reduce == ORACLE_MAX) && deg == 0) {
      for (int64_t k = 0; k < K; ++k) o[k] = 0.f;
    }
  }
}

/* Single-threaded, upstream order. arg_out may be NULL for sum/mean. */
void oracle_spmm(int reduce, const int64_t* rowptr, const int64_t* col,
                 const float* value, const float* mat, int64_t M, int64_t K,
                 int64_t nnz, float* out, int64_t* arg_out) {
  spmm_rows(reduce, rowptr, col, value, mat, 0, M, K, nnz, out, arg_out);
}

/* Same arithmetic, rows split over OpenMP threads (rows are independent, so
 * results are identical to oracle_spmm).  Used as bench.py's cpu_baseline. */
void oracle_spmm_omp(int reduce, const int64_t* rowptr, const int64_t* col,
                     const float* value, const float* mat, int64_t M,
                     int64_t K, int64_t nnz, float* out, int64_t* arg_out) {
  /* chunks of 4 x 256 rows: fine enough to balance, coarse enough to amortise */
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t blk = 0; blk < (M + 255) / 256; ++blk) {
    const int64_t r0 = blk * 256;
    const int64_t r1 = r0 + 256 < M ? r0 + 256 : M;
    spmm_rows(reduce, rowptr, col, value, mat, r0, r1, K, nnz, out, arg_out);
  }
}

/* ---- backward --------------------------------------------------------- */

/* upstream spmm_value_bw_cpu: gV[e] = sum_k mat[col[e],k] * grad[row[e],k],
 * divided by max(deg(row[e]), 1) for mean.  reduce in {SUM, MEAN}. */
void oracle_spmm_value_bw(int reduce, const int64_t* row,
                          const int64_t* rowptr, const int64_t* col,
                          const float* mat, const float* grad, int64_t nnz,
                          int64_t K, float* out) {
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t r = row[e];
    const float* b = mat + col[e] * K;
    const float* g = grad + r * K;
    float v = 0.f;
    for (int64_t k = 0; k < K; ++k) v += b[k] * g[k];
    if (reduce == ORACLE_MEAN) {
      int64_t deg = rowptr[r + 1] - rowptr[r];
      v = v / (float)(deg > 0 ? deg : 1);
    }
    out[e] = v;
  }
}

/* grad wrt the dense operand for sum/mean, computed the direct way
 * (gB[col[e],:] += w_e * gOut[row[e],:] in edge order; mean scales w_e by
 * 1/deg(row[e])).  Upstream evaluates the same quantity as an spmm_sum over
 * the CSC view (torch_sparse/matmul.py spmm_sum backward); the summation
 * order differs, hence tolerance not bit equality in the tests. */
void oracle_spmm_mat_bw(int reduce, const int64_t* row, const int64_t* rowptr,
                        const int64_t* col, const float* value,
                        const float* grad, int64_t nnz, int64_t N, int64_t K,
                        float* out) {
  memset(out, 0, sizeof(float) * (size_t)(N * K));
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t r = row[e];
    float w = value ? value[e] : 1.f;
    if (reduce == ORACLE_MEAN) {
      int64_t deg = rowptr[r + 1] - rowptr[r];
      w = w / (float)(deg > 0 ? deg : 1);
    }
    float* o = out + col[e] * K;
    const float* g = grad + r * K;
    for (int64_t k = 0; k < K; ++k) o[k] += w * g[k];
  }
}

/* min/max backward through arg_out (upstream torch_sparse/matmul.py
 * spmm_min/spmm_max backward): entries with arg == nnz are masked.
 *   gV[arg[i,k]]        += mat[col[arg[i,k]], k] * gOut[i,k]
 *   gB[col[arg[i,k]],k] += w_arg * gOut[i,k]
 * grad_value / grad_mat may each be NULL. */
void oracle_spmm_minmax_bw(const int64_t* col, const float* value,
                           const float* mat, const float* grad,
                           const int64_t* arg_out, int64_t M, int64_t N,
                           int64_t K, int64_t nnz, float* grad_value,
                           float* grad_mat) {
  if (grad_value) memset(grad_value, 0, sizeof(float) * (size_t)nnz);
  if (grad_mat) memset(grad_mat, 0, sizeof(float) * (size_t)(N * K));
  for (int64_t i = 0; i < M; ++i) {
    for (int64_t k = 0; k < K; ++k) {
      const int64_t e = arg_out[i * K + k];
      if (e == nnz) continue;
      const float g = grad[i * K + k];
      const int64_t c = col[e];
      if (grad_value) grad_value[e] += mat[c * K + k] * g;
      if (grad_mat) grad_mat[c * K + k] += (value ? value[e] : 1.f) * g;
    }
  }
}

/* Condition number helper for the fp32 tolerance: S[i,k] = sum_e |w_e *
 * mat[col[e],k]| in double.  Tests bound |gpu - oracle| by 1e-5 * S. */
void oracle_spmm_abs_sum(const int64_t* rowptr, const int64_t* col,
                         const float* value, const float* mat, int64_t M,
                         int64_t K, double* out) {
  for (int64_t i = 0; i < M; ++i) {
    double* o = out + i * K;
    for (int64_t k = 0; k < K; ++k) o[k] = 0.0;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
      const double w = value ? value[e] : 1.0;
      const float* b = mat + col[e] * K;
      for (int64_t k = 0; k < K; ++k) {
        const double x = w * (double)b[k];
        o[k] += x < 0 ? -x : x;
      }
    }
  }
}
