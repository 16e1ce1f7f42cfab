// A client of the C-ABI with no framework in the process: plain HIP runtime +
// libpaddle_sparse_hip.so, as the Paddle custom-op shim would use it.  Built and
// run by tests/test_cabi_client_gpu.py.  Exit code 0 = every check passed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "paddle_sparse_hip.h"

#define HIP_OK(x)                                                          \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                            \
    }                                                                      \
  } while (0)
#define PSA_OK_(x)                                                         \
  do {                                                                     \
    if ((x) != PSA_OK) {                                                   \
      std::printf("psa error: %s (%s:%d)\n", psa_last_error(), __FILE__, __LINE__); \
      return 3;                                                            \
    }                                                                      \
  } while (0)

template <typename T>
T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * (h.empty() ? 1 : h.size())) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  if (psa_abi_version() != PSA_ABI_VERSION || psa_sparse_cuda_version() != -1) return 1;
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  // README.md:293-305: index [[0,0,1,2,2],[0,2,1,0,1]], value [1,2,4,1,3], B [[1,4],[2,5],[3,6]]
  const int64_t M = 3, N = 3, K = 2, nnz = 5;
  std::vector<int64_t> row = {0, 0, 1, 2, 2}, col = {0, 2, 1, 0, 1};
  std::vector<float> val = {1, 2, 4, 1, 3}, B = {1, 4, 2, 5, 3, 6};
  int64_t *d_row = to_device(row), *d_col = to_device(col);
  float *d_val = to_device(val), *d_B = to_device(B);
  int64_t* d_rowptr;
  float* d_out;
  HIP_OK(hipMalloc(&d_rowptr, sizeof(int64_t) * (M + 1)));
  HIP_OK(hipMalloc(&d_out, sizeof(float) * M * K));
  PSA_OK_(psa_ind2ptr(d_row, nnz, M, d_rowptr, stream));
  PSA_OK_(psa_spmm(PSA_SUM, d_rowptr, d_col, d_val, d_B, M, N, K, nnz, d_out, nullptr, nullptr, nullptr, 0, stream));
  std::vector<int64_t> rowptr(M + 1);
  std::vector<float> out(M * K);
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(rowptr.data(), d_rowptr, sizeof(int64_t) * (M + 1), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out.data(), d_out, sizeof(float) * M * K, hipMemcpyDeviceToHost));
  const int64_t want_ptr[] = {0, 2, 3, 5};
  const float want_out[] = {7, 16, 8, 20, 7, 19};
  for (int i = 0; i <= M; ++i)
    if (rowptr[i] != want_ptr[i]) return std::printf("rowptr[%d] = %lld\n", i, (long long)rowptr[i]), 4;
  for (int i = 0; i < M * K; ++i)
    if (out[i] != want_out[i]) return std::printf("out[%d] = %f\n", i, out[i]), 5;

  // test/test_coalesce.py:7-34 through make_keys -> index_sort -> unique -> segment_reduce
  std::vector<int64_t> crow = {1, 0, 1, 0, 2, 1}, ccol = {0, 1, 1, 1, 0, 0};
  std::vector<float> cval = {1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7};  // [6, 2]
  const int64_t cn = 6, cm = 3, cN = 2;
  int64_t *d_crow = to_device(crow), *d_ccol = to_device(ccol), *d_keys, *d_sorted, *d_perm, *d_count;
  float* d_cval = to_device(cval);
  HIP_OK(hipMalloc(&d_keys, 8 * cn));
  HIP_OK(hipMalloc(&d_sorted, 8 * cn));
  HIP_OK(hipMalloc(&d_perm, 8 * cn));
  HIP_OK(hipMalloc(&d_count, 8));
  void *sort_ws, *uniq_ws;
  const size_t sort_b = psa_index_sort_workspace_bytes(cn, cm * cN), uniq_b = psa_unique_workspace_bytes(cn);
  HIP_OK(hipMalloc(&sort_ws, sort_b));
  HIP_OK(hipMalloc(&uniq_ws, uniq_b));
  PSA_OK_(psa_make_keys(d_crow, d_ccol, cN, cn, d_keys, nullptr, stream));
  PSA_OK_(psa_index_sort(d_keys, cn, cm * cN, d_sorted, d_perm, sort_ws, sort_b, stream));
  PSA_OK_(psa_unique_count(d_sorted, cn, uniq_ws, uniq_b, d_count, stream));
  int64_t count = 0;
  HIP_OK(hipMemcpyAsync(&count, d_count, 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  if (count != 4) return std::printf("count = %lld\n", (long long)count), 6;
  int64_t *d_ptr, *d_orow, *d_ocol;
  float* d_oval;
  HIP_OK(hipMalloc(&d_ptr, 8 * (count + 1)));
  HIP_OK(hipMalloc(&d_orow, 8 * count));
  HIP_OK(hipMalloc(&d_ocol, 8 * count));
  HIP_OK(hipMalloc(&d_oval, 4 * count * 2));
  PSA_OK_(psa_unique_write(d_sorted, cn, cN, uniq_ws, d_count, d_ptr, d_orow, d_ocol, stream));
  PSA_OK_(psa_segment_reduce(PSA_SUM, PSA_F32, d_cval, d_perm, d_ptr, count, 2, cn, d_oval, stream));
  std::vector<int64_t> orow(count), ocol(count);
  std::vector<float> oval(count * 2);
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(orow.data(), d_orow, 8 * count, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ocol.data(), d_ocol, 8 * count, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(oval.data(), d_oval, 4 * count * 2, hipMemcpyDeviceToHost));
  const int64_t want_r[] = {0, 1, 1, 2}, want_c[] = {1, 0, 1, 0};
  const float want_v[] = {6, 8, 7, 9, 3, 4, 5, 6};
  for (int i = 0; i < 4; ++i)
    if (orow[i] != want_r[i] || ocol[i] != want_c[i]) return std::printf("coalesced index %d wrong\n", i), 7;
  for (int i = 0; i < 8; ++i)
    if (oval[i] != want_v[i]) return std::printf("coalesced value[%d] = %f\n", i, oval[i]), 8;

  // the same known answer through the two-call chain (psa_coalesce_count / _write), both protocols:
  // (a) worst-case outputs, both calls enqueued, status read afterwards; (b) count read in between
  {
    const size_t cb = psa_coalesce_workspace_bytes(cn, cm, cN);
    void* cws;
    HIP_OK(hipMalloc(&cws, cb));
    for (int protocol = 0; protocol < 2; ++protocol) {
      int64_t status[2] = {-1, -1};
      PSA_OK_(psa_coalesce_count(d_crow, d_ccol, d_cval, PSA_F32, 2, cn, cm, cN, cws, cb, stream));
      int64_t rows = cn, known = -1;
      if (protocol == 1) {
        HIP_OK(hipMemcpyAsync(status, cws, 16, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        rows = known = status[0];
      }
      int64_t* d_index;
      float* d_cv;
      HIP_OK(hipMalloc(&d_index, 8 * 2 * rows));
      HIP_OK(hipMalloc(&d_cv, 4 * 2 * rows));
      PSA_OK_(psa_coalesce_write(d_cval, PSA_F32, 2, cn, cm, cN, PSA_SUM, known, cws, d_index, d_cv, stream));
      HIP_OK(hipMemcpyAsync(status, cws, 16, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      if (status[0] != 4 || status[1] != 2) return std::printf("chain status {%lld, %lld}\n", (long long)status[0], (long long)status[1]), 11;
      std::vector<int64_t> index(8);
      std::vector<float> cv(8);
      HIP_OK(hipMemcpy(index.data(), d_index, 8 * 8, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(cv.data(), d_cv, 4 * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < 4; ++i)
        if (index[i] != want_r[i] || index[4 + i] != want_c[i]) return std::printf("chain index %d wrong\n", i), 12;
      for (int i = 0; i < 8; ++i)
        if (cv[i] != want_v[i]) return std::printf("chain value[%d] = %f\n", i, cv[i]), 13;
    }
    // an index outside the matrix comes back as flag bit 0, not as a fault
    std::vector<int64_t> bad_col = {0, 1, 1, 1, 0, 2};
    int64_t* d_bad = to_device(bad_col);
    int64_t status[2];
    PSA_OK_(psa_coalesce_count(d_crow, d_bad, nullptr, 0, 0, cn, cm, cN, cws, cb, stream));
    HIP_OK(hipMemcpyAsync(status, cws, 16, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (!(status[1] & 1)) return 14;
  }

  // the one-launch form (scalar fp32 values): first column of the values above
  {
    std::vector<float> sval = {1, 2, 3, 4, 5, 6};
    float* d_sval = to_device(sval);
    int64_t *d_index, *d_status;
    float* d_sv;
    HIP_OK(hipMalloc(&d_index, 8 * 2 * cn));
    HIP_OK(hipMalloc(&d_sv, 4 * cn));
    HIP_OK(hipMalloc(&d_status, 16));
    if (cn > psa_coalesce_small_max_fused()) return 15;
    PSA_OK_(psa_coalesce_small_fused(d_crow, d_ccol, d_sval, PSA_F32, cn, cm, cN, PSA_SUM, d_index, d_sv, d_status, stream));
    int64_t status[2];
    std::vector<int64_t> index(8);
    std::vector<float> sv(4);
    HIP_OK(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(index.data(), d_index, 8 * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(sv.data(), d_sv, 4 * 4, hipMemcpyDeviceToHost));
    const float want_s[] = {6, 7, 3, 5};  // (0,1): 2 + 4, (1,0): 1 + 6, (1,1): 3, (2,0): 5
    if (status[0] != 4 || status[1] != 2) return 16;
    for (int i = 0; i < 4; ++i)
      if (index[i] != want_r[i] || index[4 + i] != want_c[i] || sv[i] != want_s[i]) return std::printf("fused entry %d wrong\n", i), 17;
  }

  // spmm_max forward leaving the two-byte row-local arg_out, then both gradients in one pass over
  // the CSC view (README.md:293-305 matrix; K = 4 so that the 16-byte kernels apply)
  {
    const int64_t K4 = 4;
    std::vector<float> B4 = {1, 4, 0, 0, 2, 5, 0, 0, 3, 6, 0, 0};  // [3, 4]: the README's B padded with zeros
    std::vector<float> G4(M * K4, 1.0f);
    float *d_B4 = to_device(B4), *d_G4 = to_device(G4), *d_out4, *d_gv_csc, *d_gm;
    int64_t *d_colptr, *d_perm2, *d_rowcsc, *d_keys2, *d_sorted2;
    uint16_t *d_words, *d_tags;
    HIP_OK(hipMalloc(&d_out4, 4 * M * K4));
    HIP_OK(hipMalloc(&d_words, 2 * M * K4));
    HIP_OK(hipMalloc(&d_tags, 2 * nnz));
    HIP_OK(hipMalloc(&d_gv_csc, 4 * nnz));
    HIP_OK(hipMalloc(&d_gm, 4 * N * K4));
    HIP_OK(hipMalloc(&d_colptr, 8 * (N + 1)));
    HIP_OK(hipMalloc(&d_perm2, 8 * nnz));
    HIP_OK(hipMalloc(&d_rowcsc, 8 * nnz));
    HIP_OK(hipMalloc(&d_keys2, 8 * nnz));
    HIP_OK(hipMalloc(&d_sorted2, 8 * nnz));
    const size_t fw_b = psa_spmm_workspace_bytes(PSA_MAX, K4, nnz);
    void* fw_ws = nullptr;
    if (fw_b) HIP_OK(hipMalloc(&fw_ws, fw_b));
    PSA_OK_(psa_spmm_coo(PSA_MAX, d_rowptr, d_row, d_col, d_val, d_B4, nullptr, 0, M, N, K4, nnz, d_out4, 0, nullptr, d_words, 2,
                         PSA_SPMM_AUTO, fw_ws, fw_b, stream));
    // CSC view: stable sort of col (csr2csc), row[csr2csc], colptr
    void* s_ws;
    const size_t s_b = psa_index_sort_workspace_bytes(nnz, N);
    HIP_OK(hipMalloc(&s_ws, s_b));
    PSA_OK_(psa_index_sort(d_col, nnz, N, d_sorted2, d_perm2, s_ws, s_b, stream));
    PSA_OK_(psa_gather_rows(d_row, d_perm2, nnz, 8, d_rowcsc, stream));
    PSA_OK_(psa_ind2ptr(d_sorted2, nnz, N, d_colptr, stream));
    PSA_OK_(psa_csc_edge_tags(d_rowptr, d_rowcsc, d_perm2, nnz, d_tags, 2, stream));
    const size_t bw_b = psa_spmm_minmax_bw_csc_workspace_bytes(M, K4, nnz);
    void* bw_ws;
    HIP_OK(hipMalloc(&bw_ws, bw_b));
    PSA_OK_(psa_spmm_minmax_bw_csc(d_rowptr, d_colptr, d_rowcsc, d_perm2, d_tags, d_val, d_B4, d_G4, nullptr, d_words, 2, nullptr,
                                   nullptr, 0, M, N, K4, nnz, d_gv_csc, d_gm, bw_ws, bw_b, stream));
    std::vector<float> out4(M * K4), gm(N * K4), gv(nnz);
    std::vector<int64_t> perm2(nnz);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out4.data(), d_out4, 4 * M * K4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gm.data(), d_gm, 4 * N * K4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gv.data(), d_gv_csc, 4 * nnz, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(perm2.data(), d_perm2, 8 * nnz, hipMemcpyDeviceToHost));
    // A = [[1, 0, 2], [0, 4, 0], [1, 3, 0]], entries e0..e4 in CSR order: max over each row of value * B[col]
    const float want_out4[] = {6, 12, 0, 0, 8, 20, 0, 0, 6, 15, 0, 0};
    for (int i = 0; i < M * K4; ++i)
      if (out4[i] != want_out4[i]) return std::printf("max out[%d] = %f\n", i, out4[i]), 18;
    // winners: row 0 -> e1 for k 0, 1 and e0 for the zero columns (ties go to the first entry); row 1 -> e2;
    // row 2 -> e4 for k 0, 1 and e3 for k 2, 3.  grad_mat[c, k] = sum of the values of the winners in column c
    const float want_gm[] = {0, 0, 2, 2, 7, 7, 4, 4, 2, 2, 0, 0};
    for (int i = 0; i < N * K4; ++i)
      if (gm[i] != want_gm[i]) return std::printf("grad_mat[%d] = %f\n", i, gm[i]), 19;
    // grad_value[e] = sum over the k it won of B[col[e], k] (grad = 1), read back through csr2csc
    const float want_gv[] = {0, 9, 7, 0, 7};
    for (int j = 0; j < nnz; ++j)
      if (gv[j] != want_gv[perm2[j]]) return std::printf("grad_value of entry %lld = %f\n", (long long)perm2[j], gv[j]), 20;
  }

  // error reporting: a bad enum comes back as a status + message, not a crash
  if (psa_spmm(17, d_rowptr, d_col, d_val, d_B, M, N, K, nnz, d_out, nullptr, nullptr, nullptr, 0, stream) != PSA_ERR_INVALID_ARG)
    return 9;
  if (psa_last_error()[0] == '\0') return 10;
  std::printf("C-ABI client: ind2ptr, spmm (README KAT), coalesce as primitives and as the two-call chain (test_coalesce KAT), error path OK\n");
  return 0;
}
