// Paddle custom-op shim over the C-ABI HIP core (include/paddle_sparse_hip.h).
//
// NOT compiled in this repository's image (no <paddle/extension.h> on disk);
// it is the binding a paddle_sparse maintainer adds so that
// `import paddle_sparse_ops` resolves to the MI355X kernels.  It keeps the
// reference's registration conventions (csrc/convert.cpp:13-43,46-76,
// csrc/version.cpp:14-40): one PD_BUILD_OP per op, kernels take
// paddle::Tensor& and attrs by value, outputs come from paddle::empty, work is
// enqueued on tensor.stream() and not synchronised, errors become PD_THROW.
// There is no logic here beyond argument marshalling.
#include <paddle/extension.h>

#include <vector>

#include "paddle_sparse_hip.h"

#define PSA_CALL(expr)                                  \
  do {                                                  \
    if ((expr) != PSA_OK) PD_THROW(psa_last_error());   \
  } while (0)

// A sort never hands back a wrong order (the reference's argsort cannot, storage.py:164-169):
// the look-back diagnostic of the radix passes, one 4-byte host read behind the sort.
#define PSA_SORT_OK(ws_ptr, n, max_value, stream)                                                  \
  PD_CHECK(psa_index_sort_status((ws_ptr), (n), (max_value), (stream)) == 0,                       \
           "index_sort: an inter-workgroup wait of a radix pass gave up; the order is invalid")

#define CHECK_GPU(x) PD_CHECK((x).is_gpu(), #x " must be a GPU tensor (HIP build has no CPU path)")
#define CHECK_I64(x) PD_CHECK((x).dtype() == paddle::DataType::INT64, #x " must be int64")

namespace {
inline void* stream_of(const paddle::Tensor& t) { return reinterpret_cast<void*>(t.stream()); }
inline const int64_t* i64(const paddle::Tensor& t) { return t.data<int64_t>(); }
inline const float* f32(const paddle::Tensor& t) { return t.data<float>(); }
inline const int64_t* i64_or_null(const paddle::optional<paddle::Tensor>& t) { return t ? t.get().data<int64_t>() : nullptr; }
inline const float* f32_or_null(const paddle::optional<paddle::Tensor>& t) { return t ? t.get().data<float>() : nullptr; }
inline paddle::Tensor i64_empty(int64_t n, const paddle::Place& place) { return paddle::empty({n}, paddle::DataType::INT64, place); }
inline paddle::Tensor scratch(size_t bytes, const paddle::Place& place) {
  return paddle::empty({static_cast<int64_t>(bytes > 0 ? bytes : 1)}, paddle::DataType::UINT8, place);
}
inline int dtype_id_of(const paddle::Tensor& t) {
  switch (t.dtype()) {
    case paddle::DataType::FLOAT32: return PSA_F32;
    case paddle::DataType::FLOAT64: return PSA_F64;
    case paddle::DataType::INT32: return PSA_I32;
    case paddle::DataType::INT64: return PSA_I64;
    case paddle::DataType::FLOAT16: return PSA_F16;
    case paddle::DataType::BFLOAT16: return PSA_BF16;
    default: PD_THROW("unsupported value dtype");
  }
  return -1;
}
inline int64_t row_elems(const paddle::Tensor& t) {  // elements per entry along dim 0
  int64_t d = 1;
  const auto shape = t.shape();
  for (size_t i = 1; i < shape.size(); ++i) d *= shape[i];
  return d;
}
// one small device -> host read (counts that size dynamic outputs)
inline std::vector<int64_t> read_i64(const paddle::Tensor& t, int64_t first, int64_t n) {
  auto host = paddle::experimental::slice(t, {0}, {first}, {first + n}, {}, {}).copy_to(paddle::CPUPlace(), true);
  return std::vector<int64_t>(host.data<int64_t>(), host.data<int64_t>() + n);
}
}  // namespace

// ---- sparse_cuda_version (csrc/version.cpp:14-40) ---------------------------
std::vector<paddle::Tensor> sparse_cuda_version() {
  return {paddle::full({1}, psa_sparse_cuda_version(), paddle::DataType::INT64, paddle::CPUPlace())};
}
std::vector<paddle::DataType> sparse_cuda_version_infer_dtype() { return {paddle::DataType::INT64}; }
std::vector<std::vector<int64_t>> sparse_cuda_version_infer_shape() { return {{1}}; }
PD_BUILD_OP(sparse_cuda_version)
    .Inputs({})
    .Outputs({"out"})
    .SetKernelFn(PD_KERNEL(sparse_cuda_version))
    .SetInferShapeFn(PD_INFER_SHAPE(sparse_cuda_version_infer_shape))
    .SetInferDtypeFn(PD_INFER_DTYPE(sparse_cuda_version_infer_dtype));

// ---- ind2ptr / ptr2ind (csrc/convert.cpp:13-76) --------------------------------
std::vector<paddle::Tensor> ind2ptr(paddle::Tensor& ind, int64_t M) {
  CHECK_GPU(ind);
  CHECK_I64(ind);
  auto out = paddle::empty({M + 1}, ind.dtype(), ind.place());
  PSA_CALL(psa_ind2ptr(i64(ind), ind.numel(), M, out.data<int64_t>(), stream_of(ind)));
  return {out};
}
std::vector<paddle::DataType> ind2ptr_infer_dtype(const paddle::DataType d) { return {d}; }
std::vector<std::vector<int64_t>> ind2ptr_infer_shape(int64_t M) { return {{M + 1}}; }
PD_BUILD_OP(ind2ptr)
    .Inputs({"ind"})
    .Outputs({"out"})
    .Attrs({"M: int64_t"})
    .SetKernelFn(PD_KERNEL(ind2ptr))
    .SetInferShapeFn(PD_INFER_SHAPE(ind2ptr_infer_shape))
    .SetInferDtypeFn(PD_INFER_DTYPE(ind2ptr_infer_dtype));

std::vector<paddle::Tensor> ptr2ind(paddle::Tensor& ptr, int64_t E) {
  CHECK_GPU(ptr);
  CHECK_I64(ptr);
  auto out = paddle::empty({E}, ptr.dtype(), ptr.place());
  PSA_CALL(psa_ptr2ind(i64(ptr), ptr.numel() - 1, E, out.data<int64_t>(), stream_of(ptr)));
  return {out};
}
std::vector<paddle::DataType> ptr2ind_infer_dtype(const paddle::DataType d) { return {d}; }
std::vector<std::vector<int64_t>> ptr2ind_infer_shape(int64_t E) { return {{E}}; }
PD_BUILD_OP(ptr2ind)
    .Inputs({"ptr"})
    .Outputs({"out"})
    .Attrs({"E: int64_t"})
    .SetKernelFn(PD_KERNEL(ptr2ind))
    .SetInferShapeFn(PD_INFER_SHAPE(ptr2ind_infer_shape))
    .SetInferDtypeFn(PD_INFER_DTYPE(ptr2ind_infer_dtype));

// ---- index_sort (seam: paddle_sparse/utils.py:14-23) ------------------------------
// returns (sorted, perm); utils.index_sort drops `sorted` unless asked for.
std::vector<paddle::Tensor> index_sort(paddle::Tensor& keys, int64_t max_value) {
  CHECK_GPU(keys);
  CHECK_I64(keys);
  const int64_t n = keys.numel();
  auto sorted = paddle::empty({n}, keys.dtype(), keys.place());
  auto perm = paddle::empty({n}, keys.dtype(), keys.place());
  const int64_t ws_bytes = static_cast<int64_t>(psa_index_sort_workspace_bytes(n, max_value));
  auto ws = paddle::empty({ws_bytes > 0 ? ws_bytes : 1}, paddle::DataType::UINT8, keys.place());
  PSA_CALL(psa_index_sort(i64(keys), n, max_value, sorted.data<int64_t>(), perm.data<int64_t>(),
                          ws.data<uint8_t>(), static_cast<size_t>(ws_bytes), stream_of(keys)));
  if (n > 0) PSA_SORT_OK(ws.data<uint8_t>(), n, max_value, stream_of(keys));
  return {sorted, perm};
}
std::vector<paddle::DataType> index_sort_infer_dtype(const paddle::DataType d) { return {d, d}; }
PD_BUILD_OP(index_sort)
    .Inputs({"keys"})
    .Outputs({"sorted", "perm"})
    .Attrs({"max_value: int64_t"})
    .SetKernelFn(PD_KERNEL(index_sort))
    .SetInferDtypeFn(PD_INFER_DTYPE(index_sort_infer_dtype));

// ---- spmm_{sum,mean,min,max}: new ops, same convention -------------------------------
// `value` and `row` are optional; Paddle's paddle::optional<Tensor> input
// (anticipated by PD_DISPATCH_HAS_VALUE, csrc/cpu/utils.h:11-20) maps to a NULL
// pointer in the C-ABI.  `row` = SparseStorage.row() when it is cached; `algo` =
// psa_spmm_algo, chosen once per matrix from csr_row_stats below; `hot_rows` = the
// compact copy of the hub rows of mat that `col` (then the redirected array) points into.
static std::vector<paddle::Tensor> spmm_impl(int reduce, paddle::Tensor& rowptr, paddle::Tensor& col,
                                             const paddle::optional<paddle::Tensor>& value,
                                             const paddle::optional<paddle::Tensor>& row, paddle::Tensor& mat,
                                             const paddle::optional<paddle::Tensor>& hot_rows, int64_t algo,
                                             bool want_arg, int64_t arg_width) {
  // arg_width: 0 = no row-local form of arg_out; 1 = one byte per element (exact for rows of up to
  // 128 entries); 2 = two bytes (exact up to 65 536: power-law graphs) -- see psa_spmm_minmax_bw_csc
  PD_CHECK(arg_width >= 0 && arg_width <= 2, "arg_width must be 0, 1 or 2");
  const bool want_arg_bytes = arg_width > 0;
  CHECK_GPU(mat);
  CHECK_I64(rowptr);
  CHECK_I64(col);
  PD_CHECK(mat.dtype() == paddle::DataType::FLOAT32, "spmm is fp32 in the HIP build");
  const int64_t M = rowptr.numel() - 1, N = mat.shape()[0], K = mat.shape()[1], nnz = col.numel();
  auto out = paddle::empty({M, K}, mat.dtype(), mat.place());
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  auto arg = minmax && want_arg ? paddle::empty({M, K}, paddle::DataType::INT64, mat.place())
                                : paddle::empty({0}, paddle::DataType::INT64, mat.place());
  const auto local_dtype = arg_width == 2 ? paddle::DataType::INT16 : paddle::DataType::UINT8;
  auto arg_bytes = minmax && want_arg_bytes ? paddle::empty({M, K}, local_dtype, mat.place())
                                            : paddle::empty({0}, local_dtype, mat.place());
  const size_t ws_bytes = psa_spmm_workspace_bytes(reduce, K, nnz);
  auto ws = scratch(ws_bytes, mat.place());
  PSA_CALL(psa_spmm_coo(reduce, i64(rowptr), i64_or_null(row), i64(col), f32_or_null(value), f32(mat),
                        f32_or_null(hot_rows), hot_rows ? hot_rows.get().shape()[0] : 0, M, N, K, nnz,
                        out.data<float>(), /*ldo=*/0, minmax && want_arg ? arg.data<int64_t>() : nullptr,
                        minmax && want_arg_bytes ? arg_bytes.data() : nullptr, arg_width == 2 ? 2 : 1,
                        static_cast<int>(algo),
                        ws_bytes > 0 ? ws.data<uint8_t>() : nullptr, ws_bytes, stream_of(mat)));
  return {out, arg, arg_bytes};
}
#define PSA_SPMM_OP(NAME, RED)                                                                \
  std::vector<paddle::Tensor> NAME(paddle::Tensor& rowptr, paddle::Tensor& col,                \
                                   const paddle::optional<paddle::Tensor>& value,             \
                                   const paddle::optional<paddle::Tensor>& row,               \
                                   paddle::Tensor& mat,                                       \
                                   const paddle::optional<paddle::Tensor>& hot_rows,          \
                                   int64_t algo, bool want_arg, int64_t arg_width) {          \
    return spmm_impl(RED, rowptr, col, value, row, mat, hot_rows, algo, want_arg,             \
                     arg_width);                                                              \
  }                                                                                           \
  PD_BUILD_OP(NAME)                                                                           \
      .Inputs({"rowptr", "col", paddle::Optional("value"), paddle::Optional("row"), "mat",     \
               paddle::Optional("hot_rows")})                                                 \
      .Outputs({"out", "arg_out", "arg_bytes"})                                               \
      .Attrs({"algo: int64_t", "want_arg: bool", "arg_width: int64_t"})                       \
      .SetKernelFn(PD_KERNEL(NAME));
PSA_SPMM_OP(spmm_sum, PSA_SUM)
PSA_SPMM_OP(spmm_mean, PSA_MEAN)
PSA_SPMM_OP(spmm_min, PSA_MIN)
PSA_SPMM_OP(spmm_max, PSA_MAX)

// spmm with fp16 / bf16 dense operands (fp32 sums): `value` fp32 or mat's dtype or absent
std::vector<paddle::Tensor> spmm_half(paddle::Tensor& rowptr, paddle::Tensor& col,
                                      const paddle::optional<paddle::Tensor>& value, paddle::Tensor& mat,
                                      int64_t reduce, bool want_arg) {
  CHECK_GPU(mat);
  CHECK_I64(rowptr);
  CHECK_I64(col);
  const int64_t M = rowptr.numel() - 1, N = mat.shape()[0], K = mat.shape()[1], nnz = col.numel();
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  auto out = paddle::empty({M, K}, mat.dtype(), mat.place());
  auto arg = minmax && want_arg ? paddle::empty({M, K}, paddle::DataType::INT64, mat.place())
                                : paddle::empty({0}, paddle::DataType::INT64, mat.place());
  PSA_CALL(psa_spmm_half(static_cast<int>(reduce), dtype_id_of(mat), i64(rowptr), i64(col),
                         value ? value.get().data() : nullptr, value ? dtype_id_of(value.get()) : 0, mat.data(), M, N, K,
                         nnz, out.data(), minmax && want_arg ? arg.data<int64_t>() : nullptr, stream_of(mat)));
  return {out, arg};
}
PD_BUILD_OP(spmm_half)
    .Inputs({"rowptr", "col", paddle::Optional("value"), "mat"})
    .Outputs({"out", "arg_out"})
    .Attrs({"reduce: int64_t", "want_arg: bool"})
    .SetKernelFn(PD_KERNEL(spmm_half));

// the same with the COO row ids, the kernel family and the hub-row copy (power-law graphs); fp32 or no values
std::vector<paddle::Tensor> spmm_half_coo(paddle::Tensor& rowptr, paddle::Tensor& col,
                                          const paddle::optional<paddle::Tensor>& value,
                                          const paddle::optional<paddle::Tensor>& row, paddle::Tensor& mat,
                                          const paddle::optional<paddle::Tensor>& hot_rows, int64_t reduce, int64_t algo,
                                          bool want_arg) {
  CHECK_GPU(mat);
  CHECK_I64(rowptr);
  CHECK_I64(col);
  const int64_t M = rowptr.numel() - 1, N = mat.shape()[0], K = mat.shape()[1], nnz = col.numel();
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  auto out = paddle::empty({M, K}, mat.dtype(), mat.place());
  auto arg = minmax && want_arg ? paddle::empty({M, K}, paddle::DataType::INT64, mat.place())
                                : paddle::empty({0}, paddle::DataType::INT64, mat.place());
  const size_t ws_bytes = psa_spmm_half_workspace_bytes(static_cast<int>(reduce), K, nnz);
  auto ws = scratch(ws_bytes, mat.place());
  PSA_CALL(psa_spmm_half_coo(static_cast<int>(reduce), dtype_id_of(mat), i64(rowptr), i64_or_null(row), i64(col),
                             f32_or_null(value), mat.data(), hot_rows ? hot_rows.get().data() : nullptr,
                             hot_rows ? hot_rows.get().shape()[0] : 0, M, N, K, nnz, out.data(),
                             minmax && want_arg ? arg.data<int64_t>() : nullptr, nullptr, 1, static_cast<int>(algo),
                             ws_bytes > 0 ? ws.data<uint8_t>() : nullptr, ws_bytes, stream_of(mat)));
  return {out, arg};
}
PD_BUILD_OP(spmm_half_coo)
    .Inputs({"rowptr", "col", paddle::Optional("value"), paddle::Optional("row"), "mat", paddle::Optional("hot_rows")})
    .Outputs({"out", "arg_out"})
    .Attrs({"reduce: int64_t", "algo: int64_t", "want_arg: bool"})
    .SetKernelFn(PD_KERNEL(spmm_half_coo));

// {rows without entries, rows of 1-2 entries, rows above 128 entries, longest row}: read once per
// matrix by the Python layer to pick `algo` (paddle_sparse_amd/storage.py::_spmm_algo).
std::vector<paddle::Tensor> csr_row_stats(paddle::Tensor& rowptr) {
  CHECK_GPU(rowptr);
  CHECK_I64(rowptr);
  auto stats = i64_empty(4, rowptr.place());
  PSA_CALL(psa_csr_row_stats(i64(rowptr), rowptr.numel() - 1, stats.data<int64_t>(), stream_of(rowptr)));
  return {stats};
}
PD_BUILD_OP(csr_row_stats).Inputs({"rowptr"}).Outputs({"stats"}).SetKernelFn(PD_KERNEL(csr_row_stats));

// grad wrt the values (upstream spmm_value_bw); grad wrt mat is spmm_sum over
// the CSC view and is composed in Python (paddle_sparse/matmul.py PyLayer),
// exactly as paddle_sparse_amd/matmul.py does on torch.
std::vector<paddle::Tensor> spmm_value_bw(paddle::Tensor& rowptr, paddle::Tensor& col, paddle::Tensor& mat,
                                          paddle::Tensor& grad, bool mean) {
  CHECK_GPU(mat);
  const int64_t M = rowptr.numel() - 1, K = mat.shape()[1], nnz = col.numel();
  auto out = paddle::empty({nnz}, mat.dtype(), mat.place());
  const int64_t ws_bytes = static_cast<int64_t>(psa_spmm_value_bw_workspace_bytes(nnz));
  auto ws = paddle::empty({ws_bytes > 0 ? ws_bytes : 1}, paddle::DataType::UINT8, mat.place());
  PSA_CALL(psa_spmm_value_bw(mean ? PSA_MEAN : PSA_SUM, i64(rowptr), i64(col), f32(mat), f32(grad), M, K,
                             nnz, out.data<float>(), ws_bytes > 0 ? ws.data<uint8_t>() : nullptr,
                             static_cast<size_t>(ws_bytes), stream_of(mat)));
  return {out};
}
PD_BUILD_OP(spmm_value_bw)
    .Inputs({"rowptr", "col", "mat", "grad"})
    .Outputs({"out"})
    .Attrs({"mean: bool"})
    .SetKernelFn(PD_KERNEL(spmm_value_bw));

// CSC-ordered edge weights of grad_mat = A^T grad_out (upstream torch_sparse/matmul.py backward)
std::vector<paddle::Tensor> transpose_weights(const paddle::optional<paddle::Tensor>& value, paddle::Tensor& csr2csc,
                                              const paddle::optional<paddle::Tensor>& row_csc,
                                              const paddle::optional<paddle::Tensor>& rowptr, bool mean) {
  CHECK_GPU(csr2csc);
  CHECK_I64(csr2csc);
  const int64_t nnz = csr2csc.numel();
  auto out = paddle::empty({nnz}, paddle::DataType::FLOAT32, csr2csc.place());
  PSA_CALL(psa_transpose_weights(f32_or_null(value), i64(csr2csc), i64_or_null(row_csc), i64_or_null(rowptr), nnz,
                                 mean ? 1 : 0, out.data<float>(), stream_of(csr2csc)));
  return {out};
}
PD_BUILD_OP(transpose_weights)
    .Inputs({paddle::Optional("value"), "csr2csc", paddle::Optional("row_csc"), paddle::Optional("rowptr")})
    .Outputs({"out"})
    .Attrs({"mean: bool"})
    .SetKernelFn(PD_KERNEL(transpose_weights));

// min/max backward through arg_out with float atomics (any K)
std::vector<paddle::Tensor> spmm_minmax_bw(paddle::Tensor& col, const paddle::optional<paddle::Tensor>& value,
                                           paddle::Tensor& mat, paddle::Tensor& grad, paddle::Tensor& arg_out,
                                           bool want_value, bool want_mat) {
  CHECK_GPU(mat);
  CHECK_I64(col);
  CHECK_I64(arg_out);
  const int64_t N = mat.shape()[0], K = mat.shape()[1], M = grad.shape()[0], nnz = col.numel();
  auto gv = paddle::empty({want_value ? nnz : 0}, mat.dtype(), mat.place());
  auto gm = want_mat ? paddle::empty({N, K}, mat.dtype(), mat.place()) : paddle::empty({0}, mat.dtype(), mat.place());
  PSA_CALL(psa_spmm_minmax_bw(i64(col), f32_or_null(value), f32(mat), f32(grad), i64(arg_out), M, N, K, nnz,
                              want_value ? gv.data<float>() : nullptr, want_mat ? gm.data<float>() : nullptr,
                              stream_of(mat)));
  return {gv, gm};
}
PD_BUILD_OP(spmm_minmax_bw)
    .Inputs({"col", paddle::Optional("value"), "mat", "grad", "arg_out"})
    .Outputs({"grad_value", "grad_mat"})
    .Attrs({"want_value: bool", "want_mat: bool"})
    .SetKernelFn(PD_KERNEL(spmm_minmax_bw));

// position of every CSC-ordered edge inside its CSR row (structure only: cache it with csr2csc)
std::vector<paddle::Tensor> csc_edge_tags(paddle::Tensor& rowptr, paddle::Tensor& row_csc, paddle::Tensor& csr2csc,
                                          int64_t width) {
  PD_CHECK(width == 1 || width == 2, "width must be 1 or 2");
  CHECK_GPU(csr2csc);
  CHECK_I64(rowptr);
  CHECK_I64(row_csc);
  CHECK_I64(csr2csc);
  const int64_t nnz = csr2csc.numel();
  auto tag = paddle::empty({nnz}, width == 2 ? paddle::DataType::INT16 : paddle::DataType::UINT8, csr2csc.place());
  PSA_CALL(psa_csc_edge_tags(i64(rowptr), i64(row_csc), i64(csr2csc), nnz, tag.data(), static_cast<int>(width),
                             stream_of(csr2csc)));
  return {tag};
}
PD_BUILD_OP(csc_edge_tags)
    .Inputs({"rowptr", "row_csc", "csr2csc"})
    .Outputs({"tag"})
    .Attrs({"width: int64_t"})
    .SetKernelFn(PD_KERNEL(csc_edge_tags));

// min/max backward, both gradients in one pass over the CSC view, no atomics; grad_value comes back
// in CSR order (the pass writes CSC order, one gather through csc2csr follows)
std::vector<paddle::Tensor> spmm_minmax_bw_csc(paddle::Tensor& rowptr, paddle::Tensor& colptr, paddle::Tensor& row_csc,
                                               paddle::Tensor& csr2csc, paddle::Tensor& csc2csr, paddle::Tensor& tag,
                                               const paddle::optional<paddle::Tensor>& value, paddle::Tensor& mat,
                                               paddle::Tensor& grad, const paddle::optional<paddle::Tensor>& arg_out,
                                               const paddle::optional<paddle::Tensor>& arg_bytes,
                                               const paddle::optional<paddle::Tensor>& hot_grad,
                                               const paddle::optional<paddle::Tensor>& hot_bytes, bool want_value) {
  // tag / arg_bytes: UINT8 (one byte per entry) or INT16 (two); hot_grad / hot_bytes: compact copies of
  // the rows of grad / arg_bytes that row_csc names as M + position (psa_gather_rows by the caller)
  CHECK_GPU(grad);
  const int width = tag.dtype() == paddle::DataType::INT16 ? 2 : 1;
  const int64_t M = grad.shape()[0], K = grad.shape()[1], N = colptr.numel() - 1, nnz = csr2csc.numel();
  const auto place = grad.place();
  auto gv_csc = paddle::empty({want_value ? nnz : 0}, grad.dtype(), place);
  auto gv = paddle::empty({want_value ? nnz : 0}, grad.dtype(), place);
  auto gm = paddle::empty({N, K}, grad.dtype(), place);
  const size_t ws_bytes = psa_spmm_minmax_bw_csc_workspace_bytes(M, K, nnz);
  auto ws = scratch(ws_bytes, place);
  PSA_CALL(psa_spmm_minmax_bw_csc(i64(rowptr), i64(colptr), i64(row_csc), i64(csr2csc), tag.data(),
                                  f32_or_null(value), want_value ? f32(mat) : nullptr, f32(grad), i64_or_null(arg_out),
                                  arg_bytes ? arg_bytes.get().data() : nullptr, width, f32_or_null(hot_grad),
                                  hot_bytes ? hot_bytes.get().data() : nullptr,
                                  hot_grad ? hot_grad.get().shape()[0] : 0, M, N, K, nnz,
                                  want_value ? gv_csc.data<float>() : nullptr, gm.data<float>(), ws.data<uint8_t>(),
                                  ws_bytes, stream_of(grad)));
  if (want_value) PSA_CALL(psa_gather_rows(gv_csc.data<float>(), i64(csc2csr), nnz, 4, gv.data<float>(), stream_of(grad)));
  return {gv, gm};
}
PD_BUILD_OP(spmm_minmax_bw_csc)
    .Inputs({"rowptr", "colptr", "row_csc", "csr2csc", "csc2csr", "tag", paddle::Optional("value"), "mat", "grad",
             paddle::Optional("arg_out"), paddle::Optional("arg_bytes"), paddle::Optional("hot_grad"),
             paddle::Optional("hot_bytes")})
    .Outputs({"grad_value", "grad_mat"})
    .Attrs({"want_value: bool"})
    .SetKernelFn(PD_KERNEL(spmm_minmax_bw_csc));

// min/max backward wrt the dense operand for a fixed adjacency on a power-law matrix: edge ranges over the CSC view
std::vector<paddle::Tensor> spmm_minmax_bw_eb(paddle::Tensor& colptr, const paddle::optional<paddle::Tensor>& col_csc,
                                              paddle::Tensor& row_csc, paddle::Tensor& tag,
                                              const paddle::optional<paddle::Tensor>& weight_csc, paddle::Tensor& grad,
                                              paddle::Tensor& arg_bytes, const paddle::optional<paddle::Tensor>& hot_grad,
                                              const paddle::optional<paddle::Tensor>& hot_bytes) {
  CHECK_GPU(grad);
  const int64_t M = grad.shape()[0], K = grad.shape()[1], N = colptr.numel() - 1, nnz = row_csc.numel();
  const int width = tag.dtype() == paddle::DataType::INT16 ? 2 : 1;
  auto gm = paddle::empty({N, K}, grad.dtype(), grad.place());
  const size_t ws_bytes = psa_spmm_minmax_bw_eb_workspace_bytes(K, nnz);
  auto ws = scratch(ws_bytes, grad.place());
  PSA_CALL(psa_spmm_minmax_bw_eb(i64(colptr), i64_or_null(col_csc), i64(row_csc), tag.data(), f32_or_null(weight_csc),
                                 f32(grad), arg_bytes.data(), width, f32_or_null(hot_grad),
                                 hot_bytes ? hot_bytes.get().data() : nullptr, hot_grad ? hot_grad.get().shape()[0] : 0, M, N,
                                 K, nnz, gm.data<float>(), ws.data<uint8_t>(), ws_bytes, stream_of(grad)));
  return {gm};
}
PD_BUILD_OP(spmm_minmax_bw_eb)
    .Inputs({"colptr", paddle::Optional("col_csc"), "row_csc", "tag", paddle::Optional("weight_csc"), "grad", "arg_bytes",
             paddle::Optional("hot_grad"), paddle::Optional("hot_bytes")})
    .Outputs({"grad_mat"})
    .SetKernelFn(PD_KERNEL(spmm_minmax_bw_eb));

// sum / mean backward, both gradients in one pass over the CSC view (trainable edge values)
std::vector<paddle::Tensor> spmm_sum_bw_csc(paddle::Tensor& colptr, paddle::Tensor& row_csc, paddle::Tensor& csr2csc,
                                            paddle::Tensor& csc2csr, const paddle::optional<paddle::Tensor>& value,
                                            const paddle::optional<paddle::Tensor>& row_scale, paddle::Tensor& mat,
                                            paddle::Tensor& grad, const paddle::optional<paddle::Tensor>& hot_grad,
                                            bool want_value) {
  // hot_grad: compact copy of the rows of grad that row_csc names as M + position (row_scale then
  // carries their scales behind its M entries)
  CHECK_GPU(grad);
  const int64_t M = grad.shape()[0], K = grad.shape()[1], N = colptr.numel() - 1, nnz = csr2csc.numel();
  const auto place = grad.place();
  auto gv_csc = paddle::empty({want_value ? nnz : 0}, grad.dtype(), place);
  auto gv = paddle::empty({want_value ? nnz : 0}, grad.dtype(), place);
  auto gm = paddle::empty({N, K}, grad.dtype(), place);
  const size_t ws_bytes = psa_spmm_sum_bw_csc_workspace_bytes(K, nnz);
  auto ws = scratch(ws_bytes, place);
  PSA_CALL(psa_spmm_sum_bw_csc(i64(colptr), i64(row_csc), i64(csr2csc), f32_or_null(value), f32_or_null(row_scale),
                               want_value ? f32(mat) : nullptr, f32(grad), f32_or_null(hot_grad),
                               hot_grad ? hot_grad.get().shape()[0] : 0, M, N, K, nnz,
                               want_value ? gv_csc.data<float>() : nullptr, gm.data<float>(), ws.data<uint8_t>(),
                               ws_bytes, stream_of(grad)));
  if (want_value) PSA_CALL(psa_gather_rows(gv_csc.data<float>(), i64(csc2csr), nnz, 4, gv.data<float>(), stream_of(grad)));
  return {gv, gm};
}
PD_BUILD_OP(spmm_sum_bw_csc)
    .Inputs({"colptr", "row_csc", "csr2csc", "csc2csr", paddle::Optional("value"), paddle::Optional("row_scale"), "mat",
             "grad", paddle::Optional("hot_grad")})
    .Outputs({"grad_value", "grad_mat"})
    .Attrs({"want_value: bool"})
    .SetKernelFn(PD_KERNEL(spmm_sum_bw_csc));

// the same pass with fp16 / bf16 dense operands (fp32 sums; grad_value in CSC order, fp32): the caller brings it
// to CSR order with permute_apply / gather_rows, as paddle_sparse_amd/matmul.py does
std::vector<paddle::Tensor> spmm_half_sum_bw_csc(paddle::Tensor& colptr, paddle::Tensor& row_csc,
                                                 const paddle::optional<paddle::Tensor>& weight_csc,
                                                 const paddle::optional<paddle::Tensor>& row_scale, paddle::Tensor& mat,
                                                 paddle::Tensor& grad, bool want_value) {
  CHECK_GPU(grad);
  const int64_t M = grad.shape()[0], K = grad.shape()[1], N = colptr.numel() - 1, nnz = row_csc.numel();
  const auto place = grad.place();
  auto gv_csc = paddle::empty({want_value ? nnz : 0}, paddle::DataType::FLOAT32, place);
  auto gm = paddle::empty({N, K}, grad.dtype(), place);
  const size_t ws_bytes = psa_spmm_half_bw_csc_workspace_bytes(K, nnz);  // long columns in chunks (power-law transposes)
  auto ws = paddle::empty({static_cast<int64_t>(ws_bytes)}, paddle::DataType::UINT8, place);
  PSA_CALL(psa_spmm_half_sum_bw_csc(dtype_id_of(grad), i64(colptr), i64(row_csc), f32_or_null(weight_csc),
                                    f32_or_null(row_scale), want_value ? mat.data() : nullptr, grad.data(), M, N, K, nnz,
                                    want_value ? gv_csc.data<float>() : nullptr, gm.data(), ws_bytes ? ws.data() : nullptr,
                                    ws_bytes, stream_of(grad)));
  return {gv_csc, gm};
}
PD_BUILD_OP(spmm_half_sum_bw_csc)
    .Inputs({"colptr", "row_csc", paddle::Optional("weight_csc"), paddle::Optional("row_scale"), "mat", "grad"})
    .Outputs({"grad_value_csc", "grad_mat"})
    .Attrs({"want_value: bool"})
    .SetKernelFn(PD_KERNEL(spmm_half_sum_bw_csc));

// min / max with half-width operands: the forward that leaves the row-local arg_out (arg_width bytes per element)
// and the masked pass over the CSC view that reads it (both gradients; grad_value in CSC order, fp32)
std::vector<paddle::Tensor> spmm_half_arg(paddle::Tensor& rowptr, paddle::Tensor& col,
                                          const paddle::optional<paddle::Tensor>& value, paddle::Tensor& mat,
                                          int64_t reduce, int64_t arg_width) {
  CHECK_GPU(mat);
  const int64_t M = rowptr.numel() - 1, N = mat.shape()[0], K = mat.shape()[1], nnz = col.numel();
  auto out = paddle::empty({M, K}, mat.dtype(), mat.place());
  auto bytes = paddle::empty({M, K}, arg_width == 2 ? paddle::DataType::INT16 : paddle::DataType::UINT8, mat.place());
  PSA_CALL(psa_spmm_half_arg(static_cast<int>(reduce), dtype_id_of(mat), i64(rowptr), i64(col),
                             value ? value.get().data() : nullptr, value ? dtype_id_of(value.get()) : 0, mat.data(), M, N, K,
                             nnz, out.data(), nullptr, bytes.data(), static_cast<int>(arg_width), stream_of(mat)));
  return {out, bytes};
}
PD_BUILD_OP(spmm_half_arg)
    .Inputs({"rowptr", "col", paddle::Optional("value"), "mat"})
    .Outputs({"out", "arg_bytes"})
    .Attrs({"reduce: int64_t", "arg_width: int64_t"})
    .SetKernelFn(PD_KERNEL(spmm_half_arg));

std::vector<paddle::Tensor> spmm_half_minmax_bw_csc(paddle::Tensor& colptr, paddle::Tensor& row_csc, paddle::Tensor& tag,
                                                    const paddle::optional<paddle::Tensor>& weight_csc, paddle::Tensor& mat,
                                                    paddle::Tensor& grad, paddle::Tensor& arg_bytes, bool want_value) {
  CHECK_GPU(grad);
  const int64_t M = grad.shape()[0], K = grad.shape()[1], N = colptr.numel() - 1, nnz = row_csc.numel();
  const auto place = grad.place();
  const int width = tag.dtype() == paddle::DataType::INT16 ? 2 : 1;
  auto gv_csc = paddle::empty({want_value ? nnz : 0}, paddle::DataType::FLOAT32, place);
  auto gm = paddle::empty({N, K}, grad.dtype(), place);
  const size_t ws_bytes = psa_spmm_half_bw_csc_workspace_bytes(K, nnz);
  auto ws = paddle::empty({static_cast<int64_t>(ws_bytes)}, paddle::DataType::UINT8, place);
  PSA_CALL(psa_spmm_half_minmax_bw_csc(dtype_id_of(grad), i64(colptr), i64(row_csc), tag.data(), f32_or_null(weight_csc),
                                       want_value ? mat.data() : nullptr, grad.data(), arg_bytes.data(), width, M, N, K, nnz,
                                       want_value ? gv_csc.data<float>() : nullptr, gm.data(),
                                       ws_bytes ? ws.data() : nullptr, ws_bytes, stream_of(grad)));
  return {gv_csc, gm};
}
PD_BUILD_OP(spmm_half_minmax_bw_csc)
    .Inputs({"colptr", "row_csc", "tag", paddle::Optional("weight_csc"), "mat", "grad", "arg_bytes"})
    .Outputs({"grad_value_csc", "grad_mat"})
    .Attrs({"want_value: bool"})
    .SetKernelFn(PD_KERNEL(spmm_half_minmax_bw_csc));

// ---- the index-arithmetic seams of storage.py / tensor.py / reduce.py ------
// bincount + count2ptr: colcount / colptr (storage.py:397-398, 414-418)
std::vector<paddle::Tensor> bincount(paddle::Tensor& index, int64_t size) {
  CHECK_GPU(index);
  CHECK_I64(index);
  auto out = i64_empty(size, index.place());
  PSA_CALL(psa_bincount(i64(index), index.numel(), size, out.data<int64_t>(), stream_of(index)));
  return {out};
}
PD_BUILD_OP(bincount).Inputs({"index"}).Outputs({"out"}).Attrs({"size: int64_t"}).SetKernelFn(PD_KERNEL(bincount));

std::vector<paddle::Tensor> count2ptr(paddle::Tensor& counts) {
  CHECK_GPU(counts);
  CHECK_I64(counts);
  const int64_t n = counts.numel();
  auto out = i64_empty(n + 1, counts.place());
  const size_t ws_bytes = psa_count2ptr_workspace_bytes(n);
  auto ws = scratch(ws_bytes, counts.place());
  PSA_CALL(psa_count2ptr(i64(counts), n, out.data<int64_t>(), ws.data<uint8_t>(), ws_bytes, stream_of(counts)));
  return {out};
}
PD_BUILD_OP(count2ptr).Inputs({"counts"}).Outputs({"out"}).SetKernelFn(PD_KERNEL(count2ptr));

// invert_permutation: csc2csr without the second sort (storage.py:444-445)
std::vector<paddle::Tensor> invert_permutation(paddle::Tensor& perm) {
  CHECK_GPU(perm);
  CHECK_I64(perm);
  auto inv = i64_empty(perm.numel(), perm.place());
  PSA_CALL(psa_invert_permutation(i64(perm), perm.numel(), inv.data<int64_t>(), stream_of(perm)));
  return {inv};
}
PD_BUILD_OP(invert_permutation).Inputs({"perm"}).Outputs({"inv"}).SetKernelFn(PD_KERNEL(invert_permutation));

// permute_plan_pack / permute_apply: a 4-byte array through a fixed permutation in two streaming
// passes (value[csr2csc] of tensor.py:254-257 / transpose.py:19-22, grad_value's way back from
// the pass over the CSC view).  The plan's sorts and inverse are the index_sort /
// invert_permutation ops above, composed as paddle_sparse_amd/ops.py::permute_plan does.
std::vector<paddle::Tensor> permute_plan_pack(paddle::Tensor& perm_ts, paddle::Tensor& gslot, paddle::Tensor& perm_mid,
                                              paddle::Tensor& dest) {
  CHECK_GPU(dest);
  CHECK_I64(perm_ts);
  CHECK_I64(gslot);
  CHECK_I64(perm_mid);
  CHECK_I64(dest);
  const int64_t n = dest.numel();
  auto sl = paddle::empty({n}, paddle::DataType::INT16, dest.place());
  auto gs = paddle::empty({n}, paddle::DataType::INT32, dest.place());
  auto lo = paddle::empty({n}, paddle::DataType::INT16, dest.place());
  PSA_CALL(psa_permute_plan_pack(i64(perm_ts), i64(gslot), i64(perm_mid), i64(dest), n, sl.data(), gs.data(), lo.data(),
                                 stream_of(dest)));
  return {sl, gs, lo};
}
PD_BUILD_OP(permute_plan_pack)
    .Inputs({"perm_ts", "gslot", "perm_mid", "dest"})
    .Outputs({"sl", "gs", "lo"})
    .SetKernelFn(PD_KERNEL(permute_plan_pack));

std::vector<paddle::Tensor> permute_apply(paddle::Tensor& src, paddle::Tensor& sl, paddle::Tensor& gs, paddle::Tensor& lo) {
  CHECK_GPU(src);
  const int64_t n = src.numel();
  PD_CHECK(n == sl.numel() && n == gs.numel() && n == lo.numel(), "permute_apply: the plan is for another length");
  PD_CHECK(src.dtype() == paddle::DataType::FLOAT32 || src.dtype() == paddle::DataType::INT32,
           "permute_apply takes 4-byte elements");
  auto mid = paddle::empty({n}, src.dtype(), src.place());
  auto out = paddle::empty({n}, src.dtype(), src.place());
  PSA_CALL(psa_permute_apply_u32(src.data(), sl.data(), gs.data(), lo.data(), n, mid.data(), out.data(), stream_of(src)));
  return {out};
}
PD_BUILD_OP(permute_apply).Inputs({"src", "sl", "gs", "lo"}).Outputs({"out"}).SetKernelFn(PD_KERNEL(permute_apply));

// make_keys / split_keys: row * N + col and back (storage.py:159-163, 166-168)
std::vector<paddle::Tensor> make_keys(paddle::Tensor& a, paddle::Tensor& b, int64_t mul) {
  CHECK_GPU(a);
  CHECK_I64(a);
  CHECK_I64(b);
  auto keys = i64_empty(a.numel(), a.place());
  auto unsorted = paddle::zeros({1}, paddle::DataType::INT32, a.place());
  PSA_CALL(psa_make_keys(i64(a), i64(b), mul, a.numel(), keys.data<int64_t>(), unsorted.data<int32_t>(), stream_of(a)));
  return {keys, unsorted};
}
PD_BUILD_OP(make_keys).Inputs({"a", "b"}).Outputs({"keys", "unsorted"}).Attrs({"mul: int64_t"}).SetKernelFn(PD_KERNEL(make_keys));

std::vector<paddle::Tensor> make_keys_checked(paddle::Tensor& row, paddle::Tensor& col, int64_t M, int64_t N) {
  CHECK_GPU(row);
  CHECK_I64(row);
  CHECK_I64(col);
  auto keys = i64_empty(row.numel(), row.place());
  auto status = i64_empty(4, row.place());
  PSA_CALL(psa_make_keys_checked(i64(row), i64(col), row.numel(), M, N, keys.data<int64_t>(), status.data<int64_t>(),
                                 stream_of(row)));
  return {keys, status};
}
PD_BUILD_OP(make_keys_checked)
    .Inputs({"row", "col"})
    .Outputs({"keys", "status"})
    .Attrs({"M: int64_t", "N: int64_t"})
    .SetKernelFn(PD_KERNEL(make_keys_checked));

std::vector<paddle::Tensor> split_keys(paddle::Tensor& keys, int64_t div) {
  CHECK_GPU(keys);
  CHECK_I64(keys);
  auto hi = i64_empty(keys.numel(), keys.place()), lo = i64_empty(keys.numel(), keys.place());
  PSA_CALL(psa_split_keys(i64(keys), keys.numel(), div, hi.data<int64_t>(), lo.data<int64_t>(), stream_of(keys)));
  return {hi, lo};
}
PD_BUILD_OP(split_keys).Inputs({"keys"}).Outputs({"hi", "lo"}).Attrs({"div: int64_t"}).SetKernelFn(PD_KERNEL(split_keys));

// sort_pairs: the stable sort carrying a 4-byte value instead of the permutation (storage.py:164-169)
std::vector<paddle::Tensor> sort_pairs(paddle::Tensor& keys, paddle::Tensor& payload, int64_t max_value) {
  CHECK_GPU(keys);
  CHECK_I64(keys);
  const int64_t n = keys.numel();
  auto sorted = i64_empty(n, keys.place());
  auto out = paddle::empty({n}, payload.dtype(), keys.place());
  const size_t ws_bytes = psa_index_sort_workspace_bytes(n, max_value);
  auto ws = scratch(ws_bytes, keys.place());
  PSA_CALL(psa_sort_pairs_u32(i64(keys), payload.data(), n, max_value, sorted.data<int64_t>(), out.data(),
                              ws.data<uint8_t>(), ws_bytes, stream_of(keys)));
  if (n > 0) PSA_SORT_OK(ws.data<uint8_t>(), n, max_value, stream_of(keys));
  return {sorted, out};
}
PD_BUILD_OP(sort_pairs)
    .Inputs({"keys", "payload"})
    .Outputs({"sorted", "payload_out"})
    .Attrs({"max_value: int64_t"})
    .SetKernelFn(PD_KERNEL(sort_pairs));

// gather_rows: x[perm] along dim 0 (storage.py:166-169, transpose.py:14-22, tensor.py:252-257)
std::vector<paddle::Tensor> gather_rows(paddle::Tensor& src, paddle::Tensor& perm) {
  CHECK_GPU(src);
  CHECK_I64(perm);
  auto shape = src.shape();
  const int64_t n = perm.numel();
  const int64_t row_bytes = row_elems(src) * static_cast<int64_t>(paddle::SizeOf(src.dtype()));
  shape[0] = n;
  auto out = paddle::empty(shape, src.dtype(), src.place());
  PSA_CALL(psa_gather_rows(src.data(), i64(perm), n, row_bytes, out.data(), stream_of(src)));
  return {out};
}
PD_BUILD_OP(gather_rows).Inputs({"src", "perm"}).Outputs({"out"}).SetKernelFn(PD_KERNEL(gather_rows));

// gather_rows_window: src[perm, col0:col0 + width] packed densely (halo exchange of the row-partitioned SpMM)
std::vector<paddle::Tensor> gather_rows_window(paddle::Tensor& src, paddle::Tensor& perm, int64_t col0, int64_t width) {
  CHECK_GPU(src);
  CHECK_I64(perm);
  PD_CHECK(src.shape().size() == 2, "src must be 2-D");
  const int64_t n = perm.numel(), es = static_cast<int64_t>(paddle::SizeOf(src.dtype()));
  auto out = paddle::empty({n, width}, src.dtype(), src.place());
  PSA_CALL(psa_gather_rows_window(src.data(), src.shape()[1] * es, col0 * es, width * es, i64(perm), n, out.data(),
                                  stream_of(src)));
  return {out};
}
PD_BUILD_OP(gather_rows_window)
    .Inputs({"src", "perm"})
    .Outputs({"out"})
    .Attrs({"col0: int64_t", "width: int64_t"})
    .SetKernelFn(PD_KERNEL(gather_rows_window));

// merge_sorted: the cat + argsort of add.py:30-47 / mul.py:57-73 / tensor.py:415-451 on two sorted halves
std::vector<paddle::Tensor> merge_sorted(paddle::Tensor& a, paddle::Tensor& b) {
  CHECK_GPU(a);
  CHECK_I64(a);
  CHECK_I64(b);
  const int64_t na = a.numel(), nb = b.numel();
  auto merged = i64_empty(na + nb, a.place()), source = i64_empty(na + nb, a.place());
  PSA_CALL(psa_merge_sorted(i64(a), na, i64(b), nb, nullptr, nullptr, merged.data<int64_t>(), source.data<int64_t>(),
                            nullptr, stream_of(a)));
  return {merged, source};
}
PD_BUILD_OP(merge_sorted).Inputs({"a", "b"}).Outputs({"merged", "source"}).SetKernelFn(PD_KERNEL(merge_sorted));

// unique_sorted: head flags / row[mask] / col[mask] / ptr of storage.py:455-470 (one host read: the count)
std::vector<paddle::Tensor> unique_sorted(paddle::Tensor& sorted_keys, int64_t N) {
  CHECK_GPU(sorted_keys);
  CHECK_I64(sorted_keys);
  const int64_t n = sorted_keys.numel();
  const auto place = sorted_keys.place();
  void* s = stream_of(sorted_keys);
  const size_t ws_bytes = psa_unique_workspace_bytes(n);
  auto ws = scratch(ws_bytes, place);
  auto count = i64_empty(1, place);
  PSA_CALL(psa_unique_count(i64(sorted_keys), n, ws.data<uint8_t>(), ws_bytes, count.data<int64_t>(), s));
  const int64_t distinct = read_i64(count, 0, 1)[0];
  auto ptr = i64_empty(distinct + 1, place);
  auto index = paddle::empty({2, distinct}, paddle::DataType::INT64, place);
  if (n > 0)
    PSA_CALL(psa_unique_write(i64(sorted_keys), n, N, ws.data<uint8_t>(), i64(count), ptr.data<int64_t>(),
                              index.data<int64_t>(), index.data<int64_t>() + distinct, s));
  return {ptr, index};
}
PD_BUILD_OP(unique_sorted)
    .Inputs({"sorted_keys"})
    .Outputs({"ptr", "index"})
    .Attrs({"N: int64_t"})
    .SetKernelFn(PD_KERNEL(unique_sorted));

// unique_sorted + segment_csr for 4-byte scalar values already in the keys' sorted order (they rode the sort as its
// payload): the [2, count] index and the reduced values; one launch for both when the runs are short (count * 32 > n),
// else psa_unique_write + psa_segment_reduce (a wave per run) — as paddle_sparse_amd/ops.py::unique_sorted_reduce
std::vector<paddle::Tensor> unique_sorted_reduce(paddle::Tensor& sorted_keys, paddle::Tensor& payload, int64_t N,
                                                 int64_t reduce) {
  CHECK_GPU(sorted_keys);
  CHECK_I64(sorted_keys);
  const int64_t n = sorted_keys.numel();
  PD_CHECK(payload.numel() == n && (payload.dtype() == paddle::DataType::FLOAT32 || payload.dtype() == paddle::DataType::INT32),
           "unique_sorted_reduce: payload must be float32[n] or int32[n]");
  const auto place = sorted_keys.place();
  void* s = stream_of(sorted_keys);
  const size_t ws_bytes = psa_unique_workspace_bytes(n);
  auto ws = scratch(ws_bytes, place);
  auto count = i64_empty(1, place);
  PSA_CALL(psa_unique_count(i64(sorted_keys), n, ws.data<uint8_t>(), ws_bytes, count.data<int64_t>(), s));
  const int64_t distinct = read_i64(count, 0, 1)[0];
  auto index = paddle::empty({2, distinct}, paddle::DataType::INT64, place);
  auto value = paddle::empty({distinct}, payload.dtype(), place);
  if (n == 0) return {index, value};
  if (distinct < n && distinct * 32 > n) {
    PSA_CALL(psa_unique_write_reduce(static_cast<int>(reduce), dtype_id_of(payload), i64(sorted_keys), n, N, ws.data<uint8_t>(),
                                     i64(count), index.data<int64_t>(), payload.data(), value.data(), s));
    return {index, value};
  }
  auto ptr = i64_empty(distinct + 1, place);
  PSA_CALL(psa_unique_write(i64(sorted_keys), n, N, ws.data<uint8_t>(), i64(count), ptr.data<int64_t>(),
                            index.data<int64_t>(), index.data<int64_t>() + distinct, s));
  PSA_CALL(psa_segment_reduce(static_cast<int>(reduce), dtype_id_of(payload), payload.data(), nullptr, ptr.data<int64_t>(),
                              distinct, 1, n, value.data(), s));
  return {index, value};
}
PD_BUILD_OP(unique_sorted_reduce)
    .Inputs({"sorted_keys", "payload"})
    .Outputs({"index", "value"})
    .Attrs({"N: int64_t", "reduce: int64_t"})
    .SetKernelFn(PD_KERNEL(unique_sorted_reduce));

// scatter: paddle_scatter.scatter(src, index, 0, None, dim_size, reduce) of reduce.py:42
std::vector<paddle::Tensor> scatter(paddle::Tensor& src, paddle::Tensor& index, int64_t dim_size, int64_t reduce) {
  CHECK_GPU(src);
  CHECK_I64(index);
  auto shape = src.shape();
  shape[0] = dim_size;
  auto out = paddle::empty(shape, src.dtype(), src.place());
  const size_t ws_bytes = psa_scatter_workspace_bytes(dim_size);
  auto ws = scratch(ws_bytes, src.place());
  PSA_CALL(psa_scatter_reduce(static_cast<int>(reduce), dtype_id_of(src), src.data(), i64(index), index.numel(),
                              row_elems(src), dim_size, out.data(), ws.data<uint8_t>(), ws_bytes, stream_of(src)));
  return {out};
}
PD_BUILD_OP(scatter)
    .Inputs({"src", "index"})
    .Outputs({"out"})
    .Attrs({"dim_size: int64_t", "reduce: int64_t"})
    .SetKernelFn(PD_KERNEL(scatter));

// ---- segment_csr with an optional gather permutation (storage.py:471) -----------------
std::vector<paddle::Tensor> segment_csr_perm(paddle::Tensor& src, paddle::Tensor& ptr,
                                             const paddle::optional<paddle::Tensor>& perm, int64_t reduce) {
  CHECK_GPU(src);
  const int dtype = dtype_id_of(src);
  auto shape = src.shape();
  const int64_t nseg = ptr.numel() - 1;
  int64_t D = 1;
  for (size_t i = 1; i < shape.size(); ++i) D *= shape[i];
  const int64_t n = perm ? perm.get().numel() : shape[0];
  shape[0] = nseg;
  auto out = paddle::empty(shape, src.dtype(), src.place());
  PSA_CALL(psa_segment_reduce(static_cast<int>(reduce), dtype, src.data(), perm ? i64(perm.get()) : nullptr,
                              i64(ptr), nseg, D, n, out.data(), stream_of(src)));
  return {out};
}
PD_BUILD_OP(segment_csr_perm)
    .Inputs({"src", "ptr", paddle::Optional("perm")})
    .Outputs({"out"})
    .Attrs({"reduce: int64_t"})
    .SetKernelFn(PD_KERNEL(segment_csr_perm));

// ---- sample_adj (csrc/sample.cpp:8-40: same name, inputs, outputs, attrs) ----------
// The reference's kernel function throws "No CUDA version supported" for GPU
// tensors (csrc/sample.cpp:13-18); this one runs the chain of csrc/sample.hip.
// Two sizes are data dependent (as in the reference, which skips InferShape for
// that reason, csrc/sample.cpp:33): E and the number of new nodes are read back
// from the device.  `seed` comes from Paddle's default generator so that
// paddle.seed() governs the draws as it does in the reference.
std::vector<paddle::Tensor> sample_adj(paddle::Tensor& rowptr, paddle::Tensor& col, paddle::Tensor& idx,
                                       int64_t num_neighbors, bool replace) {
  CHECK_GPU(rowptr);
  CHECK_GPU(col);
  CHECK_GPU(idx);
  CHECK_I64(rowptr);
  CHECK_I64(col);
  CHECK_I64(idx);
  PD_CHECK(idx.shape().size() == 1, "idx must be 1-D");
  void* s = stream_of(rowptr);
  const auto place = rowptr.place();
  const int64_t S = idx.numel(), num_rows = rowptr.numel() - 1;
  const uint64_t seed = static_cast<uint64_t>(
      paddle::experimental::randint(0, INT64_MAX, {1}, paddle::DataType::INT64, paddle::CPUPlace()).data<int64_t>()[0]);
  auto i64_empty = [&](int64_t n) { return paddle::empty({n}, paddle::DataType::INT64, place); };
  auto scan = [&](const paddle::Tensor& counts, int64_t n) {
    auto out = i64_empty(n + 1);
    const int64_t wsb = static_cast<int64_t>(psa_count2ptr_workspace_bytes(n));
    auto ws = paddle::empty({wsb}, paddle::DataType::UINT8, place);
    PSA_CALL(psa_count2ptr(i64(counts), n, out.data<int64_t>(), ws.data<uint8_t>(), static_cast<size_t>(wsb), s));
    return out;
  };
  auto last = [&](const paddle::Tensor& ptr, int64_t n) {  // one 8-byte device -> host read
    return paddle::experimental::slice(ptr, {0}, {n}, {n + 1}, {}, {}).copy_to(paddle::CPUPlace(), true).data<int64_t>()[0];
  };

  auto counts = i64_empty(S);
  PSA_CALL(psa_sample_count(i64(rowptr), i64(idx), S, num_neighbors, replace, counts.data<int64_t>(), s));
  auto out_rowptr = scan(counts, S);
  const int64_t E = last(out_rowptr, S);
  auto owner = i64_empty(E), e_raw = i64_empty(E), flags = i64_empty(E), keys = i64_empty(E);
  PSA_CALL(psa_ptr2ind(i64(out_rowptr), S, E, owner.data<int64_t>(), s));
  PSA_CALL(psa_sample_select(i64(rowptr), i64(idx), S, i64(out_rowptr), i64(owner), E, num_neighbors, replace, seed,
                             e_raw.data<int64_t>(), s));
  const int64_t num_nodes =
      std::max<int64_t>(num_rows, col.numel() ? paddle::experimental::max(col, {}, false)
                                                        .copy_to(paddle::CPUPlace(), true).data<int64_t>()[0] + 1 : 0);
  auto newid = i64_empty(num_nodes);
  PSA_CALL(psa_relabel_mark(i64(idx), S, i64(col), i64(e_raw), E, num_nodes, newid.data<int64_t>(),
                            flags.data<int64_t>(), s));
  auto rank = scan(flags, E);
  const int64_t n_out = S + last(rank, E);
  auto n_id = i64_empty(n_out);
  PSA_CALL(psa_relabel_finish(i64(idx), S, i64(col), i64(e_raw), E, i64(newid), i64(rank), i64(owner), n_out,
                              n_id.data<int64_t>(), keys.data<int64_t>(), s));
  // rows by new column id: one stable sort of i * n_out + id, then read id back
  auto sorted = i64_empty(E), perm = i64_empty(E), out_col = i64_empty(E), e_id = i64_empty(E);
  const int64_t max_key = std::max<int64_t>(S * n_out, 1);
  const int64_t wsb = static_cast<int64_t>(psa_index_sort_workspace_bytes(E, max_key));
  auto ws = paddle::empty({wsb > 0 ? wsb : 1}, paddle::DataType::UINT8, place);
  PSA_CALL(psa_index_sort(i64(keys), E, max_key, sorted.data<int64_t>(), perm.data<int64_t>(), ws.data<uint8_t>(),
                          static_cast<size_t>(wsb), s));
  if (E > 0) PSA_SORT_OK(ws.data<uint8_t>(), E, max_key, s);
  if (E > 0) PSA_CALL(psa_split_keys(i64(sorted), E, n_out, nullptr, out_col.data<int64_t>(), s));
  PSA_CALL(psa_gather_rows(e_raw.data<int64_t>(), i64(perm), E, 8, e_id.data<int64_t>(), s));
  return {out_rowptr, out_col, n_id, e_id};
}
std::vector<paddle::DataType> sample_adj_infer_dtype(const paddle::DataType rowptr_dtype, const paddle::DataType col_dtype,
                                                     const paddle::DataType idx_dtype) {
  return {rowptr_dtype, col_dtype, col_dtype, col_dtype};
}
PD_BUILD_OP(sample_adj)
    .Inputs({"rowptr", "col", "idx"})
    .Outputs({"out_rowptr", "out_col", "out_n_id", "out_e_id"})
    .Attrs({"num_neighbors: int64_t", "replace: bool"})
    .SetKernelFn(PD_KERNEL(sample_adj))
    .SetInferDtypeFn(PD_INFER_DTYPE(sample_adj_infer_dtype));

// ---- coalesce as ONE op (seam: paddle_sparse/coalesce.py:25-29) ----------------------
// psa_coalesce_count -> one host read (count + flags) -> psa_coalesce_write; the same two
// calls paddle_sparse_amd/coalesce.py makes.  index int64[2, nnz] + optional value
// [nnz, ...] -> coalesced pair.  `reduce`: psa_reduce.  transpose(index, value, m, n) of
// transpose.py:41-65 is this op on the swapped index rows with (n, m).
std::vector<paddle::Tensor> coalesce(paddle::Tensor& index, const paddle::optional<paddle::Tensor>& value,
                                     int64_t m, int64_t n, int64_t reduce) {
  CHECK_GPU(index);
  CHECK_I64(index);
  PD_CHECK(index.shape().size() == 2 && index.shape()[0] == 2, "index must be [2, nnz]");
  const int64_t nnz = index.shape()[1];
  void* s = stream_of(index);
  const auto place = index.place();
  // row, then col: the op reads `index` as one dense [2, nnz] block.  coalesce.py hands over
  // what `paddle.stack([row, col])` / a user's fresh tensor gives; a strided view (e.g. a
  // transposed [nnz, 2] array) must be made dense by the Python seam first
  // (`index = index.contiguous()` in coalesce.py:25 — INTEGRATION.md §2), since the
  // custom-op Tensor API the reference uses has no stride query.
  const int64_t* row = i64(index);
  const int64_t* col = i64(index) + nnz;
  if (nnz == 0) return {index, value ? value.get() : paddle::empty({0}, paddle::DataType::FLOAT32, place)};
  const int dtype = value ? dtype_id_of(value.get()) : 0;
  const int64_t D = value ? row_elems(value.get()) : 0;
  const void* v = value ? value.get().data() : nullptr;
  const size_t ws_bytes = psa_coalesce_workspace_bytes(nnz, m, n);
  auto ws = scratch(ws_bytes, place);
  PSA_CALL(psa_coalesce_count(row, col, v, dtype, D, nnz, m, n, ws.data<uint8_t>(), ws_bytes, s));
  // the status words {count, flags} are the first 16 bytes of the workspace: the one host read
  auto status_words = paddle::experimental::slice(ws, {0}, {0}, {16}, {}, {}).copy_to(paddle::CPUPlace(), true);
  const int64_t distinct = reinterpret_cast<const int64_t*>(status_words.data<uint8_t>())[0];
  const int64_t flags = reinterpret_cast<const int64_t*>(status_words.data<uint8_t>())[1];
  PD_CHECK(!(flags & 1), "coalesce: an index lies outside the m x n matrix");  // storage.py:78-91 asserts it
  PD_CHECK(!(flags & 4), "coalesce: an inter-workgroup wait of the radix sort gave up; the result is invalid");
  auto out_index = paddle::empty({2, distinct}, paddle::DataType::INT64, place);
  auto shape = value ? value.get().shape() : std::vector<int64_t>{0};
  shape[0] = value ? distinct : 0;
  auto out_value = paddle::empty(shape, value ? value.get().dtype() : paddle::DataType::FLOAT32, place);
  PSA_CALL(psa_coalesce_write(v, dtype, D, nnz, m, n, static_cast<int>(reduce), distinct, ws.data<uint8_t>(),
                              out_index.data<int64_t>(), value ? out_value.data() : nullptr, s));
  return {out_index, out_value};
}
PD_BUILD_OP(coalesce)
    .Inputs({"index", paddle::Optional("value")})
    .Outputs({"out_index", "out_value"})
    .Attrs({"m: int64_t", "n: int64_t", "reduce: int64_t"})
    .SetKernelFn(PD_KERNEL(coalesce));
