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

#define CHECK_GPU(x) PD_CHECK((x).is_gpu(), #x " must be a GPU tensor (HIP build has no CPU path)")
#define CHECK_I64(x) PD_CHECK((x).dtype() == paddle::DataType::INT64, #x " must be int64")

namespace {
inline void* stream_of(const paddle::Tensor& t) { return reinterpret_cast<void*>(t.stream()); }
inline const int64_t* i64(const paddle::Tensor& t) { return t.data<int64_t>(); }
inline const float* f32(const paddle::Tensor& t) { return t.data<float>(); }
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
// `value` is optional in the Python API; Paddle's paddle::optional<Tensor>
// input (anticipated by PD_DISPATCH_HAS_VALUE, csrc/cpu/utils.h:11-20) maps
// to a NULL pointer in the C-ABI.
static std::vector<paddle::Tensor> spmm_impl(int reduce, paddle::Tensor& rowptr, paddle::Tensor& col,
                                             const paddle::optional<paddle::Tensor>& value,
                                             paddle::Tensor& mat) {
  CHECK_GPU(mat);
  CHECK_I64(rowptr);
  CHECK_I64(col);
  PD_CHECK(mat.dtype() == paddle::DataType::FLOAT32, "spmm is fp32 in the HIP build");
  const int64_t M = rowptr.numel() - 1, N = mat.shape()[0], K = mat.shape()[1], nnz = col.numel();
  auto out = paddle::empty({M, K}, mat.dtype(), mat.place());
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  auto arg = minmax ? paddle::empty({M, K}, paddle::DataType::INT64, mat.place())
                    : paddle::empty({0}, paddle::DataType::INT64, mat.place());
  // scratch of the long-row path (rows > 128 edges are reduced chunk-wise)
  const int64_t ws_bytes = static_cast<int64_t>(psa_spmm_workspace_bytes(reduce, K, nnz));
  auto ws = paddle::empty({ws_bytes > 0 ? ws_bytes : 1}, paddle::DataType::UINT8, mat.place());
  PSA_CALL(psa_spmm(reduce, i64(rowptr), i64(col), value ? f32(value.get()) : nullptr, f32(mat), M, N,
                    K, nnz, out.data<float>(), minmax ? arg.data<int64_t>() : nullptr,
                    /*arg_bytes=*/nullptr, ws_bytes > 0 ? ws.data<uint8_t>() : nullptr,
                    static_cast<size_t>(ws_bytes),
                    stream_of(mat)));
  return {out, arg};
}
#define PSA_SPMM_OP(NAME, RED)                                                                \
  std::vector<paddle::Tensor> NAME(paddle::Tensor& rowptr, paddle::Tensor& col,                \
                                   const paddle::optional<paddle::Tensor>& value,             \
                                   paddle::Tensor& mat) {                                     \
    return spmm_impl(RED, rowptr, col, value, mat);                                           \
  }                                                                                           \
  PD_BUILD_OP(NAME)                                                                           \
      .Inputs({"rowptr", "col", paddle::Optional("value"), "mat"})                            \
      .Outputs({"out", "arg_out"})                                                            \
      .SetKernelFn(PD_KERNEL(NAME));
PSA_SPMM_OP(spmm_sum, PSA_SUM)
PSA_SPMM_OP(spmm_mean, PSA_MEAN)
PSA_SPMM_OP(spmm_min, PSA_MIN)
PSA_SPMM_OP(spmm_max, PSA_MAX)

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

// ---- segment_csr with an optional gather permutation (storage.py:471) -----------------
std::vector<paddle::Tensor> segment_csr_perm(paddle::Tensor& src, paddle::Tensor& ptr,
                                             const paddle::optional<paddle::Tensor>& perm, int64_t reduce) {
  CHECK_GPU(src);
  int dtype = -1;
  switch (src.dtype()) {
    case paddle::DataType::FLOAT32: dtype = PSA_F32; break;
    case paddle::DataType::FLOAT64: dtype = PSA_F64; break;
    case paddle::DataType::INT32: dtype = PSA_I32; break;
    case paddle::DataType::INT64: dtype = PSA_I64; break;
    case paddle::DataType::FLOAT16: dtype = PSA_F16; break;
    case paddle::DataType::BFLOAT16: dtype = PSA_BF16; break;
    default: PD_THROW("segment_csr: unsupported dtype");
  }
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
// The Python layer of this repository composes the chain call by call (and pays
// ~10 us of interpreter per call); inside a custom op the same chain is plain
// C++.  index int64[2, nnz] + optional value [nnz, ...] -> coalesced pair.
// `reduce`: psa_reduce.  One device -> host read (the number of distinct entries).
std::vector<paddle::Tensor> coalesce(paddle::Tensor& index, const paddle::optional<paddle::Tensor>& value,
                                     int64_t m, int64_t n, int64_t reduce) {
  CHECK_GPU(index);
  CHECK_I64(index);
  PD_CHECK(index.shape().size() == 2 && index.shape()[0] == 2, "index must be [2, nnz]");
  const int64_t nnz = index.shape()[1];
  void* s = stream_of(index);
  const auto place = index.place();
  auto i64_empty = [&](std::vector<int64_t> shape) { return paddle::empty(shape, paddle::DataType::INT64, place); };
  auto bytes = [&](size_t b) { return paddle::empty({static_cast<int64_t>(b > 0 ? b : 1)}, paddle::DataType::UINT8, place); };
  const int64_t* row = i64(index);            // index is contiguous [2, nnz]: row, then col
  const int64_t* col = i64(index) + nnz;
  if (nnz == 0) return {index, value ? value.get() : paddle::empty({0}, paddle::DataType::FLOAT32, place)};

  auto keys = i64_empty({nnz}), sorted = i64_empty({nnz}), perm = i64_empty({nnz}), count = i64_empty({1});
  PSA_CALL(psa_make_keys(row, col, n, nnz, keys.data<int64_t>(), nullptr, s));
  auto sort_ws = bytes(psa_index_sort_workspace_bytes(nnz, m * n));
  PSA_CALL(psa_index_sort(i64(keys), nnz, m * n, sorted.data<int64_t>(), perm.data<int64_t>(),
                          sort_ws.data<uint8_t>(), psa_index_sort_workspace_bytes(nnz, m * n), s));
  auto uniq_ws = bytes(psa_unique_workspace_bytes(nnz));
  PSA_CALL(psa_unique_count(i64(sorted), nnz, uniq_ws.data<uint8_t>(), psa_unique_workspace_bytes(nnz),
                            count.data<int64_t>(), s));
  const int64_t distinct = count.copy_to(paddle::CPUPlace(), true).data<int64_t>()[0];
  auto out_index = i64_empty({2, distinct});
  auto ptr = i64_empty({distinct + 1});
  PSA_CALL(psa_unique_write(i64(sorted), nnz, n, uniq_ws.data<uint8_t>(), i64(count), ptr.data<int64_t>(),
                            out_index.data<int64_t>(), out_index.data<int64_t>() + distinct, s));
  if (!value) return {out_index, paddle::empty({0}, paddle::DataType::FLOAT32, place)};
  const paddle::Tensor& v = value.get();
  int dtype = -1;
  switch (v.dtype()) {
    case paddle::DataType::FLOAT32: dtype = PSA_F32; break;
    case paddle::DataType::FLOAT64: dtype = PSA_F64; break;
    case paddle::DataType::INT32: dtype = PSA_I32; break;
    case paddle::DataType::INT64: dtype = PSA_I64; break;
    case paddle::DataType::FLOAT16: dtype = PSA_F16; break;
    case paddle::DataType::BFLOAT16: dtype = PSA_BF16; break;
    default: PD_THROW("coalesce: unsupported value dtype");
  }
  auto shape = v.shape();
  int64_t D = 1;
  for (size_t i = 1; i < shape.size(); ++i) D *= shape[i];
  shape[0] = distinct;
  auto out_value = paddle::empty(shape, v.dtype(), place);
  PSA_CALL(psa_segment_reduce(static_cast<int>(reduce), dtype, v.data(), i64(perm), i64(ptr), distinct, D, nnz,
                              out_value.data(), s));
  return {out_index, out_value};
}
PD_BUILD_OP(coalesce)
    .Inputs({"index", paddle::Optional("value")})
    .Outputs({"out_index", "out_value"})
    .Attrs({"m: int64_t", "n: int64_t", "reduce: int64_t"})
    .SetKernelFn(PD_KERNEL(coalesce));
