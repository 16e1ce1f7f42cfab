// SpMM backward pieces, fp32, gfx950.  Not in the reference (README.md:47-50);
// semantics are upstream pytorch_sparse (spmm_value_bw_cpu and the Python
// backward of torch_sparse/matmul.py), restated in oracle/spmm_oracle.c.
//
//   psa_spmm_value_bw      gV[e] = <mat[col[e],:], gOut[row(e),:]> (/deg for
//                          mean) — SDDMM-shaped.  One wave per CSR row: the
//                          gOut row is read once per ROW (kept in registers),
//                          so HBM traffic is the forward's, not the 2x of an
//                          edge-parallel kernel; dots fold with wave shuffles.
//   psa_transpose_weights  w'[j] = value[csr2csc[j]] (/deg(row) for mean): the
//                          CSC-ordered weights that turn gB = A^T gOut into a
//                          plain psa_spmm over (colptr, row[csr2csc], w').
//   psa_spmm_minmax_bw     scatter through arg_out with float atomics.
#include "common.h"
#include "lane_fold.h"
#include "long_rows.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<4> {
  using T = float4;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&dst)[VEC]) {
  using T = typename Vec<VEC>::T;
  const T v = *reinterpret_cast<const T*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) dst[i] = f[i];
}

// out[e] = <mat[col[e], :], grow[:]> / denom for the edges [s, e) of ONE row
// (grow = that row of gOut).  Called by the row wave for ordinary rows and by a
// chunk wave for a 128-edge piece of a long row.  ONE: K fits one tile of LPR * VEC floats (no tile loop, the
// column ids are consumed where they are fetched: 72 -> 64 VGPRs at K = 128, 8 waves per SIMD).
template <int VEC, int LPR, int U, bool ONE>
__device__ __forceinline__ void value_bw_range(const int64_t* __restrict__ col,
                                               const float* __restrict__ mat,
                                               const float* __restrict__ grow,
                                               float* __restrict__ out, int64_t K,
                                               int64_t s, int64_t e, float denom, int lane) {
  constexpr int G = 64 / LPR;
  constexpr int TILE = LPR * VEC;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t ntiles = ONE ? 1 : (K + TILE - 1) / TILE;
  const bool kact0 = l * VEC < K;

  float gr[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) gr[i] = 0.f;
  if (kact0) load_vec<VEC>(grow + l * VEC, gr);  // tile 0 stays in registers

  for (int64_t base = s; base < e; base += 64) {
    const int n = (e - base) < 64 ? static_cast<int>(e - base) : 64;
    int64_t c_l = 0;
    float keep = 0.f;  // the dot of this lane's edge, collected step by step
    if (lane < n) c_l = col[base + lane];
    for (int j = 0; j < n; j += G * U) {
      float dot[U];
      if constexpr (ONE) {
        float b[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = j + u * G + g;
          const int64_t c = __shfl(static_cast<long long>(c_l), idx);
#pragma unroll
          for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
          if (idx < n && kact0) load_vec<VEC>(mat + c * K + l * VEC, b[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          dot[u] = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) dot[u] += b[u][i] * gr[i];
        }
      } else {
        int64_t c[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = j + u * G + g;
          c[u] = __shfl(static_cast<long long>(c_l), idx);
          ok[u] = idx < n;
          dot[u] = 0.f;
        }
        for (int64_t t = 0; t < ntiles; ++t) {
          const int64_t k0 = t * TILE + l * VEC;
          const bool kact = k0 < K;
          float gt[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) gt[i] = gr[i];
          if (t > 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) gt[i] = 0.f;
            if (kact) load_vec<VEC>(grow + k0, gt);
          }
          float b[U][VEC];
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
            if (ok[u] && kact) load_vec<VEC>(mat + c[u] * K + k0, b[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) dot[u] += b[u][i] * gt[i];
          }
        }
      }
      // Fold the U partial dots of this lane group together (lane_fold.h: DPP / permlane moves in the
      // VALU, no ds_bpermute): lane l < U of group g ends up with the whole dot of edge slot j + l * G + g.
      psa::fold_group_dots<LPR, U, true>(dot, l);  // bits 16 / 32 by permlane swaps: this kernel waits on memory, not on the VALU
      // It is handed to the lane that loaded that edge (lane == slot) and stored once per 64-edge batch —
      // 256 contiguous bytes instead of G * U floats per step (partial-line writes: 199 MB written for 80 MB
      // of grad_value at config 3, profiles/r03_pmc_backward.json).
      const unsigned rel = static_cast<unsigned>(lane - j);  // this lane's edge belongs to the step iff rel < G * U
      const float got = __shfl(dot[0], static_cast<int>(((rel % G) * LPR + rel / G) & 63u));
      if (rel < static_cast<unsigned>(G * U)) keep = got;
    }
    if (lane < n) __builtin_nontemporal_store(keep / denom, out + base + lane);  // written once: keep it out of the caches
  }
}

template <int VEC, int LPR, int U, bool ONE>
__global__ void __launch_bounds__(kThreads)
spmm_value_bw_kernel(const int64_t* __restrict__ rowptr,
                     const int64_t* __restrict__ col,
                     const float* __restrict__ mat,
                     const float* __restrict__ grad, float* __restrict__ out,
                     int64_t M, int64_t K, int mean,
                     unsigned long long* __restrict__ long_ctr,
                     psa::LongEntry* __restrict__ long_list) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  if (row >= M) return;
  const int64_t s = rowptr[row];
  const int64_t e = rowptr[row + 1];
  if (long_list && e - s > psa::kLongRow) {  // wave-uniform: hand the row to chunk waves
    if (lane == 0) psa::push_long_row(long_ctr, long_list, row, e - s);
    return;
  }
  const float denom = (mean && e - s > 1) ? static_cast<float>(e - s) : 1.0f;
  value_bw_range<VEC, LPR, U, ONE>(col, mat, grad + row * K, out, K, s, e, denom, lane);
}

// One wave per 128-edge chunk of a long row; chunks write disjoint out[e], so
// no combine step is needed.
template <int VEC, int LPR, int U, bool ONE>
__global__ void __launch_bounds__(psa::kLongThreads)
spmm_value_bw_long_kernel(const int64_t* __restrict__ rowptr,
                          const int64_t* __restrict__ col,
                          const float* __restrict__ mat,
                          const float* __restrict__ grad, float* __restrict__ out,
                          int64_t K, int mean,
                          const unsigned long long* __restrict__ long_ctr,
                          const psa::LongEntry* __restrict__ long_list) {
  const int lane = threadIdx.x & 63;
  const unsigned long long ctr = *long_ctr;
  const uint32_t total = static_cast<uint32_t>(ctr & 0xffffffffull);
  const int nrows = static_cast<int>(ctr >> 32);
  const uint32_t num_waves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c < total; c += num_waves) {
    const psa::LongEntry ent = psa::find_long_entry(long_list, nrows, c);
    const int64_t rs = rowptr[ent.row], re = rowptr[ent.row + 1];
    const int64_t s = rs + static_cast<int64_t>(c - ent.first_chunk) * psa::kLongChunk;
    const int64_t e = s + psa::kLongChunk < re ? s + psa::kLongChunk : re;
    const float denom = mean ? static_cast<float>(re - rs) : 1.0f;
    value_bw_range<VEC, LPR, U, ONE>(col, mat, grad + ent.row * K, out, K, s, e, denom, lane);
  }
}

__global__ void __launch_bounds__(kThreads)
transpose_weights_kernel(const float* __restrict__ value,
                         const int64_t* __restrict__ csr2csc,
                         const int64_t* __restrict__ row_csc,
                         const int64_t* __restrict__ rowptr, int64_t nnz,
                         int mean, float* __restrict__ out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (j >= nnz) return;
  float w = value ? value[csr2csc[j]] : 1.f;
  if (mean) {
    const int64_t r = row_csc[j];
    const int64_t deg = rowptr[r + 1] - rowptr[r];
    w = w / static_cast<float>(deg > 0 ? deg : 1);
  }
  out[j] = w;
}

__global__ void __launch_bounds__(kThreads)
spmm_minmax_bw_kernel(const int64_t* __restrict__ col,
                      const float* __restrict__ value,
                      const float* __restrict__ mat,
                      const float* __restrict__ grad,
                      const int64_t* __restrict__ arg_out, int64_t MK, int64_t K,
                      int64_t nnz, float* __restrict__ grad_value,
                      float* __restrict__ grad_mat) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= MK) return;
  const int64_t e = arg_out[t];
  if (e == nnz) return;  // empty row sentinel
  const int64_t k = t % K;
  const float g = grad[t];
  const int64_t c = col[e];
  if (grad_value) atomicAdd(grad_value + e, mat[c * K + k] * g);
  if (grad_mat) atomicAdd(grad_mat + c * K + k, (value ? value[e] : 1.f) * g);
}

// Row-owned form of the same backward (K <= 64*KT): one wave per output row.
// Every winning edge of row i belongs to row i only, so grad_value needs no
// atomics: the wave walks its DISTINCT winners (ballot loop), folds the
// contributions of the lanes that selected the same edge with shuffles and
// stores gV[e] once (deterministic).  grad_mat still needs atomics (different
// rows hit the same column) but now one atomic per (row, k) issued from
// coalesced reads of arg_out / grad.  grad_value must be zero-filled before.
template <int KT>
__global__ void __launch_bounds__(kThreads)
spmm_minmax_bw_row_kernel(const int64_t* __restrict__ col,
                          const float* __restrict__ value,
                          const float* __restrict__ mat,
                          const float* __restrict__ grad,
                          const int64_t* __restrict__ arg_out, int64_t M, int64_t K,
                          int64_t nnz, float* __restrict__ grad_value,
                          float* __restrict__ grad_mat) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  if (row >= M) return;
  int64_t a[KT];
  float x[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int64_t k = lane + 64 * t;
    a[t] = -1;
    x[t] = 0.f;
    if (k < K) {
      const int64_t e = arg_out[row * K + k];
      if (e != nnz) {
        const int64_t c = col[e];
        const float g = grad[row * K + k];
        a[t] = e;
        if (grad_value) x[t] = mat[c * K + k] * g;
        if (grad_mat) atomicAdd(grad_mat + c * K + k, (value ? value[e] : 1.f) * g);
      }
    }
  }
  if (!grad_value) return;
  unsigned long long pend[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) pend[t] = __ballot(a[t] >= 0);
#pragma unroll
  for (int t0 = 0; t0 < KT; ++t0) {
    while (pend[t0]) {
      const int src = __ffsll(static_cast<long long>(pend[t0])) - 1;
      const int64_t e0 = __shfl(static_cast<long long>(a[t0]), src);
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const bool hit = a[t] == e0;
        v += hit ? x[t] : 0.f;
        pend[t] &= ~__ballot(hit);
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) grad_value[e0] = v;
    }
  }
}

template <int VEC, int LPR, int U>
int launch_value_bw(const int64_t* rowptr, const int64_t* col, const float* mat,
                    const float* grad, float* out, int64_t M, int64_t K,
                    int mean, unsigned long long* ctr, psa::LongEntry* list,
                    hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, kWaves);
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  if (K <= static_cast<int64_t>(LPR) * VEC) {  // one tile
    hipLaunchKernelGGL((spmm_value_bw_kernel<VEC, LPR, U, true>),
                       dim3(static_cast<unsigned>(gx)), dim3(kThreads), 0, s,
                       rowptr, col, mat, grad, out, M, K, mean, ctr, list);
    if (list) {
      hipLaunchKernelGGL((spmm_value_bw_long_kernel<VEC, LPR, U, true>), dim3(psa::kLongBlocks),
                         dim3(psa::kLongThreads), 0, s, rowptr, col, mat, grad, out, K, mean, ctr, list);
    }
  } else {
    hipLaunchKernelGGL((spmm_value_bw_kernel<VEC, LPR, U, false>),
                       dim3(static_cast<unsigned>(gx)), dim3(kThreads), 0, s,
                       rowptr, col, mat, grad, out, M, K, mean, ctr, list);
    if (list) {
      hipLaunchKernelGGL((spmm_value_bw_long_kernel<VEC, LPR, U, false>), dim3(psa::kLongBlocks),
                         dim3(psa::kLongThreads), 0, s, rowptr, col, mat, grad, out, K, mean, ctr, list);
    }
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // namespace

extern "C" {

size_t psa_spmm_value_bw_workspace_bytes(int64_t nnz) {
  return nnz > psa::kLongRow ? psa::long_list_bytes(nnz) : 0;
}

int psa_spmm_value_bw(int reduce, const int64_t* rowptr, const int64_t* col,
                      const float* mat, const float* grad, int64_t M, int64_t K,
                      int64_t nnz, float* out, void* workspace,
                      size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(reduce == PSA_SUM || reduce == PSA_MEAN, "reduce must be sum or mean");
  PSA_REQUIRE(M >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (nnz == 0) return PSA_OK;
  PSA_REQUIRE(out != nullptr, "out is NULL");
  hipStream_t s = psa::as_stream(stream);
  if (M == 0 || K == 0) {
    PSA_ZERO(out, sizeof(float) * nnz, s);
    return PSA_OK;
  }
  PSA_REQUIRE(rowptr && col && mat && grad, "NULL pointer");
  const int mean = reduce == PSA_MEAN;
  // long rows: handed to chunk waves when the caller brings the work list
  unsigned long long* ctr = nullptr;
  psa::LongEntry* list = nullptr;
  if (workspace != nullptr && nnz > psa::kLongRow) {
    if (workspace_bytes < psa::long_list_bytes(nnz)) {
      psa::set_error("psa_spmm_value_bw: workspace too small");
      return PSA_ERR_WORKSPACE;
    }
    PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
    ctr = static_cast<unsigned long long*>(workspace);
    list = reinterpret_cast<psa::LongEntry*>(static_cast<char*>(workspace) + 256);
    PSA_ZERO(ctr, 8, s);
  }
#define PSA_VBW(VEC, LPR, U) \
  return launch_value_bw<VEC, LPR, U>(rowptr, col, mat, grad, out, M, K, mean, ctr, list, s)
  const bool v4 = (K % 4 == 0) && psa::aligned(mat, 16) && psa::aligned(grad, 16);
  if (v4) {
    const int64_t q = K / 4;
    if (q <= 4) PSA_VBW(4, 4, 1);
    // 16 edges per step at every width (the single-tile form has the registers for it): K = 32 0.71 -> 0.65 ms at config-3 size
    if (q <= 8) PSA_VBW(4, 8, 2);
    if (q <= 16) PSA_VBW(4, 16, 4);
    // K = 128: 16 gathers in flight per wave (the single-tile form needs 37 VGPRs at U = 4, so 8 waves per SIMD stay):
    // 1.85 -> 1.62 ms at config 3 (profiles/r04_fold_ab.txt)
    if (q <= 32) PSA_VBW(4, 32, 8);
    PSA_VBW(4, 64, 8);
  }
  if (K <= 4) PSA_VBW(1, 4, 1);
  if (K <= 16) PSA_VBW(1, 16, 2);
  PSA_VBW(1, 64, 8);
#undef PSA_VBW
}

int psa_transpose_weights(const float* value, const int64_t* csr2csc,
                          const int64_t* row_csc, const int64_t* rowptr,
                          int64_t nnz, int mean, float* out,
                          psa_stream_t stream) {
  PSA_REQUIRE(nnz >= 0, "negative size");
  if (nnz == 0) return PSA_OK;
  PSA_REQUIRE(out != nullptr, "out is NULL");
  PSA_REQUIRE(value == nullptr || csr2csc != nullptr, "csr2csc needed with value");
  PSA_REQUIRE(!mean || (row_csc && rowptr), "row_csc/rowptr needed for mean");
  const int64_t blocks = psa::ceil_div(nnz, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "nnz too large for one launch");
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, psa::as_stream(stream), value, csr2csc,
                     row_csc, rowptr, nnz, mean, out);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_spmm_minmax_bw(const int64_t* col, const float* value, const float* mat,
                       const float* grad, const int64_t* arg_out, int64_t M,
                       int64_t N, int64_t K, int64_t nnz, float* grad_value,
                       float* grad_mat, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  hipStream_t s = psa::as_stream(stream);
  if (grad_value && nnz) PSA_ZERO(grad_value, sizeof(float) * nnz, s);
  if (grad_mat && N * K) PSA_ZERO(grad_mat, sizeof(float) * N * K, s);
  if (M * K == 0 || nnz == 0 || (!grad_value && !grad_mat)) return PSA_OK;
  PSA_REQUIRE(col && grad && arg_out, "NULL pointer");
  PSA_REQUIRE(!grad_value || mat, "mat needed for grad_value");
  if (K <= 256) {  // row-owned kernel: no atomics on grad_value
    const int64_t gx = psa::ceil_div(M, kWaves);
    PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
    const dim3 grid(static_cast<unsigned>(gx)), block(kThreads);
#define PSA_ROWBW(KT)                                                                  \
  hipLaunchKernelGGL((spmm_minmax_bw_row_kernel<KT>), grid, block, 0, s, col, value, \
                     mat, grad, arg_out, M, K, nnz, grad_value, grad_mat)
    if (K <= 64) PSA_ROWBW(1);
    else if (K <= 128) PSA_ROWBW(2);
    else PSA_ROWBW(4);
#undef PSA_ROWBW
    PSA_LAUNCH_CHECK();
    return PSA_OK;
  }
  const int64_t blocks = psa::ceil_div(M * K, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "M*K too large for one launch");
  hipLaunchKernelGGL(spmm_minmax_bw_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, col, value, mat, grad, arg_out, M * K,
                     K, nnz, grad_value, grad_mat);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
