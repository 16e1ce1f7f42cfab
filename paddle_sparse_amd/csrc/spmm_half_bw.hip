// Passes over the CSC view with HALF-WIDTH dense operands (fp16 / bf16 mat, grad and grad_mat; fp32 sums): the backward
// of spmm_{sum,mean} with trained values and of spmm_{min,max}, both gradients in one pass.  gfx950.  The forward is
// spmm_half.hip; shared helpers in half_util.h.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"
#include "half_util.h"
#include "lane_fold.h"
#include "long_rows.h"
#include "vec_io.h"

namespace {

using namespace psa_half;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

using psa::shfl_i64;

// ---- backward over the CSC view with half-width operands, trained edge values ---------------------
// The fp32 pass (spmm.hip, M_CSC) with 2-byte dense operands: one wave per column c of the CSC
// view; for every stored entry (r, c), in CSC order,
//   grad_mat[c, :] += w * grad[r, :]                  (fp32 sums, one rounding on store)
//   grad_value_csc[j] = s_r * <mat[c, :], grad[r, :]>  (fp32 out; s_r = 1 / deg(r) for mean, else 1)
// w = weights in CSC ORDER (fp32: value[csr2csc] along the planned route, times s_r for mean is done
// here), so the pass reads two streams (row ids, weights), gathers 2 K bytes per entry and writes
// 2 K per column: half the bytes of the fp32 pass, and no fp32 copies of mat / grad (VERDICT r02
// #9a: the widening route wrote and re-read 2 x 4 N K + 4 M K bytes per step before it even started).
// The whole K must sit in ONE tile (the dot needs every column): K <= 512.
// MW (min / max): bytes per entry of the forward's row-local arg_out `words` [M, K] and of the per-entry
// `tags` [nnz] (CSC order); an entry's term counts for column k only where words[r, k] == its tag — the
// masked form of the fp32 pass (spmm.hip, M_MASK) in its exact forms (1 byte: no row above 128 entries;
// 2 bytes: none above 65 535).  MW = 0: sum / mean, every term counts.
typedef float F2 __attribute__((ext_vector_type(2)));

// Entries [s, e) of ONE column of the CSC view (all of it for a row wave, a 128-entry chunk for a chunk wave):
// acc += w * grad[r, k0 .. k0 + 7] (masked for MW != 0), grad_value[j] = <mat[c, :], grad[r, :]> stored per
// 64-entry batch.  mr: the lane's slice of mat[c, :] (GV only).
template <typename T, int LPR, int U, bool GV, int MW, bool SMALL>
__device__ __forceinline__ void half_csc_bw_range(const int64_t* __restrict__ row_csc, const float* __restrict__ w_csc,
                                                  const float* __restrict__ row_scale, const uint16_t* __restrict__ grad,
                                                  float* __restrict__ grad_value, int64_t K, int64_t k0, bool kact,
                                                  int lane, int64_t s, int64_t e, const uint8_t* __restrict__ words,
                                                  const uint8_t* __restrict__ tags, const F2 (&mr)[4], F2 (&acc)[4]) {
  constexpr int G = 64 / LPR;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int g = lane / LPR;
  const int l = lane % LPR;
  const uint16_t* gk = grad + k0;
  for (int64_t base = s; base < e; base += 64) {
    const int n = (e - base) < 64 ? static_cast<int>(e - base) : 64;
    int64_t r_l = 0;
    float v_l = 1.f, s_l = 1.f, gv_keep = 0.f;
    uint32_t t_l = 0;
    if (lane < n) {
      r_l = row_csc[base + lane];
      if (w_csc != nullptr) v_l = w_csc[base + lane];
      if (row_scale != nullptr) s_l = row_scale[r_l];
      if constexpr (MW == 1) t_l = tags[base + lane];
      if constexpr (MW == 2) t_l = reinterpret_cast<const uint16_t*>(tags)[base + lane];
    }
    for (int j = 0; j < n; j += G * U) {
      uint4 raw[U];
      using Words = typename std::conditional<MW == 2, uint4, uint2>::type;  // the 8 entries of words[r, k0 ..]: 8 or 16 bytes
      Words wd[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * G + g;  // < 64
        ok[u] = (idx < n) && kact;
        raw[u] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (MW != 0) wd[u] = {};
        if constexpr (SMALL) {  // (see spmm_half_row_kernel)
          const uint32_t r = static_cast<uint32_t>(__shfl(static_cast<int>(r_l), idx));
          const uint32_t eo = r * static_cast<uint32_t>(K) + static_cast<uint32_t>(k0);
          if (ok[u]) raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(grad) + eo * 2u);
          if constexpr (MW != 0) {
            if (ok[u]) wd[u] = *reinterpret_cast<const Words*>(words + eo * static_cast<uint32_t>(MW));
          }
        } else {
          const int64_t r = shfl_i64(r_l, idx);
          if (ok[u]) raw[u] = *reinterpret_cast<const uint4*>(gk + r * K);
          if constexpr (MW != 0) {
            if (ok[u]) wd[u] = *reinterpret_cast<const Words*>(words + (r * K + k0) * MW);
          }
        }
      }
      float dot[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * G + g;
        float w = __shfl(v_l, idx);  // fetched after the gathers are out
        if (row_scale != nullptr) w *= __shfl(s_l, idx);
        float b[8];
        widen8<T>(raw[u], b);
        if constexpr (MW != 0) {  // only where the forward named this entry the winner
          const uint32_t tag = static_cast<uint32_t>(__shfl(static_cast<int>(t_l), idx));
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            uint32_t f;
            if constexpr (MW == 2) {
              const uint32_t ws[4] = {wd[u].x, wd[u].y, wd[u].z, wd[u].w};
              f = (ws[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            } else {
              f = ((i < 4 ? wd[u].x : wd[u].y) >> (8 * (i & 3))) & 0xffu;
            }
            if (f != tag) b[i] = 0.f;
          }
        }
        const F2 w2 = F2{w, w};
        F2 d2 = F2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const F2 b2 = F2{b[2 * i], b[2 * i + 1]};
          acc[i] = __builtin_elementwise_fma(w2, b2, acc[i]);  // a masked slot adds w * 0
          if constexpr (GV) d2 = __builtin_elementwise_fma(b2, mr[i], d2);
        }
        dot[u] = d2[0] + d2[1];
      }
      if constexpr (GV) {
        // fold the U partial dots of the lane group transposing as it goes (lane_fold.h: DPP operands of the adds,
        // ds_bpermute for 16 / 32): lane l < U of group g then holds the dot of edge slot j + l * G + g; the lane that
        // loaded that edge (lane == slot) keeps it for one 256-byte store per 64-edge batch
        psa::fold_group_dots<LPR, U>(dot, l);
        if constexpr (MW != 0) {
          // masked form (94 VGPRs, 5 waves per SIMD): the lanes that hold the dots store them themselves — 4-byte
          // stores in G * U pieces per step instead of one 256-byte store per batch, but no hand-over shuffle and
          // no value kept across the steps: spmm_max bf16 step 3.39 -> 3.20 ms (the sum form gains nothing from it)
          if (l < U) {
            const int idx = j + l * G + g;
            if (idx < n) grad_value[base + idx] = dot[0];
          }
        } else {
          const unsigned rel = static_cast<unsigned>(lane - j);
          const float got = __shfl(dot[0], static_cast<int>(((rel % G) * LPR + rel / G) & 63u));
          if (rel < static_cast<unsigned>(G * U)) gv_keep = got;
        }
      }
    }
    if (GV && MW == 0 && lane < n) __builtin_nontemporal_store(gv_keep * s_l, grad_value + base + lane);
  }
}

template <typename T, bool GV>
__device__ __forceinline__ void half_load_mat_row(const uint16_t* __restrict__ mat, int64_t c, int64_t K, int64_t k0,
                                                  bool kact, F2 (&mr)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) mr[i] = F2{0.f, 0.f};
  if constexpr (GV) {
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
    if (kact) raw = *reinterpret_cast<const uint4*>(mat + c * K + k0);
    float m8[8];
    widen8<T>(raw, m8);
#pragma unroll
    for (int i = 0; i < 4; ++i) mr[i] = F2{m8[2 * i], m8[2 * i + 1]};
  }
}

// One wave per column.  FUSED (the caller brought the long-column workspace): columns above psa::kLongRow entries — a hub
// row of a power-law graph is a 40 000-entry column of the CSC view, which one wave would walk for milliseconds — are
// listed by a pre-pass over colptr (psa::find_long_rows_kernel) and the launch runs TWO ROLES, as the fp32 passes do
// (spmm.hip): the first `chunk_blocks` workgroups take 128-entry chunks of the long columns (grid-stride; a chunk's fp32
// partial of grad_mat[c, :] goes to part[chunk, :], its grad_value entries are final: chunks own disjoint entries), all
// the others take ordinary columns and skip the long ones.  On R-MAT 21 60 % of the entries sit in long columns: chunk
// waves bound by HBM and column waves bound by per-column latency then overlap instead of running back to back
// (three launches in a row: 1.55 ms for the sum pass).
template <typename T, int LPR, int U, bool GV, int MW = 0, bool SMALL = false, bool FUSED = false>
__global__ void __launch_bounds__(kThreads, (MW != 0 && GV) ? 5 : 7)  // sum + grad_value: 70-73 VGPRs unconstrained, 72 fit 7 waves per SIMD; masked + grad_value: 5
spmm_half_csc_bw_kernel(const int64_t* __restrict__ colptr, const int64_t* __restrict__ row_csc,
                        const float* __restrict__ w_csc, const float* __restrict__ row_scale,
                        const uint16_t* __restrict__ mat, const uint16_t* __restrict__ grad,
                        uint16_t* __restrict__ grad_mat, float* __restrict__ grad_value, int64_t N, int64_t K,
                        int mix_xcds, const uint8_t* __restrict__ words = nullptr,
                        const uint8_t* __restrict__ tags = nullptr,
                        const unsigned long long* __restrict__ long_ctr = nullptr,
                        const psa::LongEntry* __restrict__ long_list = nullptr, float* __restrict__ part = nullptr,
                        int chunk_blocks = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = static_cast<int64_t>(l) * 8;
  const bool kact = k0 < K;
  if constexpr (FUSED) {
    if (static_cast<int>(blockIdx.x) < chunk_blocks) {  // chunk role (block-uniform)
      const unsigned long long ctr = *long_ctr;
      const uint32_t total = static_cast<uint32_t>(ctr & 0xffffffffull);
      const int ncols = static_cast<int>(ctr >> 32);
      const uint32_t num_waves = static_cast<uint32_t>(chunk_blocks) * kWaves;
      for (uint32_t ch = blockIdx.x * kWaves + wave; ch < total; ch += num_waves) {
        const psa::LongEntry ent = psa::find_long_entry(long_list, ncols, ch);
        const int64_t cs = colptr[ent.row], ce = colptr[ent.row + 1];
        const int64_t s = cs + static_cast<int64_t>(ch - ent.first_chunk) * psa::kLongChunk;
        const int64_t e = s + psa::kLongChunk < ce ? s + psa::kLongChunk : ce;
        F2 acc[4], mr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = F2{0.f, 0.f};
        half_load_mat_row<T, GV>(mat, ent.row, K, k0, kact, mr);
        half_csc_bw_range<T, LPR, U, GV, MW, SMALL>(row_csc, w_csc, row_scale, grad, grad_value, K, k0, kact, lane, s, e,
                                                     words, tags, mr, acc);
        float a8[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a8[2 * i] = acc[i][0];
          a8[2 * i + 1] = acc[i][1];
        }
        psa::fold_lane_groups<LPR, 8>(a8);
        if (g == 0 && kact) {
          float* dst = part + static_cast<int64_t>(ch) * K + k0;
          *reinterpret_cast<float4*>(dst) = make_float4(a8[0], a8[1], a8[2], a8[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
        }
      }
      return;
    }
  }
  int64_t rb = static_cast<int64_t>(blockIdx.x) - (FUSED ? chunk_blocks : 0);
  if (mix_xcds) rb ^= static_cast<int64_t>((static_cast<uint32_t>(rb >> 3) * 0x9E3779B1u) >> 29);
  const int64_t c = rb * kWaves + wave;
  if (c >= N) return;
  const int64_t s = colptr[c], e = colptr[c + 1];
  if (FUSED && e - s > psa::kLongRow) return;  // wave-uniform: a chunk wave's work (listed by the pre-pass)
  // sums and products as pairs: v_pk_fma_f32 does two fp32 FMAs per issue slot, and a wave64 VALU
  // instruction takes 4 issue cycles on this chip — with ~10 entries per column the pass is bound by
  // instruction issue, not by bytes (profiles/r03_half_train_step.txt)
  F2 acc[4], mr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = F2{0.f, 0.f};
  half_load_mat_row<T, GV>(mat, c, K, k0, kact, mr);
  half_csc_bw_range<T, LPR, U, GV, MW, SMALL>(row_csc, w_csc, row_scale, grad, grad_value, K, k0, kact, lane, s, e, words,
                                               tags, mr, acc);
  float a8[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a8[2 * i] = acc[i][0];
    a8[2 * i + 1] = acc[i][1];
  }
  psa::fold_lane_groups<LPR, 8>(a8);  // the lane groups' sums (lane_fold.h), the bits of the xor shuffles
  if (g == 0 && kact) {
    const uint4 packed = narrow8<T>(a8);
    typedef unsigned int U4 __attribute__((ext_vector_type(4)));
    U4 st;
    st[0] = packed.x;
    st[1] = packed.y;
    st[2] = packed.z;
    st[3] = packed.w;
    __builtin_nontemporal_store(st, reinterpret_cast<U4*>(grad_mat + c * K + k0));
  }
}

// One wave per long column: its chunks' partials summed in chunk order (fp32), one rounding on store.
template <typename T>
__global__ void __launch_bounds__(psa::kLongThreads)
spmm_half_csc_bw_combine_kernel(int64_t K, const unsigned long long* __restrict__ long_ctr,
                                const psa::LongEntry* __restrict__ long_list, const float* __restrict__ part,
                                uint16_t* __restrict__ grad_mat) {
  const int lane = threadIdx.x & 63;
  const int ncols = static_cast<int>(*long_ctr >> 32);
  const int num_waves = gridDim.x * (blockDim.x >> 6);
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < ncols; r += num_waves) {
    const psa::LongEntry ent = long_list[r];
    for (int64_t k0 = static_cast<int64_t>(lane) * 8; k0 < K; k0 += 512) {
      const float* src = part + static_cast<int64_t>(ent.first_chunk) * K + k0;
      float a8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a8[i] = 0.f;
      for (uint32_t ch = 0; ch < ent.num_chunks; ch += 4) {  // four partials requested per step, added in chunk order
        float4 x[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool on = ch + t < ent.num_chunks;
          const float* p = src + static_cast<int64_t>(ch + t) * K;
          x[t][0] = on ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
          x[t][1] = on ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (ch + t >= ent.num_chunks) break;
          a8[0] += x[t][0].x; a8[1] += x[t][0].y; a8[2] += x[t][0].z; a8[3] += x[t][0].w;
          a8[4] += x[t][1].x; a8[5] += x[t][1].y; a8[6] += x[t][1].z; a8[7] += x[t][1].w;
        }
      }
      const uint4 packed = narrow8<T>(a8);
      *reinterpret_cast<uint4*>(grad_mat + ent.row * K + k0) = packed;
    }
  }
}

// scratch of the long-column path: {counter, list of long columns, fp32 partials [chunks, K]}
struct HalfLong {
  unsigned long long* ctr = nullptr;
  psa::LongEntry* list = nullptr;
  float* part = nullptr;
};

size_t half_long_bytes(int64_t K, int64_t nnz) {
  if (K <= 0 || nnz <= psa::kLongRow) return 0;
  return psa::long_list_bytes(nnz) + psa::align256(sizeof(float) * static_cast<size_t>(psa::max_long_chunks(nnz)) * K);
}

constexpr int kHalfChunkBlocks = 768;  // workgroups in the chunk role (x 4 waves; the fp32 passes use 768 as well)

template <typename T, int LPR, int U, bool GV, int MW, bool SMALL>
int launch_half_csc_bw(const int64_t* colptr, const int64_t* row_csc, const float* w_csc, const float* row_scale,
                       const uint16_t* mat, const uint16_t* grad, uint16_t* grad_mat, float* grad_value, int64_t N,
                       int64_t K, hipStream_t s, const uint8_t* words, const uint8_t* tags, const HalfLong& w) {
  const int64_t gx = psa::ceil_div(psa::ceil_div(N, kWaves), 8) * 8;
  PSA_REQUIRE(gx + kHalfChunkBlocks <= 0x7fffffff, "problem too large for one launch");
  if (w.list == nullptr) {
    hipLaunchKernelGGL((spmm_half_csc_bw_kernel<T, LPR, U, GV, MW, SMALL, false>), dim3(static_cast<unsigned>(gx)),
                       dim3(kThreads), 0, s, colptr, row_csc, w_csc, row_scale, mat, grad, grad_mat, grad_value, N, K, 1,
                       words, tags);
  } else {  // columns above 128 entries: list pre-pass, both roles in one launch, then the partials in chunk order
    hipLaunchKernelGGL(psa::find_long_rows_kernel,
                       dim3(static_cast<unsigned>(psa::ceil_div(N, psa::kFindThreads * psa::kFindIters))),
                       dim3(psa::kFindThreads), 0, s, colptr, N, w.ctr, w.list);
    hipLaunchKernelGGL((spmm_half_csc_bw_kernel<T, LPR, U, GV, MW, SMALL, true>),
                       dim3(static_cast<unsigned>(gx + kHalfChunkBlocks)), dim3(kThreads), 0, s, colptr, row_csc, w_csc,
                       row_scale, mat, grad, grad_mat, grad_value, N, K, 1, words, tags, w.ctr, w.list, w.part,
                       kHalfChunkBlocks);
    hipLaunchKernelGGL((spmm_half_csc_bw_combine_kernel<T>), dim3(psa::kLongBlocks), dim3(psa::kLongThreads), 0, s, K,
                       w.ctr, w.list, w.part, grad_mat);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

template <typename T, int MW = 0>
int dispatch_half_csc_bw(const int64_t* colptr, const int64_t* row_csc, const float* w_csc, const float* row_scale,
                         const uint16_t* mat, const uint16_t* grad, uint16_t* grad_mat, float* grad_value, int64_t N,
                         int64_t K, int64_t nnz, hipStream_t s, const uint8_t* words, const uint8_t* tags, bool small,
                         void* workspace, size_t workspace_bytes) {
  HalfLong w;
  if (workspace != nullptr && nnz > psa::kLongRow) {
    if (workspace_bytes < half_long_bytes(K, nnz)) {
      psa::set_error("half-width pass over the CSC view: workspace too small");
      return PSA_ERR_WORKSPACE;
    }
    PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
    w.ctr = static_cast<unsigned long long*>(workspace);
    w.list = reinterpret_cast<psa::LongEntry*>(static_cast<char*>(workspace) + 256);
    w.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + psa::long_list_bytes(nnz));
    PSA_ZERO(w.ctr, 8, s);
  }
  const int64_t q = K / 8;
#define PSA_GO(LPR, U)                                                                                                   \
  do {                                                                                                                   \
    if (grad_value != nullptr && small)                                                                                  \
      return launch_half_csc_bw<T, LPR, U, true, MW, true>(colptr, row_csc, w_csc, row_scale, mat, grad, grad_mat,       \
                                                            grad_value, N, K, s, words, tags, w);                        \
    if (grad_value != nullptr)                                                                                           \
      return launch_half_csc_bw<T, LPR, U, true, MW, false>(colptr, row_csc, w_csc, row_scale, mat, grad, grad_mat,      \
                                                             grad_value, N, K, s, words, tags, w);                       \
    if (small)                                                                                                           \
      return launch_half_csc_bw<T, LPR, U, false, MW, true>(colptr, row_csc, w_csc, row_scale, mat, grad, grad_mat,      \
                                                             grad_value, N, K, s, words, tags, w);                       \
    return launch_half_csc_bw<T, LPR, U, false, MW, false>(colptr, row_csc, w_csc, row_scale, mat, grad, grad_mat,       \
                                                            grad_value, N, K, s, words, tags, w);                        \
  } while (0)
  if (q <= 1) PSA_GO(1, 1);
  if (q <= 2) PSA_GO(2, 2);
  if (q <= 4) PSA_GO(4, 4);
  if (q <= 8) PSA_GO(8, 4);
  if (q <= 16) PSA_GO(16, 4);  // K = 128, config 3 in bf16: 1.28 ms; U = 2 the same (1.29), U = 8 spills (6.8 ms)
  if (q <= 32) PSA_GO(32, 4);
  PSA_GO(64, 8);
#undef PSA_GO
}

}  // namespace

extern "C" int psa_spmm_half_sum_bw_csc(int dtype, const int64_t* colptr, const int64_t* row_csc, const float* weight_csc,
                                        const float* row_scale, const void* mat, const void* grad, int64_t M, int64_t N,
                                        int64_t K, int64_t nnz, float* grad_value_csc, void* grad_mat,
                                        void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (dtype != PSA_F16 && dtype != PSA_BF16) {
    psa::set_error("psa_spmm_half_sum_bw_csc: dtype must be PSA_F16 or PSA_BF16");
    return PSA_ERR_UNSUPPORTED;
  }
  if (K % 8 != 0 || K > 512 || !psa::aligned(mat, 16) || !psa::aligned(grad, 16) || !psa::aligned(grad_mat, 16)) {
    psa::set_error("psa_spmm_half_sum_bw_csc: needs K % 8 == 0, K <= 512 and 16-byte aligned operands");
    return PSA_ERR_UNSUPPORTED;
  }
  if (N == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(colptr && grad_mat, "NULL pointer");
  PSA_REQUIRE(nnz == 0 || (row_csc && grad), "NULL pointer");
  PSA_REQUIRE(grad_value_csc == nullptr || mat != nullptr || nnz == 0, "grad_value needs mat");
  hipStream_t s = psa::as_stream(stream);
  const uint16_t* m = static_cast<const uint16_t*>(mat);
  const uint16_t* g = static_cast<const uint16_t*>(grad);
  uint16_t* gm = static_cast<uint16_t*>(grad_mat);
  if (dtype == PSA_BF16)
    return dispatch_half_csc_bw<BF16>(colptr, row_csc, weight_csc, row_scale, m, g, gm, grad_value_csc, N, K, nnz, s, nullptr,
                                      nullptr, g_half_variant != 4 && M * K * 2 < (1ll << 32), workspace, workspace_bytes);
  return dispatch_half_csc_bw<F16>(colptr, row_csc, weight_csc, row_scale, m, g, gm, grad_value_csc, N, K, nnz, s, nullptr,
                                   nullptr, g_half_variant != 4 && M * K * 2 < (1ll << 32), workspace, workspace_bytes);
}

extern "C" size_t psa_spmm_half_bw_csc_workspace_bytes(int64_t K, int64_t nnz) { return half_long_bytes(K, nnz); }

extern "C" int psa_spmm_half_minmax_bw_csc(int dtype, const int64_t* colptr, const int64_t* row_csc, const void* tag,
                                           const float* weight_csc, const void* mat, const void* grad,
                                           const void* arg_bytes, int arg_width, int64_t M, int64_t N, int64_t K,
                                           int64_t nnz, float* grad_value_csc, void* grad_mat, void* workspace,
                                           size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (dtype != PSA_F16 && dtype != PSA_BF16) {
    psa::set_error("psa_spmm_half_minmax_bw_csc: dtype must be PSA_F16 or PSA_BF16");
    return PSA_ERR_UNSUPPORTED;
  }
  if (K % 8 != 0 || K > 512 || !psa::aligned(mat, 16) || !psa::aligned(grad, 16) || !psa::aligned(grad_mat, 16)) {
    psa::set_error("psa_spmm_half_minmax_bw_csc: needs K % 8 == 0, K <= 512 and 16-byte aligned operands");
    return PSA_ERR_UNSUPPORTED;
  }
  PSA_REQUIRE(arg_width == 1 || arg_width == 2, "arg_width must be 1 or 2");
  if (N == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(colptr && grad_mat, "NULL pointer");
  PSA_REQUIRE(nnz == 0 || (row_csc && grad && tag && arg_bytes), "NULL pointer");
  PSA_REQUIRE(arg_bytes == nullptr || psa::aligned(arg_bytes, 8 * arg_width), "arg_bytes alignment");
  PSA_REQUIRE(grad_value_csc == nullptr || mat != nullptr || nnz == 0, "grad_value needs mat");
  hipStream_t s = psa::as_stream(stream);
  const uint16_t* m = static_cast<const uint16_t*>(mat);
  const uint16_t* g = static_cast<const uint16_t*>(grad);
  uint16_t* gm = static_cast<uint16_t*>(grad_mat);
  const uint8_t* words = static_cast<const uint8_t*>(arg_bytes);
  const uint8_t* tags = static_cast<const uint8_t*>(tag);
#define PSA_MM(T, MW)                                                                                                  \
  return dispatch_half_csc_bw<T, MW>(colptr, row_csc, weight_csc, nullptr, m, g, gm, grad_value_csc, N, K, nnz, s, words, \
                                     tags, g_half_variant != 4 && M * K * 2 < (1ll << 32), workspace, workspace_bytes)
  if (dtype == PSA_BF16) {
    if (arg_width == 2) PSA_MM(BF16, 2);
    PSA_MM(BF16, 1);
  }
  if (arg_width == 2) PSA_MM(F16, 2);
  PSA_MM(F16, 1);
#undef PSA_MM
}
