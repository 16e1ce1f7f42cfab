// CSR x dense SpMM forward (sum / mean / min / max), fp32, gfx950.
//
// Not present in the reference (README.md:47-50); semantics are upstream
// pytorch_sparse spmm, restated in oracle/spmm_oracle.c.
//
// The op is a pure HBM gather: 2*K flop per (12 + 4K) bytes, so no MFMA.
// What the kernels are built around:
//   * one wavefront owns one CSR row (or a run of rows) and is the only
//     writer of that output row: no atomics, deterministic sums;
//   * the row's col/value entries are read ONCE, coalesced, one per lane,
//     and handed to the gather loop by cross-lane moves;
//   * every gather of a dense row is one contiguous, 16-B-per-lane load
//     (LPR lanes x float4 cover the K-tile; 64/LPR edges share a wave
//     instruction), several of them in flight before the first use;
//   * partial sums of the 64/LPR edge slots are folded with wave shuffles.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<2> {
  using T = float2;
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

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&src)[VEC]) {
  using T = typename Vec<VEC>::T;
  T v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = src[i];
  *reinterpret_cast<T*>(p) = v;
}

__device__ __forceinline__ int64_t shfl_i64(int64_t x, int src) {
  return __shfl(static_cast<long long>(x), src);
}

// ---------------------------------------------------------------------------
// Variant A: one wavefront per CSR row.
//   LPR lanes x VEC floats cover one K-tile (blockIdx.y selects the tile);
//   G = 64/LPR edge slots work on G edges of the row per step; U steps are
//   issued before any is consumed (G*U gathers in flight per wave).
// ---------------------------------------------------------------------------
template <int VEC, int LPR, int RED, int U>
__global__ void __launch_bounds__(kThreads)
spmm_row_kernel(const int64_t* __restrict__ rowptr,
                const int64_t* __restrict__ col,
                const float* __restrict__ val, const float* __restrict__ mat,
                float* __restrict__ out, int64_t* __restrict__ arg_out,
                int64_t M, int64_t K, int64_t nnz, int mean) {
  constexpr int G = 64 / LPR;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  if (row >= M) return;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = static_cast<int64_t>(blockIdx.y) * (LPR * VEC) + l * VEC;
  const bool kact = k0 < K;  // K % VEC == 0 (dispatch guarantees it)
  const float* matk = mat + k0;

  const int64_t s = rowptr[row];
  const int64_t e = rowptr[row + 1];

  float acc[VEC];
  int64_t arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc[i] = RED == R_SUM ? 0.f
                          : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
    arg[i] = nnz;
  }

  for (int64_t base = s; base < e; base += 64) {
    const int n = (e - base) < 64 ? static_cast<int>(e - base) : 64;
    int64_t c_l = 0;
    float v_l = 0.f;
    if (lane < n) {
      c_l = col[base + lane];
      v_l = val ? val[base + lane] : 1.f;
    }
    for (int j = 0; j < n; j += G * U) {
      float b[U][VEC];
      float w[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * G + g;  // < 64 by the static_assert
        const int64_t c = shfl_i64(c_l, idx);
        w[u] = __shfl(v_l, idx);
        ok[u] = (idx < n) && kact;
#pragma unroll
        for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
        if (ok[u]) load_vec<VEC>(matk + c * K, b[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (RED == R_SUM) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += w[u] * b[u][i];
        } else if (ok[u]) {
          const int64_t eid = base + j + u * G + g;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float x = w[u] * b[u][i];
            const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
            if (better) {
              acc[i] = x;
              arg[i] = eid;
            }
          }
        }
      }
    }
  }

  // Fold the G edge slots.
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float o = __shfl_xor(acc[i], off);
      if (RED == R_SUM) {
        acc[i] += o;
      } else {
        const int64_t oa = shfl_i64(arg[i], lane ^ off);
        // first winner in edge order: ties go to the smaller edge id
        const bool better = RED == R_MAX ? (o > acc[i]) : (o < acc[i]);
        if (better || (o == acc[i] && oa < arg[i])) {
          acc[i] = o;
          arg[i] = oa;
        }
      }
    }
  }

  if (g == 0 && kact) {
    const int64_t deg = e - s;
    if (RED == R_SUM) {
      if (mean && deg > 1) {
        const float d = static_cast<float>(deg);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = acc[i] / d;
      }
    } else {
      if (deg == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) arg_out[row * K + k0 + i] = arg[i];
    }
    store_vec<VEC>(out + row * K + k0, acc);
  }
}

// ---------------------------------------------------------------------------
// Variant B (narrow K, K <= LPR*VEC <= 64 floats): G = 64/LPR ROWS per wave.
// A dense row is only LPR*16 bytes, so one row per wave leaves most lanes in
// the cross-group fold and the per-row prologue dominates.  Here every lane
// group owns a whole CSR row: it walks its edges in order (U gathers in flight),
// needs no cross-lane traffic at all, and the sum runs in exact edge order.
// Groups of one wave finish at different trip counts (exec-masked loop).
// ---------------------------------------------------------------------------
template <int VEC, int LPR, int RED, int U>
__global__ void __launch_bounds__(kThreads)
spmm_multirow_kernel(const int64_t* __restrict__ rowptr,
                     const int64_t* __restrict__ col,
                     const float* __restrict__ val, const float* __restrict__ mat,
                     float* __restrict__ out, int64_t* __restrict__ arg_out,
                     int64_t M, int64_t K, int64_t nnz, int mean) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t row =
      (static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6)) * G + g;
  const int64_t k0 = l * VEC;
  if (row >= M || k0 >= K) return;  // no wave-level operation below
  const float* matk = mat + k0;
  const int64_t s = rowptr[row];
  const int64_t e = rowptr[row + 1];
  float acc[VEC];
  int64_t arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc[i] = RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
    arg[i] = nnz;
  }
  for (int64_t p = s; p < e; p += U) {
    float b[U][VEC];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w[u] = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
      if (p + u < e) {
        const int64_t c = col[p + u];
        w[u] = val ? val[p + u] : 1.f;
        load_vec<VEC>(matk + c * K, b[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (RED == R_SUM) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += w[u] * b[u][i];
      } else if (p + u < e) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float x = w[u] * b[u][i];
          const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
          if (better) {
            acc[i] = x;
            arg[i] = p + u;
          }
        }
      }
    }
  }
  const int64_t deg = e - s;
  if (RED == R_SUM) {
    if (mean && deg > 1) {
      const float d = static_cast<float>(deg);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = acc[i] / d;
    }
  } else {
    if (deg == 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) arg_out[row * K + k0 + i] = arg[i];
  }
  store_vec<VEC>(out + row * K + k0, acc);
}

int g_variant = 0;

template <int VEC, int LPR, int U>
int launch_multirow(int red, const int64_t* rowptr, const int64_t* col,
                    const float* val, const float* mat, float* out,
                    int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
                    hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, static_cast<int64_t>(kWaves) * (64 / LPR));
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx)), block(kThreads);
  if (red == R_SUM) {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_SUM, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean);
  } else if (red == R_MIN) {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_MIN, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean);
  } else {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_MAX, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

template <int VEC, int LPR, int U>
int launch_row(int red, const int64_t* rowptr, const int64_t* col,
               const float* val, const float* mat, float* out,
               int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
               hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, kWaves);
  const int64_t gy = psa::ceil_div(K, static_cast<int64_t>(LPR) * VEC);
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  PSA_REQUIRE(gy <= 65535, "K too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  const dim3 block(kThreads);
  if (red == R_SUM) {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_SUM, U>), grid, block, 0,
                       s, rowptr, col, val, mat, out, arg_out, M, K, nnz,
                       mean);
  } else if (red == R_MIN) {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_MIN, U>), grid, block, 0,
                       s, rowptr, col, val, mat, out, arg_out, M, K, nnz,
                       mean);
  } else {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_MAX, U>), grid, block, 0,
                       s, rowptr, col, val, mat, out, arg_out, M, K, nnz,
                       mean);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // namespace

extern "C" {

int psa_spmm_set_variant(int variant) {
  const int prev = g_variant;
  g_variant = variant;
  return prev;
}

int psa_spmm(int reduce, const int64_t* rowptr, const int64_t* col,
             const float* value, const float* mat, int64_t M, int64_t N,
             int64_t K, int64_t nnz, float* out, int64_t* arg_out,
             psa_stream_t stream) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (M == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(rowptr != nullptr, "rowptr is NULL");
  PSA_REQUIRE(out != nullptr, "out is NULL");
  PSA_REQUIRE(nnz == 0 || (col != nullptr && mat != nullptr),
              "col/mat is NULL");
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  PSA_REQUIRE(!minmax || arg_out != nullptr, "arg_out required for min/max");
  hipStream_t s = psa::as_stream(stream);
  const int red = reduce == PSA_MIN ? R_MIN : (reduce == PSA_MAX ? R_MAX : R_SUM);
  const int mean = reduce == PSA_MEAN;

#define PSA_ROW(VEC, LPR, U)                                                 \
  return launch_row<VEC, LPR, U>(red, rowptr, col, value, mat, out, arg_out, \
                                 M, K, nnz, mean, s)

  const bool v4 = (K % 4 == 0) && psa::aligned(mat, 16) && psa::aligned(out, 16);
  if (v4) {
    const int64_t q = K / 4;  // float4 per row
    if (g_variant == 2 && K % 128 == 0 && psa::aligned(mat, 8)) PSA_ROW(2, 64, 8);
    if (g_variant == 3 && q >= 32) PSA_ROW(4, 32, 8);
    if (g_variant == 4 && K % 128 == 0) PSA_ROW(2, 64, 16);
#define PSA_MULTI(VEC, LPR, U)                                                     \
  return launch_multirow<VEC, LPR, U>(red, rowptr, col, value, mat, out, arg_out, \
                                      M, K, nnz, mean, s)
    // K <= 64: several rows per wave (multirow, 8 gathers in flight per row)
    // measured at 2M rows / 20M edges: K=16 0.58 -> 0.41 ms, K=32 0.69 -> 0.45,
    // K=64 1.01 -> 0.88 (variant 1 forces the one-row-per-wave kernel back)
    if (g_variant != 1) {
      if (q <= 4) PSA_MULTI(4, 4, 8);
      if (q <= 8) PSA_MULTI(4, 8, 8);
      if (q <= 16) PSA_MULTI(4, 16, 8);
    }
    if (g_variant == 7) {  // experiment: the same structure at K = 128 / 256
      if (q <= 32) PSA_MULTI(4, 32, 8);
      if (q <= 64) PSA_MULTI(4, 64, 8);
    }
#undef PSA_MULTI
    if (q <= 4) PSA_ROW(4, 4, 1);
    if (q <= 8) PSA_ROW(4, 8, 1);
    if (q <= 16) PSA_ROW(4, 16, 2);
    if (q <= 32) PSA_ROW(4, 32, 4);
    if (g_variant == 8) PSA_ROW(4, 64, 16);
    if (g_variant == 9) PSA_ROW(4, 64, 4);
    PSA_ROW(4, 64, 8);
  }
  if (K <= 4) PSA_ROW(1, 4, 1);
  if (K <= 16) PSA_ROW(1, 16, 2);
  PSA_ROW(1, 64, 8);
#undef PSA_ROW
}

}  // extern "C"
