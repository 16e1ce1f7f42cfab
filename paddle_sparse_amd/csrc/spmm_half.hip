// CSR x dense SpMM forward with HALF-WIDTH dense operands (fp16 / bf16 `mat` and
// `out`, fp32 accumulation), gfx950.
//
// The reference parametrises its tests over float16 / bfloat16 / float32 /
// float64 values (paddle_sparse/testing.py:12-21); its SpMM is absent
// (README.md:47-50), semantics as in spmm.hip.  Why a second dtype on this path:
// the forward is a pure HBM gather of dense rows, 4 K of every (12 + 4 K) bytes
// per edge — storing B and the output in 2-byte floats halves the dominant
// stream, the only ~2x lever left on a kernel that already moves fp32 rows at
// 0.9 of the HBM peak.  Products and sums are fp32 (the 2-byte inputs are
// widened exactly; the result is rounded to the 2-byte type once, on store).
//
// Same structure as the fp32 row kernels: one wavefront per CSR row, the row's
// col / value entries read once, coalesced, and handed to the gather loop by
// cross-lane moves; LPR lanes x 16 bytes (8 elements) cover the K tile, the
// G = 64 / LPR lane groups take different edges of a step (K = 128: 16 lanes per
// dense row, FOUR edges per global_load_dwordx4 instruction), U steps in flight.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"
#include "half_util.h"
#include "lane_fold.h"
#include "spmm_eb.h"
#include "vec_io.h"

namespace psa_half {
int g_half_variant = 0;  // A/B hook (half_util.h)
}

namespace {

using namespace psa_half;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

// value: fp32[nnz] (VAL32) or the same 2-byte type as mat, or NULL (weights 1)
// SMALL: the dense operand is below 4 GiB, so a row's byte offset fits 32 bits (and the column ids 31): the gather
// address is one 32-bit multiply-add on a uniform base instead of a 64-bit multiply built from three 32-bit ones, and
// only the low word of the column id crosses lanes — 4 VALU + 1 LDS instruction less per gathered row in a kernel
// that spends 80 % of its cycles issuing VALU instructions (profiles/r03_pmc_half.json).
template <typename T, int LPR, int RED, int U, bool TRACK, bool VAL32, bool SMALL = false>
__global__ void __launch_bounds__(kThreads)
spmm_half_row_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                     const void* __restrict__ val, const uint16_t* __restrict__ mat,
                     uint16_t* __restrict__ out, int64_t* __restrict__ arg_out, int64_t M, int64_t K,
                     int64_t nnz, int mean, int mix_xcds, uint8_t* __restrict__ arg_bytes = nullptr, int arg_width = 1) {
  constexpr int G = 64 / LPR;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (variant 3 switches it off for A/B: no difference on the uniform config-3 graph) inside every group of 8 consecutive row blocks the blocks go to the XCDs (round-robin by
  // blockIdx) in an order hashed from the group number: on graphs whose row lengths follow the
  // bits of the row id, XCD 0 would otherwise own all the heavy rows (spmm.hip, MaskArgs.mix_xcds)
  int64_t rb = blockIdx.x;
  if (mix_xcds) rb ^= static_cast<int64_t>((static_cast<uint32_t>(rb >> 3) * 0x9E3779B1u) >> 29);  // the grid holds whole groups of 8
  const int64_t row = rb * kWaves + wave;
  if (row >= M) return;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = (static_cast<int64_t>(blockIdx.y) * LPR + l) * 8;
  const bool kact = k0 < K;  // K % 8 == 0 (dispatch guarantees it)
  const uint16_t* matk = mat + k0;
  const int64_t s = rowptr[row], e = rowptr[row + 1];

  float acc[8];
  int64_t arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
    arg[i] = nnz;
  }
  for (int64_t base = s; base < e; base += 64) {
    const int n = (e - base) < 64 ? static_cast<int>(e - base) : 64;
    int64_t c_l = 0;
    float v_l = 1.f;
    if (lane < n) {
      c_l = col[base + lane];
      if (val != nullptr) v_l = VAL32 ? static_cast<const float*>(val)[base + lane] : widen1<T>(val, base + lane);
    }
    for (int j = 0; j < n; j += G * U) {
      uint4 raw[U];
      float w[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * G + g;  // < 64
        w[u] = __shfl(v_l, idx);
        ok[u] = (idx < n) && kact;
        raw[u] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SMALL) {
          const uint32_t c = static_cast<uint32_t>(__shfl(static_cast<int>(c_l), idx));
          const uint32_t off = c * (static_cast<uint32_t>(K) * 2u) + static_cast<uint32_t>(k0) * 2u;
          if (ok[u]) raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(mat) + off);
        } else {
          const int64_t c = shfl_i64(c_l, idx);
          if (ok[u]) raw[u] = *reinterpret_cast<const uint4*>(matk + c * K);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float b[8];
        widen8<T>(raw[u], b);
        if (RED == R_SUM) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += w[u] * b[i];  // a masked slot adds w * 0
        } else if (ok[u]) {
          const int64_t eid = base + j + u * G + g;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = w[u] * b[i];
            const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
            if (better) {
              acc[i] = x;
              if (TRACK) arg[i] = eid;
            }
          }
        }
      }
    }
  }
  // fold the G edge slots (sums: in the VALU, lane_fold.h)
  if constexpr (RED == R_SUM) psa::fold_lane_groups<LPR, 8>(acc);
#pragma unroll
  for (int off = LPR; off < 64 && RED != R_SUM; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float o = __shfl_xor(acc[i], off);
      if (RED == R_SUM) {
        acc[i] += o;
      } else if (!TRACK) {
        if (RED == R_MAX ? (o > acc[i]) : (o < acc[i])) acc[i] = o;
      } else {
        const int64_t oa = shfl_i64(arg[i], lane ^ off);
        const bool better = RED == R_MAX ? (o > acc[i]) : (o < acc[i]);
        if (better || (o == acc[i] && oa < arg[i])) {  // first winner in edge order
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
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] / d;
      }
    } else {
      if (deg == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      }
      if (TRACK && arg_out != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          typedef long V2 __attribute__((ext_vector_type(2)));
          V2 v;
          v[0] = arg[i];
          v[1] = arg[i + 1];
          __builtin_nontemporal_store(v, reinterpret_cast<V2*>(arg_out + row * K + k0 + i));
        }
      }
      if (TRACK && arg_bytes != nullptr) {  // the row-local form the one-pass backward reads (vec_io.h)
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
          uint32_t f[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) f[i] = psa::arg_local(arg[h + i] - s, deg, arg_width);
          psa::store_arg_local4(arg_bytes, row * K + k0 + h, f, arg_width);
        }
      }
    }
    const uint4 packed = narrow8<T>(acc);
    typedef unsigned int U4 __attribute__((ext_vector_type(4)));
    U4 st;
    st[0] = packed.x;
    st[1] = packed.y;
    st[2] = packed.z;
    st[3] = packed.w;
    __builtin_nontemporal_store(st, reinterpret_cast<U4*>(out + row * K + k0));
  }
}

// A/B variant (psa_spmm_half_set_variant(1)), K <= 128: G = 64 / LPR ROWS per wave, as the
// fp32 multirow kernel does for K <= 64 — every lane group owns a whole CSR row and walks its
// edges in order with U gathers in flight.  Measured on config 3 in bf16: 1.45 ms against
// 1.02 ms for one row per wave (whose col / value reads are coalesced), so it is not the default.
template <typename T, int LPR, int RED, int U, bool TRACK, bool VAL32>
__global__ void __launch_bounds__(kThreads)
spmm_half_multirow_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                          const void* __restrict__ val, const uint16_t* __restrict__ mat,
                          uint16_t* __restrict__ out, int64_t* __restrict__ arg_out, int64_t M, int64_t K,
                          int64_t nnz, int mean) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6)) * G + g;
  const int64_t k0 = static_cast<int64_t>(l) * 8;
  if (row >= M || k0 >= K) return;  // no wave-level operation below
  const uint16_t* matk = mat + k0;
  const int64_t s = rowptr[row], e = rowptr[row + 1];
  float acc[8];
  int64_t arg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
    arg[i] = nnz;
  }
  for (int64_t p = s; p < e; p += U) {
    uint4 raw[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w[u] = 0.f;
      raw[u] = make_uint4(0u, 0u, 0u, 0u);
      if (p + u < e) {
        const int64_t c = col[p + u];
        w[u] = val == nullptr ? 1.f : (VAL32 ? static_cast<const float*>(val)[p + u] : widen1<T>(val, p + u));
        raw[u] = *reinterpret_cast<const uint4*>(matk + c * K);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float b[8];
      widen8<T>(raw[u], b);
      if (RED == R_SUM) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += w[u] * b[i];
      } else if (p + u < e) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = w[u] * b[i];
          const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
          if (better) {
            acc[i] = x;
            if (TRACK) arg[i] = p + u;
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
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] / d;
    }
  } else {
    if (deg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    }
    if (TRACK && arg_out != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        typedef long V2 __attribute__((ext_vector_type(2)));
        V2 v;
        v[0] = arg[i];
        v[1] = arg[i + 1];
        __builtin_nontemporal_store(v, reinterpret_cast<V2*>(arg_out + row * K + k0 + i));
      }
    }
  }
  const uint4 packed = narrow8<T>(acc);
  typedef unsigned int U4 __attribute__((ext_vector_type(4)));
  U4 st;
  st[0] = packed.x;
  st[1] = packed.y;
  st[2] = packed.z;
  st[3] = packed.w;
  __builtin_nontemporal_store(st, reinterpret_cast<U4*>(out + row * K + k0));
}



template <typename T, int LPR, int U>
int launch_half_multirow(int red, bool track, bool val32, const int64_t* rowptr, const int64_t* col, const void* val,
                         const uint16_t* mat, uint16_t* out, int64_t* arg_out, int64_t M, int64_t K, int64_t nnz,
                         int mean, hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, static_cast<int64_t>(kWaves) * (64 / LPR));
  PSA_REQUIRE(gx <= 0x7fffffff, "problem too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx)), block(kThreads);
#define PSA_H(R, TR, V32)                                                                                   \
  hipLaunchKernelGGL((spmm_half_multirow_kernel<T, LPR, R, U, TR, V32>), grid, block, 0, s, rowptr, col, val, mat, \
                     out, arg_out, M, K, nnz, mean)
#define PSA_HV(R, TR)              \
  do {                             \
    if (val32) PSA_H(R, TR, true); \
    else PSA_H(R, TR, false);      \
  } while (0)
  if (red == R_SUM) PSA_HV(R_SUM, false);
  else if (red == R_MIN) {
    if (track) PSA_HV(R_MIN, true);
    else PSA_HV(R_MIN, false);
  } else {
    if (track) PSA_HV(R_MAX, true);
    else PSA_HV(R_MAX, false);
  }
#undef PSA_HV
#undef PSA_H
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

template <typename T, int LPR, int U>
int launch_half(int red, bool track, bool val32, const int64_t* rowptr, const int64_t* col, const void* val,
                const uint16_t* mat, uint16_t* out, int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
                hipStream_t s, uint8_t* arg_bytes = nullptr, int arg_width = 1, bool small = false) {
  const int64_t gx = psa::ceil_div(psa::ceil_div(M, kWaves), 8) * 8;  // whole groups of 8 row blocks (XCD mixing)
  const int64_t gy = psa::ceil_div(K, static_cast<int64_t>(LPR) * 8);
  PSA_REQUIRE(gx <= 0x7fffffff && gy <= 65535, "problem too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy)), block(kThreads);
#define PSA_H1(R, TR, V32, SM)                                                                                        \
  hipLaunchKernelGGL((spmm_half_row_kernel<T, LPR, R, U, TR, V32, SM>), grid, block, 0, s, rowptr, col, val, mat, out, \
                     arg_out, M, K, nnz, mean, g_half_variant == 3 ? 0 : 1, arg_bytes, arg_width)
#define PSA_H(R, TR, V32)              \
  do {                                 \
    if (small) PSA_H1(R, TR, V32, true); \
    else PSA_H1(R, TR, V32, false);    \
  } while (0)
#define PSA_HV(R, TR)        \
  do {                       \
    if (val32) PSA_H(R, TR, true); \
    else PSA_H(R, TR, false);      \
  } while (0)
  if (red == R_SUM) PSA_HV(R_SUM, false);
  else if (red == R_MIN) {
    if (track) PSA_HV(R_MIN, true);
    else PSA_HV(R_MIN, false);
  } else {
    if (track) PSA_HV(R_MAX, true);
    else PSA_HV(R_MAX, false);
  }
#undef PSA_HV
#undef PSA_H
#undef PSA_H1
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

template <typename T>
int dispatch_half(int red, bool track, bool val32, const int64_t* rowptr, const int64_t* col, const void* val,
                  const uint16_t* mat, uint16_t* out, int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
                  hipStream_t s, uint8_t* arg_bytes = nullptr, int arg_width = 1, bool small = false) {
  const int64_t q = K / 8;  // 16-byte pieces per dense row
#define PSA_MULTI(LPR, U) return launch_half_multirow<T, LPR, U>(red, track, val32, rowptr, col, val, mat, out, arg_out, M, K, nnz, mean, s)
  if (g_half_variant == 1 && arg_bytes == nullptr) {
    if (q <= 1) PSA_MULTI(1, 4);
    if (q <= 2) PSA_MULTI(2, 8);
    if (q <= 4) PSA_MULTI(4, 8);
    if (q <= 8) PSA_MULTI(8, 8);
    if (q <= 16) PSA_MULTI(16, 8);
  }
#undef PSA_MULTI
#define PSA_GO(LPR, U) return launch_half<T, LPR, U>(red, track, val32, rowptr, col, val, mat, out, arg_out, M, K, nnz, mean, s, arg_bytes, arg_width, small)
  if (q <= 1) PSA_GO(1, 1);
  if (q <= 2) PSA_GO(2, 2);
  if (q <= 4) PSA_GO(4, 4);
  if (q <= 8) PSA_GO(8, 4);
  if (q <= 16 && g_half_variant == 2) PSA_GO(16, 8);
  if (q <= 16) PSA_GO(16, 4);   // K = 128: 4 edges per gather instruction, 16 in flight
  if (q <= 32) PSA_GO(32, 4);
  PSA_GO(64, 8);                // wider K: tiles of 512 columns over grid.y
#undef PSA_GO
}


}  // namespace

extern "C" int psa_spmm_half_set_variant(int v) {
  const int prev = g_half_variant;
  g_half_variant = v;
  return prev;
}

extern "C" int psa_spmm_half_arg(int reduce, int dtype, const int64_t* rowptr, const int64_t* col, const void* value,
                                 int value_dtype, const void* mat, int64_t M, int64_t N, int64_t K, int64_t nnz,
                                 void* out, int64_t* arg_out, void* arg_bytes, int arg_width, psa_stream_t stream) {
  // a row's byte offset fits 32 bits: N is the caller's bound on col AND the height of mat (only the low 32 bits of a
  // column id are used then); variant 4 forces the 64-bit form (test hook)
  const bool small = g_half_variant != 4 && N > 0 && N * K * 2 < (1ll << 32);
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(M >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (dtype != PSA_F16 && dtype != PSA_BF16) {
    psa::set_error("psa_spmm_half: dtype must be PSA_F16 or PSA_BF16");
    return PSA_ERR_UNSUPPORTED;
  }
  if (K % 8 != 0 || !psa::aligned(mat, 16) || !psa::aligned(out, 16)) {
    psa::set_error("psa_spmm_half: needs K % 8 == 0 and 16-byte aligned operands");
    return PSA_ERR_UNSUPPORTED;
  }
  PSA_REQUIRE(value == nullptr || value_dtype == PSA_F32 || value_dtype == dtype, "value must be fp32 or mat's dtype");
  PSA_REQUIRE(arg_bytes == nullptr || ((arg_width == 1 || arg_width == 2) && psa::aligned(arg_bytes, 8)),
              "arg_bytes: width 1 or 2, 8-byte aligned");
  if (M == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(rowptr != nullptr && out != nullptr, "NULL pointer");
  PSA_REQUIRE(nnz == 0 || (col != nullptr && mat != nullptr), "col/mat is NULL");
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  const int red = reduce == PSA_MIN ? R_MIN : (reduce == PSA_MAX ? R_MAX : R_SUM);
  uint8_t* ab = minmax ? static_cast<uint8_t*>(arg_bytes) : nullptr;
  const bool track = minmax && (arg_out != nullptr || ab != nullptr);
  const bool val32 = value != nullptr && value_dtype == PSA_F32;
  hipStream_t s = psa::as_stream(stream);
  const uint16_t* m = static_cast<const uint16_t*>(mat);
  uint16_t* o = static_cast<uint16_t*>(out);
  if (dtype == PSA_BF16)
    return dispatch_half<BF16>(red, track, val32, rowptr, col, value, m, o, arg_out, M, K, nnz, reduce == PSA_MEAN, s, ab,
                               arg_width, small);
  return dispatch_half<F16>(red, track, val32, rowptr, col, value, m, o, arg_out, M, K, nnz, reduce == PSA_MEAN, s, ab,
                            arg_width, small);
}

extern "C" int psa_spmm_half(int reduce, int dtype, const int64_t* rowptr, const int64_t* col, const void* value,
                             int value_dtype, const void* mat, int64_t M, int64_t N, int64_t K, int64_t nnz,
                             void* out, int64_t* arg_out, psa_stream_t stream) {
  return psa_spmm_half_arg(reduce, dtype, rowptr, col, value, value_dtype, mat, M, N, K, nnz, out, arg_out, nullptr, 1,
                           stream);
}

extern "C" size_t psa_spmm_half_workspace_bytes(int reduce, int64_t K, int64_t nnz) {
  if (K <= 0 || nnz <= 0 || K % 8 != 0) return 0;
  return psa::eb_workspace_bytes(reduce == PSA_MIN || reduce == PSA_MAX, K, nnz, 1);
}

extern "C" int psa_spmm_half_coo(int reduce, int dtype, const int64_t* rowptr, const int64_t* row, const int64_t* col,
                                 const float* value, const void* mat, const void* hot_rows, int64_t num_hot, int64_t M,
                                 int64_t N, int64_t K, int64_t nnz, void* out, int64_t* arg_out, void* arg_bytes,
                                 int arg_width, int algo, void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(algo >= PSA_SPMM_AUTO && algo <= PSA_SPMM_EDGE_RANGES, "bad algo");
  PSA_REQUIRE(num_hot >= 0 && (num_hot == 0 || hot_rows != nullptr), "hot_rows is NULL");
  if (dtype != PSA_F16 && dtype != PSA_BF16) {
    psa::set_error("psa_spmm_half_coo: dtype must be PSA_F16 or PSA_BF16");
    return PSA_ERR_UNSUPPORTED;
  }
  const int half = dtype == PSA_BF16 ? 2 : 1;
  const bool eb = algo == PSA_SPMM_EDGE_RANGES && workspace != nullptr && nnz > 0 && N < (1ll << 31) &&
                  psa::aligned(mat, 16) && psa::aligned(out, 16) && psa::eb_supported(M, K, nnz, half) &&
                  (num_hot == 0 || psa::aligned(hot_rows, 16));
  if (!eb) {
    if (num_hot > 0) {
      psa::set_error("psa_spmm_half_coo: hot_rows is served by the edge-range kernels only (algo = PSA_SPMM_EDGE_RANGES, "
                     "K % 8 == 0, 16-byte aligned operands, a workspace)");
      return PSA_ERR_UNSUPPORTED;
    }
    return psa_spmm_half_arg(reduce, dtype, rowptr, col, value, PSA_F32, mat, M, N, K, nnz, out, arg_out, arg_bytes,
                             arg_width, stream);
  }
  PSA_REQUIRE(arg_bytes == nullptr || ((arg_width == 1 || arg_width == 2) && psa::aligned(arg_bytes, 8)),
              "arg_bytes: width 1 or 2, 8-byte aligned");
  PSA_REQUIRE(rowptr != nullptr && col != nullptr && mat != nullptr && out != nullptr, "NULL pointer");
  const int red = reduce == PSA_MIN ? R_MIN : (reduce == PSA_MAX ? R_MAX : R_SUM);
  return psa::launch_spmm_eb(red, reduce == PSA_MEAN, rowptr, row, col, value, mat, out, K, arg_out,
                             static_cast<uint8_t*>(arg_bytes), arg_bytes != nullptr ? arg_width : 1, M, N, K,
                             nnz, hot_rows, num_hot, workspace, workspace_bytes, false, 0, 0, psa::as_stream(stream), half);
}
