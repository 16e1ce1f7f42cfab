// Edge-balanced CSR x dense SpMM forward (sum / mean / min / max), gfx950: fp32 operands, or
// fp16 / bf16 dense operands with fp32 products and sums (template parameter E = elements per
// 16-byte lane load: 4 floats or 8 two-byte floats; everything else is the same kernel).
//
// Not present in the reference (README.md:47-50); semantics are upstream
// pytorch_sparse spmm (README.md:267-306), restated in oracle/spmm_oracle.c.
//
// Why a second forward beside the one-wave-per-row kernels of spmm.hip: on a
// power-law graph most rows hold 0-3 edges, so a row's wave walks the chain
// rowptr -> col/value -> gather -> store with one or two gathers in flight and
// the chip sits at ~4.5 TB/s (R-MAT scale 21) against 6.3 TB/s on a uniform
// graph.  Here the unit of work is a RANGE of `range_len` consecutive edges,
// whatever rows they belong to:
//   * a lane group (LPR lanes x float4 = one K tile) owns one range, a wave owns
//     64 / LPR ranges; the range's col / value / row ids are read coalesced, one
//     edge per lane, and handed to the gather loop by cross-lane moves — no
//     dependent pointer chase, U gather instructions always in flight;
//   * the group walks its edges in order and keeps the running reduction of the
//     current row in registers; when the row id changes it stores the finished
//     row (it is the only writer: no atomics, sums in exact edge order);
//   * a row that crosses a range boundary leaves a partial in one of the range's
//     two slots (head: the row came in from the left; tail: it goes on to the
//     right), and a second launch folds each such row's partials in range
//     order — deterministic, min/max keep the first winner;
//   * rows without edges never show up in a range: the first `fill_blocks`
//     workgroups of the same launch read rowptr (64 rows per wave instruction)
//     and store their zeros (and the arg_out sentinel).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"
#include "long_rows.h"
#include "spmm_eb.h"
#include "vec_io.h"

extern "C" int psa_ptr2ind(const int64_t* ptr, int64_t M, int64_t E, int64_t* out,
                           psa_stream_t stream);

namespace {

using psa::load_vec;
using psa::load_vec_nt;
using psa::store_vec;
using psa::store_vec_nt;
using psa::store_arg_nt;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kFillRows = 1024;  // rows per workgroup of the fill role
using psa::arg_local;
using psa::store_arg_local1;
using psa::store_arg_local4;

enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

struct EbArgs {
  const int64_t* rowptr;
  const int64_t* row;
  const int64_t* col;
  const float* val;
  const void* mat;    // [N, K] fp32, or fp16 / bf16 (half != 0)
  const void* hot;    // compact copy of the most referenced rows of mat, or NULL: column ids >= ncols index it
  int64_t ncols;      // rows of mat (INT64_MAX without a hot copy: no id is "hot")
  void* out;          // [M, K] in mat's type
  int half;           // 0: fp32 operands; 1: fp16; 2: bf16 (mat, hot and out; sums and partials stay fp32)
  int64_t* arg_out;
  uint8_t* arg_bytes;
  int arg_width;  // bytes per entry of arg_bytes: 1 or 2 (vec_io.h)
  float* part_val;    // [2 * ranges, K]: slot 2r = head partial of range r, 2r + 1 = tail partial
  int64_t* part_arg;  // same shape, winners' edge ids (min/max with tracking)
  int64_t M, K, nnz, num_ranges;
  int64_t ldo;        // elements between output rows (>= K: out may be a column slice of a wider matrix)
  int range_len;      // edges per range, a multiple of LPR
  unsigned fill_blocks;
  int mean;
  int nt_gather;
  int minmax;
  int dbg;  // A/B hooks: 1 = drop the row stores (timing only), 2 = ordinary instead of non-temporal stores
  // masked sum (template MW = 1 / 2: grad of the dense operand of spmm_min / max over the CSC view, psa_spmm_minmax_bw_eb):
  // an edge's term counts for column k only where words[c, k] (c = its gathered row id) equals the edge's tag
  const uint8_t* words;      // [rows of mat, K] entries of MW bytes: the forward's row-local arg_out
  const uint8_t* hot_words;  // the same rows as `hot`, compact
  const uint8_t* tags;       // [nnz] entries of MW bytes, in the order of col / row
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// E elements of one 16-byte load as floats
template <int E>
__device__ __forceinline__ void eb_unpack(const f32x4& raw, int half, float (&f)[E]) {
  if constexpr (E == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = raw[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = __float_as_uint(raw[i]);
      if (half == 2) {  // bf16: the upper half of an fp32
        f[2 * i] = __uint_as_float(w << 16);
        f[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
      } else {
        const float2 v = __half22float2(*reinterpret_cast<const __half2*>(&w));
        f[2 * i] = v.x;
        f[2 * i + 1] = v.y;
      }
    }
  }
}

// E floats -> 16 bytes of the output type (round to nearest even), stored at element index `elem`
template <int E>
__device__ __forceinline__ void eb_store_out(void* out, int64_t elem, int half, const float (&f)[E], bool nt) {
  f32x4 v;
  if constexpr (E == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t w;
      if (half == 2) {
        const __hip_bfloat162 p = __float22bfloat162_rn(make_float2(f[2 * i], f[2 * i + 1]));
        w = *reinterpret_cast<const uint32_t*>(&p);
      } else {
        const __half2 p = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
        w = *reinterpret_cast<const uint32_t*>(&p);
      }
      v[i] = __uint_as_float(w);
    }
  }
  f32x4* dst = reinterpret_cast<f32x4*>(static_cast<char*>(out) + elem * (16 / E));
  if (nt) __builtin_nontemporal_store(v, dst);
  else *dst = v;
}

// one element (the combine kernel writes boundary rows k by k)
__device__ __forceinline__ void eb_store_out1(void* out, int64_t elem, int half, float x) {
  if (half == 0) {
    __builtin_nontemporal_store(x, static_cast<float*>(out) + elem);
  } else if (half == 2) {
    const __hip_bfloat16 h = __float2bfloat16(x);
    static_cast<uint16_t*>(out)[elem] = *reinterpret_cast<const uint16_t*>(&h);
  } else {
    const __half h = __float2half_rn(x);
    static_cast<uint16_t*>(out)[elem] = *reinterpret_cast<const uint16_t*>(&h);
  }
}

// Rows without edges: out = 0, arg_out = nnz.  One wave tests 64 rows per step
// and zeroes the empty ones, 64 / P rows per store instruction (P lanes x 16 B
// cover a row of K elements, or a 1 KiB slice of a wider one).
template <int E>
__device__ __forceinline__ void eb_fill_role(const EbArgs& a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t KU = a.K / E;  // 16-byte units per row
  int P = 1;
  while (P < KU && P < 64) P <<= 1;
  const int rps = 64 / P;
  const int sub = lane / P, q0 = lane % P;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kFillRows + wave * (kFillRows / kWaves);
  float zero[E];
#pragma unroll
  for (int i = 0; i < E; ++i) zero[i] = 0.f;
  const int64_t sentinel[4] = {a.nnz, a.nnz, a.nnz, a.nnz};
  for (int it = 0; it < kFillRows / kWaves / 64; ++it) {
    const int64_t r0 = base + it * 64;
    if (r0 >= a.M) break;
    const int64_t r = r0 + lane;
    bool empty = false;
    if (r < a.M) empty = a.rowptr[r + 1] == a.rowptr[r];
    unsigned long long mask = __ballot(empty);
    while (mask) {
      int mine = -1;
      for (int t = 0; t < rps && mask; ++t) {
        const int i = __builtin_ctzll(mask);
        mask &= mask - 1;
        if (sub == t) mine = i;
      }
      if (mine >= 0) {
        const int64_t rr = r0 + mine;
        for (int64_t q = q0; q < KU; q += P) {
          eb_store_out<E>(a.out, rr * a.ldo + E * q, a.half, zero, true);
          if (a.minmax) {
#pragma unroll
            for (int h = 0; h < E; h += 4) {
              if (a.arg_out) store_arg_nt<4>(a.arg_out + rr * a.K + E * q + h, sentinel);
              if (a.arg_bytes) {
                const uint32_t nw = arg_local(0, 0, a.arg_width);  // "no winner", as in every other kernel
                const uint32_t none[4] = {nw, nw, nw, nw};
                store_arg_local4(a.arg_bytes, rr * a.K + E * q + h, none, a.arg_width);
              }
            }
          }
        }
      }
    }
  }
}

template <int RED>
__device__ __forceinline__ float red_init() {
  return RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
}

// One finished row of a range: out (and arg_out / arg_bytes) straight from the
// registers of the lane group that reduced it.  arg < 0: no product of the row beat the
// init (all NaN, or all -inf under max): the `nnz` sentinel, as the row-wave kernels leave it.
template <int RED, bool TRACK, int E>
__device__ __forceinline__ void eb_store_row(const EbArgs& a, int64_t row, int64_t k0, int64_t start,
                                             int seg_first, int cnt, float (&acc)[E], const int (&arg)[E]) {
  if (RED == R_SUM) {
    if (a.mean && cnt > 1) {
      const float d = static_cast<float>(cnt);
#pragma unroll
      for (int i = 0; i < E; ++i) acc[i] = acc[i] / d;
    }
  } else if (TRACK) {
#pragma unroll
    for (int h = 0; h < E; h += 4) {
      if (a.arg_out) {
        int64_t g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = arg[h + i] < 0 ? a.nnz : start + arg[h + i];
        store_arg_nt<4>(a.arg_out + row * a.K + k0 + h, g);
      }
      if (a.arg_bytes) {
        uint32_t f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          f[i] = arg_local((arg[h + i] < 0 ? a.nnz - start : arg[h + i]) - seg_first, cnt, a.arg_width);
        store_arg_local4(a.arg_bytes, row * a.K + k0 + h, f, a.arg_width);
      }
    }
  }
  if (a.dbg & 1) {
    if (acc[0] != 12345.678f) return;
  }
  if (a.dbg & 4) {  // same store instructions, all into a cache-resident scratch (timing only)
    eb_store_out<E>(a.part_val, (row & 4095) * a.K + k0, E == 4 ? 0 : a.half, acc, true);
    return;
  }
  eb_store_out<E>(a.out, row * a.ldo + k0, a.half, acc, !(a.dbg & 2));
}

template <int RED, bool TRACK, int E>
__device__ __forceinline__ void eb_store_partial(const EbArgs& a, int64_t slot, int64_t k0, int64_t start,
                                                 const float (&acc)[E], const int (&arg)[E]) {
#pragma unroll
  for (int h = 0; h < E; h += 4) {
    const float part[4] = {acc[h], acc[h + 1], acc[h + 2], acc[h + 3]};
    store_vec<4>(a.part_val + slot * a.K + k0 + h, part);
  }
  if (RED != R_SUM && TRACK) {
#pragma unroll
    for (int i = 0; i < E; ++i) a.part_arg[slot * a.K + k0 + i] = arg[i] < 0 ? a.nnz : start + arg[i];
  }
}

// static_for<N>(f): f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) —
// the ring buffers below must be indexed by constants to stay in registers.
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<0, N>(f);
}

// Range role.  The gathers run through a ring of D register buffers of U edges:
// round t + D - 1 is requested, then round t is consumed, so (D - 1) * U gather
// instructions stay in flight per wave at all times.
//
// The loads of the walk (gathers and the staged col / row / value words) are
// issued from inline asm and waited for with hand-counted s_waitcnt vmcnt(N).
// gfx950 retires stores through the same in-order counter as loads, and a flush
// stores a finished row in the middle of the walk: the compiler's own
// bookkeeping answers the divergent flush with vmcnt(0) before every round (first
// form of this kernel: 1.95 ms on config 3 against 1.65 ms for one wave per row),
// which waits for the stores' round trip to HBM as well.  Counted, the wait for
// round t is vmcnt((D - 1) * U): the rounds t + 1 ... t + D - 1 requested after it
// may stay in flight, and the flush stores of round t - 1, younger than all but
// the last of them, fall inside that window — no wait ever covers a store.
// Every asm load is unconditional (lanes and rounds past the end of the range
// read a clamped, valid address and are ignored on consumption), so the count
// holds on every path.  A destination register is named by the wait statement
// that precedes its first use ("+v"), which keeps the compiler from reading it
// early (cdna_hip_programming.md 5.7, form ii).
template <bool NT>
__device__ __forceinline__ void asm_gather16(f32x4& dst, const char* p) {
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_word(int& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void asm_gather8(u32x2& dst, const uint8_t* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_gather4(u32x2& dst, const uint8_t* p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst.x) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_ushort(int& dst, const void* p) {
  asm volatile("global_load_ushort %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_ubyte(int& dst, const void* p) {
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// the same wait with the mask words of the round named as well (rounds of 2 edges)
template <int N>
__device__ __forceinline__ void asm_wait_round_masked(f32x4 (&b)[2], u32x2 (&m)[2]) {
  asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(b[0]), "+v"(b[1]), "+v"(m[0]), "+v"(m[1]) : "i"(N) : "memory");
}

template <int N, int U>
__device__ __forceinline__ void asm_wait_round(f32x4 (&b)[U]) {
  static_assert(U == 1 || U == 2 || U == 4, "rounds of 1, 2 or 4 edges");
  if constexpr (U == 1) asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(b[0]) : "i"(N) : "memory");
  else if constexpr (U == 2) asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(b[0]), "+v"(b[1]) : "i"(N) : "memory");
  else asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "i"(N) : "memory");
}

template <int LPR, int RED, int U, int D, bool TRACK, bool NT, int E, int MW = 0>
__global__ void __launch_bounds__(kThreads) spmm_eb_kernel(EbArgs a) {
  static_assert(MW == 0 || (RED == R_SUM && !TRACK && E == 4 && U == 2), "masked form: fp32 sum, rounds of 2 edges");
  constexpr int LPE = MW ? 2 : 1;  // gather instructions per edge
  if (blockIdx.x < a.fill_blocks) {  // ---- fill role ----
    if (blockIdx.y == 0) eb_fill_role<E>(a);
    return;
  }
  constexpr int ESZ = 16 / E;  // bytes per element of mat / out
  // ---- range role ----
  constexpr int G = 64 / LPR;
  constexpr int RPS = LPR / U;  // rounds per staged batch of LPR edges
  static_assert(LPR % U == 0 && RPS % D == 0 && D >= 2, "ring positions must repeat every staged batch");
  static_assert((D - 1) * U * LPE < 64, "vmcnt is a 6-bit field");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int gsel = (lane - l) << 2;  // ds_bpermute byte address of the group's lane 0
  const int64_t k0 = (static_cast<int64_t>(blockIdx.y) * LPR + l) * E;
  const bool kact = k0 < a.K;
  const char* matk = static_cast<const char*>(a.mat) + (kact ? k0 : 0) * ESZ;  // idle K lanes gather (and drop) column 0
  const char* hotk = static_cast<const char*>(a.hot) + (kact ? k0 : 0) * ESZ;
  const int64_t row_bytes = a.K * ESZ;
  const uint8_t* wordsk = a.words + (kact ? k0 : 0) * MW;
  const uint8_t* hot_wordsk = a.hot_words + (kact ? k0 : 0) * MW;
  const int64_t words_row_bytes = a.K * MW;
  const int64_t rg = ((static_cast<int64_t>(blockIdx.x) - a.fill_blocks) * kWaves + wave) * G + g;
  const int64_t start = rg * a.range_len;
  const bool active = start < a.nnz;
  const int64_t end = !active ? start : (start + a.range_len < a.nnz ? start + a.range_len : a.nnz);
  const int len = static_cast<int>(end - start);
  const int64_t last = a.nnz - 1;

  // rows of the neighbouring edges decide where a boundary row's partial goes
  int prev_row = -1, next_row = -1;
  if (active) {
    if (start > 0) prev_row = static_cast<int>(a.row[start - 1]);
    if (end < a.nnz) next_row = static_cast<int>(a.row[end]);
  }
  // the compiler's wait for these two (ordinary) loads lands here, ahead of the
  // hand-counted part; inside the walk its vmcnt(0) would drain the ring
  asm volatile("" : "+v"(prev_row), "+v"(next_row));

  float acc[E];
  int arg[E];
  int cur_row = -1;        // row being reduced (-1: none yet)
  int seg_first = 0;       // range-local index of its first edge in this range
  bool head_open = false;  // cur_row came in from the previous range
#pragma unroll
  for (int i = 0; i < E; ++i) {
    acc[i] = red_init<RED>();
    arg[i] = -1;  // nothing has won yet (see eb_store_row)
  }

  // staged edges: one per lane of the group (col and row ids fit 31 bits:
  // eb_supported; the low words of the int64 entries are read)
  struct Staged {
    int c, r, v;
    int t = 0;  // the edge's tag (masked form only)
  };
  const bool has_val = a.val != nullptr;
  const void* vsrc = has_val ? static_cast<const void*>(a.val) : static_cast<const void*>(a.col);  // no values: any readable word
  auto stage = [&](Staged& s, int64_t sb) {  // 3 loads
    int64_t e = sb + l;
    e = e < last ? e : last;
    asm_load_word(s.c, a.col + e);
    asm_load_word(s.r, a.row + e);
    asm_load_word(s.v, static_cast<const char*>(vsrc) + 4 * e);
    if constexpr (MW == 2) asm_load_ushort(s.t, a.tags + 2 * e);
    if constexpr (MW == 1) asm_load_ubyte(s.t, a.tags + e);
  };

  f32x4 b[D][U];
  u32x2 mk[D][U];  // masked form: words[c, k0 .. k0 + 3] of the edge's gathered row
  int tg[D][U];
  float w[D][U];
  int rr[D][U];
  auto issue = [&](auto buf, const Staged& s, int idx0) {  // U loads
    constexpr int B = decltype(buf)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sel = gsel + ((idx0 + u) << 2);
      const int64_t c = __builtin_amdgcn_ds_bpermute(sel, s.c);
      const float wv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, s.v));
      w[B][u] = has_val ? wv : 1.f;
      rr[B][u] = __builtin_amdgcn_ds_bpermute(sel, s.r);
      // ids at or above ncols name a row of the compact hot copy (psa_spmm_coo, hot_rows)
      const bool is_hot = c >= a.ncols;
      asm_gather16<NT>(b[B][u], (is_hot ? hotk : matk) + (is_hot ? c - a.ncols : c) * row_bytes);
      if constexpr (MW != 0) {
        tg[B][u] = __builtin_amdgcn_ds_bpermute(sel, s.t);
        const uint8_t* wp = (is_hot ? hot_wordsk : wordsk) + (is_hot ? c - a.ncols : c) * words_row_bytes;
        if constexpr (MW == 2) asm_gather8(mk[B][u], wp);
        else asm_gather4(mk[B][u], wp);
      }
    }
  };
  auto consume = [&](auto buf, int local0) {
    constexpr int B = decltype(buf)::value;
    if constexpr (MW != 0) asm_wait_round_masked<(D - 1) * U * LPE>(b[B], mk[B]);
    else asm_wait_round<(D - 1) * U, U>(b[B]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int local = local0 + u;
      if (local < len) {  // uniform inside the lane group
        if (rr[B][u] != cur_row) {
          if (cur_row >= 0 && kact) {
            if (head_open) eb_store_partial<RED, TRACK, E>(a, 2 * rg, k0, start, acc, arg);
            else eb_store_row<RED, TRACK, E>(a, cur_row, k0, start, seg_first, local - seg_first, acc, arg);
          }
          head_open = cur_row < 0 && rr[B][u] == prev_row;
          cur_row = rr[B][u];
          seg_first = local;
#pragma unroll
          for (int i = 0; i < E; ++i) {
            acc[i] = red_init<RED>();
            if (RED != R_SUM && TRACK) arg[i] = -1;  // not the previous row's winner when nothing beats the init
          }
        }
        float bf[E];
        eb_unpack<E>(b[B][u], a.half, bf);
        if constexpr (MW != 0) {  // only where the forward named this edge the winner
          const uint32_t tag = static_cast<uint32_t>(tg[B][u]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t f = MW == 2 ? (((i < 2 ? mk[B][u].x : mk[B][u].y) >> (16 * (i & 1))) & 0xffffu)
                                       : ((mk[B][u].x >> (8 * i)) & 0xffu);
            if (f != tag) bf[i] = 0.f;
          }
        }
        if (RED == R_SUM) {
#pragma unroll
          for (int i = 0; i < E; ++i) acc[i] += w[B][u] * bf[i];
        } else {
#pragma unroll
          for (int i = 0; i < E; ++i) {
            const float x = w[B][u] * bf[i];
            const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
            if (better) {
              acc[i] = x;
              if (TRACK) arg[i] = local;
            }
          }
        }
      }
    }
  };

  Staged cur, nxt, pre;
  stage(cur, start);
  stage(nxt, start + LPR);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur.c), "+v"(cur.r), "+v"(cur.v), "+v"(nxt.c), "+v"(nxt.r), "+v"(nxt.v) : : "memory");
  if constexpr (MW != 0) asm volatile("" : "+v"(cur.t), "+v"(nxt.t));
  static_for<D - 1>([&](auto q) { issue(q, cur, decltype(q)::value * U); });
  for (int64_t sb = start; sb < end; sb += LPR) {
    stage(pre, sb + 2 * LPR);
    const int local0 = static_cast<int>(sb - start);
#pragma unroll 1
    for (int q0 = 0; q0 < RPS; q0 += D) {  // D rounds per trip: ring positions are constants
      static_for<D>([&](auto d) {
        constexpr int DD = decltype(d)::value;
        const int qi = q0 + DD + D - 1;  // round requested while round q0 + DD is consumed
        const bool from_next = qi >= RPS;  // it belongs to the next staged batch
        Staged src;
        src.c = from_next ? nxt.c : cur.c;
        src.r = from_next ? nxt.r : cur.r;
        src.v = from_next ? nxt.v : cur.v;
        src.t = from_next ? nxt.t : cur.t;
        issue(std::integral_constant<int, (DD + D - 1) % D>{}, src, (from_next ? qi - RPS : qi) * U);
        consume(d, local0 + (q0 + DD) * U);
      });
    }
    // `pre` was requested before every gather of this batch, the last counted wait
    // covered it; naming it here keeps the copies below behind that wait
    asm volatile("" : "+v"(pre.c), "+v"(pre.r), "+v"(pre.v));
    if constexpr (MW != 0) asm volatile("" : "+v"(pre.t));
    cur = nxt;
    nxt = pre;
  }
  // the D - 1 rounds requested past the end: their registers must not be reused in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (cur_row >= 0 && kact) {
    if (head_open) eb_store_partial<RED, TRACK, E>(a, 2 * rg, k0, start, acc, arg);
    else if (cur_row == next_row) eb_store_partial<RED, TRACK, E>(a, 2 * rg + 1, k0, start, acc, arg);
    else eb_store_row<RED, TRACK, E>(a, cur_row, k0, start, seg_first, len - seg_first, acc, arg);
  }
}

// Rows that cross range boundaries: the wave of the range where such a row
// STARTS folds its partials in range order — the starting range's tail slot,
// then the head slot of every following range the row reaches.
template <int RED, bool TRACK>
__global__ void __launch_bounds__(kThreads) spmm_eb_combine_kernel(EbArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * kWaves + (threadIdx.x >> 6)));
  const int64_t num_waves = static_cast<int64_t>(gridDim.x) * kWaves;
  const int64_t T = a.range_len;
  for (int64_t rg = wave0; rg < a.num_ranges; rg += num_waves) {
    const int64_t start = rg * T;
    const int64_t end = start + T < a.nnz ? start + T : a.nnz;
    if (end >= a.nnz) continue;  // nothing to the right
    const int64_t r_last = a.row[end - 1];
    if (a.row[end] != r_last) continue;  // the boundary is closed
    if (a.row[start] == r_last && start > 0 && a.row[start - 1] == r_last) continue;  // a middle piece: its starter folds it
    // pieces to add: head slots of ranges rg + 1 ... rg + pieces
    int64_t pieces = 0;
    for (;;) {
      const int64_t m = rg + 1 + pieces + lane;
      const int64_t e_m = (m + 1) * T;
      const bool cont = e_m < a.nnz && a.row[e_m - 1] == r_last && a.row[e_m] == r_last;
      const unsigned long long mask = __ballot(cont);
      const int t = ~mask == 0ull ? 64 : __builtin_ctzll(~mask);
      pieces += t;
      if (t < 64) break;
    }
    pieces += 1;  // the range in which the row ends
    const int64_t rs = a.rowptr[r_last];
    const int64_t deg = a.rowptr[r_last + 1] - rs;
    for (int64_t kb = lane; kb < a.K; kb += 128) {
      const bool two = kb + 64 < a.K;
      const int64_t p_tail = (2 * rg + 1) * a.K + kb;
      float acc[2];
      int64_t arg[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool on = t == 0 || two;
        acc[t] = on ? a.part_val[p_tail + 64 * t] : 0.f;
        arg[t] = (RED == R_SUM || !TRACK || !on) ? 0 : a.part_arg[p_tail + 64 * t];
      }
      const int64_t p_head = 2 * (rg + 1) * a.K + kb;  // + 2 K per further range
      // kFold partials per K position requested per step and folded in range order.  (32 per step helps
      // the one row that spans thousands of ranges — 760 k entries: 2 970 pieces, 2.37 -> 2.29 ms — and costs
      // every ordinary boundary row 64 predicated loads: R-MAT 21 spmm_max + 0.25 ms.  8 it stays.)
      constexpr int kFold = 8;
      for (int64_t c = 0; c < pieces; c += kFold) {
        float x[2][kFold];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < kFold; ++i)
            x[t][i] = (c + i < pieces && (t == 0 || two)) ? a.part_val[p_head + 64 * t + (c + i) * 2 * a.K] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < kFold; ++i) {
            if (c + i >= pieces || (t == 1 && !two)) break;
            if (RED == R_SUM) {
              acc[t] += x[t][i];
            } else if (RED == R_MAX ? (x[t][i] > acc[t]) : (x[t][i] < acc[t])) {
              acc[t] = x[t][i];
              if (TRACK) arg[t] = a.part_arg[p_head + 64 * t + (c + i) * 2 * a.K];
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
        const int64_t k = kb + 64 * t;
        if (RED == R_SUM) {
          if (a.mean) acc[t] = acc[t] / static_cast<float>(deg);
        } else if (TRACK) {
          if (a.arg_out) __builtin_nontemporal_store(arg[t], a.arg_out + r_last * a.K + k);
          if (a.arg_bytes) store_arg_local1(a.arg_bytes, r_last * a.K + k, arg_local(arg[t] - rs, deg, a.arg_width), a.arg_width);
        }
        eb_store_out1(a.out, r_last * a.ldo + k, a.half, acc[t]);
      }
    }
  }
}

// {empty rows, rows of 1-2 entries, rows above 128 entries, longest row}: one
// 64-bit atomic per counter and workgroup.
__global__ void __launch_bounds__(kThreads)
csr_row_stats_kernel(const int64_t* __restrict__ rowptr, int64_t M, unsigned long long* __restrict__ stats) {
  __shared__ unsigned long long sh[kWaves][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long empty = 0, tiny = 0, big = 0, longest = 0;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; r < M;
       r += static_cast<int64_t>(gridDim.x) * kThreads) {
    const unsigned long long deg = static_cast<unsigned long long>(rowptr[r + 1] - rowptr[r]);
    empty += deg == 0;
    tiny += deg >= 1 && deg <= 2;
    big += deg > 128;
    longest = deg > longest ? deg : longest;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    empty += __shfl_xor(empty, off);
    tiny += __shfl_xor(tiny, off);
    big += __shfl_xor(big, off);
    const unsigned long long o = __shfl_xor(longest, off);
    longest = o > longest ? o : longest;
  }
  if (lane == 0) {
    sh[wave][0] = empty;
    sh[wave][1] = tiny;
    sh[wave][2] = big;
    sh[wave][3] = longest;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long v = sh[0][threadIdx.x];
    for (int w = 1; w < kWaves; ++w) v = threadIdx.x == 3 ? (sh[w][3] > v ? sh[w][3] : v) : v + sh[w][threadIdx.x];
    if (threadIdx.x == 3) atomicMax(stats + 3, v);
    else if (v) atomicAdd(stats + threadIdx.x, v);
  }
}

struct EbPlan {
  int lpr;
  int range_len;
  int k_tiles;
};

EbPlan eb_plan(int64_t K, int range_len_override, int E = 4) {
  const int64_t q = K / E;  // 16-byte units per row
  EbPlan p;
  p.k_tiles = 1;
  if (q <= 4) p.lpr = 4;
  else if (q <= 8) p.lpr = 8;
  else if (q <= 16) p.lpr = 16;
  else if (q <= 32) p.lpr = 32;
  else if (q < 48) p.lpr = 64;
  else {  // ceil(q / 32) tiles of the 32-lane form over grid.y (K = 128 fp32), as the row kernels do
    p.lpr = 32;
    p.k_tiles = static_cast<int>(psa::ceil_div(q, 32));
  }
  // a wave (64 / LPR ranges) takes 512 edges, a range at least 32
  p.range_len = p.lpr >= 32 ? 256 : (p.lpr == 16 ? 128 : (p.lpr == 8 ? 64 : 32));
  if (range_len_override > 0) p.range_len = static_cast<int>(psa::ceil_div(range_len_override, p.lpr)) * p.lpr;
  return p;
}

// The scratch holds two partial slots per range of the planned length (an
// override for A/B runs may go down to 128 edges, not below the plan).
int scratch_range_len(int64_t K, int E = 4) {
  const int d = eb_plan(K, 0, E).range_len;
  return d < 128 ? d : 128;
}

}  // namespace

namespace psa {

bool eb_supported(int64_t M, int64_t K, int64_t nnz, int half) {
  const int E = half ? 8 : 4;
  return K > 0 && K % E == 0 && M > 0 && M < (1ll << 31) && nnz < (1ll << 31);  // and N < 2^31: the caller checks
}

size_t eb_workspace_bytes(bool minmax, int64_t K, int64_t nnz, int half) {
  if (K <= 0 || nnz <= 0) return 256;
  const size_t ranges = static_cast<size_t>(ceil_div(nnz, scratch_range_len(K, half ? 8 : 4)));
  return align256(static_cast<size_t>(nnz) * sizeof(int64_t)) + align256(2 * ranges * K * sizeof(float)) +
         (minmax ? align256(2 * ranges * K * sizeof(int64_t)) : 0);
}

int launch_spmm_eb(int red, int mean, const int64_t* rowptr, const int64_t* row,
                   const int64_t* col, const float* val, const void* mat, void* out, int64_t ldo,
                   int64_t* arg_out, uint8_t* arg_bytes, int arg_width, int64_t M, int64_t N, int64_t K,
                   int64_t nnz, const void* hot_rows, int64_t num_hot, void* workspace, size_t workspace_bytes,
                   bool nt_gather, int range_len_override, int dbg, hipStream_t s, int half, const EbMask* mask) {
  PSA_REQUIRE(num_hot >= 0 && N + num_hot < (1ll << 31), "column ids (with the hot copy) must fit 31 bits");
  PSA_REQUIRE(half >= 0 && half <= 2, "half: 0 fp32, 1 fp16, 2 bf16");
  PSA_REQUIRE(eb_supported(M, K, nnz, half), "shape not served by the edge-balanced kernels");
  const int E = half ? 8 : 4;
  const bool minmax = red != R_SUM;
  if (workspace == nullptr || workspace_bytes < eb_workspace_bytes(minmax, K, nnz, half)) {
    set_error("psa_spmm: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  PSA_REQUIRE(aligned(workspace, 16), "workspace must be 16-byte aligned");
  const EbPlan plan = eb_plan(K, range_len_override, E);
  PSA_REQUIRE(plan.range_len >= scratch_range_len(K, E), "range too short for the scratch layout");
  char* p = static_cast<char*>(workspace);
  if (row == nullptr && nnz > 0) {
    int64_t* built = reinterpret_cast<int64_t*>(p);
    const int st = psa_ptr2ind(rowptr, M, nnz, built, reinterpret_cast<psa_stream_t>(s));
    if (st != PSA_OK) return st;
    row = built;
  }
  p += align256(static_cast<size_t>(nnz > 0 ? nnz : 0) * sizeof(int64_t));
  EbArgs a;
  a.rowptr = rowptr;
  a.row = row;
  a.col = col;
  a.val = val;
  a.mat = mat;
  a.hot = num_hot > 0 ? hot_rows : nullptr;
  a.ncols = num_hot > 0 ? N : INT64_MAX;
  a.out = out;
  a.half = half;
  a.ldo = ldo;
  a.arg_out = minmax ? arg_out : nullptr;
  a.arg_bytes = minmax ? arg_bytes : nullptr;
  a.arg_width = arg_width;
  a.M = M;
  a.K = K;
  a.nnz = nnz;
  a.range_len = plan.range_len;
  a.num_ranges = ceil_div(nnz, plan.range_len);
  const size_t slots = 2 * static_cast<size_t>(ceil_div(nnz > 0 ? nnz : 1, scratch_range_len(K, E)));
  a.part_val = reinterpret_cast<float*>(p);
  p += align256(slots * K * sizeof(float));
  a.part_arg = minmax ? reinterpret_cast<int64_t*>(p) : nullptr;
  a.fill_blocks = static_cast<unsigned>(ceil_div(M, kFillRows));
  a.mean = mean;
  a.nt_gather = nt_gather ? 1 : 0;
  a.minmax = minmax ? 1 : 0;
  a.dbg = dbg;
  a.words = a.hot_words = a.tags = nullptr;
  const int mw = mask != nullptr ? mask->width : 0;
  if (mw != 0) {
    PSA_REQUIRE(!half && red == R_SUM && (mw == 1 || mw == 2), "masked form: fp32 sum, entries of 1 or 2 bytes");
    PSA_REQUIRE(mask->words != nullptr && mask->tags != nullptr && (num_hot == 0 || mask->hot_words != nullptr), "NULL mask");
    PSA_REQUIRE(aligned(mask->words, 4 * mw) && (num_hot == 0 || aligned(mask->hot_words, 4 * mw)), "mask alignment");
    a.words = static_cast<const uint8_t*>(mask->words);
    a.hot_words = static_cast<const uint8_t*>(num_hot > 0 ? mask->hot_words : mask->words);
    a.tags = static_cast<const uint8_t*>(mask->tags);
  }
  const int G = 64 / plan.lpr;
  const int64_t range_blocks = ceil_div(a.num_ranges, static_cast<int64_t>(G) * kWaves);
  const int64_t gx = static_cast<int64_t>(a.fill_blocks) + range_blocks;
  PSA_REQUIRE(gx <= 0x7fffffff, "problem too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(plan.k_tiles)), block(kThreads);
  const bool track = minmax && (arg_out != nullptr || arg_bytes != nullptr);
  int64_t cblocks = ceil_div(a.num_ranges, kWaves);
  if (cblocks > 4096) cblocks = 4096;
  const dim3 cgrid(static_cast<unsigned>(cblocks > 0 ? cblocks : 1));

#define PSA_EB_NT(LPR, U, D, NT, E)                                                                        \
  do {                                                                                            \
    if (red == R_SUM) {                                                                           \
      hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_SUM, U, D, false, NT, E>), grid, block, 0, s, a);        \
      if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_SUM, false>), cgrid, block, 0, s, a); \
    } else if (red == R_MIN) {                                                                    \
      if (track) {                                                                                \
        hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_MIN, U, D, true, NT, E>), grid, block, 0, s, a);       \
        if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_MIN, true>), cgrid, block, 0, s, a); \
      } else {                                                                                    \
        hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_MIN, U, D, false, NT, E>), grid, block, 0, s, a);      \
        if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_MIN, false>), cgrid, block, 0, s, a); \
      }                                                                                           \
    } else {                                                                                      \
      if (track) {                                                                                \
        hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_MAX, U, D, true, NT, E>), grid, block, 0, s, a);       \
        if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_MAX, true>), cgrid, block, 0, s, a); \
      } else {                                                                                    \
        hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_MAX, U, D, false, NT, E>), grid, block, 0, s, a);      \
        if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_MAX, false>), cgrid, block, 0, s, a); \
      }                                                                                           \
    }                                                                                             \
  } while (0)

#define PSA_EB(LPR, U, D)                  \
  do {                                     \
    if (nt_gather) PSA_EB_NT(LPR, U, D, true, 4); \
    else PSA_EB_NT(LPR, U, D, false, 4);      \
  } while (0)
#define PSA_EB_MASKED(LPR, D, MW)                                                                              \
  do {                                                                                                        \
    hipLaunchKernelGGL((spmm_eb_kernel<LPR, R_SUM, 2, D, false, false, 4, MW>), grid, block, 0, s, a);          \
    if (a.num_ranges > 1) hipLaunchKernelGGL((spmm_eb_combine_kernel<R_SUM, false>), cgrid, block, 0, s, a); \
  } while (0)
  if (mw != 0) {  // masked sum: two gathers per edge (the row of mat and its row of words)
    switch (plan.lpr * 4 + mw) {
      case 4 * 4 + 1: PSA_EB_MASKED(4, 2, 1); break;
      case 4 * 4 + 2: PSA_EB_MASKED(4, 2, 2); break;
      case 8 * 4 + 1: PSA_EB_MASKED(8, 4, 1); break;
      case 8 * 4 + 2: PSA_EB_MASKED(8, 4, 2); break;
      case 16 * 4 + 1: PSA_EB_MASKED(16, 4, 1); break;
      case 16 * 4 + 2: PSA_EB_MASKED(16, 4, 2); break;
      case 32 * 4 + 1: PSA_EB_MASKED(32, 4, 1); break;
      case 32 * 4 + 2: PSA_EB_MASKED(32, 4, 2); break;
      case 64 * 4 + 1: PSA_EB_MASKED(64, 4, 1); break;
      default: PSA_EB_MASKED(64, 4, 2); break;
    }
  } else if (half) {  // two-byte operands: 8 elements per lane, ordinary (cached) gathers
    switch (plan.lpr) {
      case 4: PSA_EB_NT(4, 2, 2, false, 8); break;
      case 8: PSA_EB_NT(8, 2, 4, false, 8); break;
      case 16: PSA_EB_NT(16, 2, 4, false, 8); break;
      case 32: PSA_EB_NT(32, 2, 4, false, 8); break;
      default: PSA_EB_NT(64, 2, 4, false, 8); break;
    }
  } else {
    switch (plan.lpr) {
      case 4: PSA_EB_NT(4, 2, 2, false, 4); break;
      case 8: PSA_EB_NT(8, 2, 4, false, 4); break;
      case 16: PSA_EB_NT(16, 2, 4, false, 4); break;
      case 32: PSA_EB(32, 2, 4); break;
      default: PSA_EB(64, 2, 4); break;
    }
  }
#undef PSA_EB
#undef PSA_EB_NT
#undef PSA_EB_MASKED
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // namespace psa

extern "C" int psa_csr_row_stats(const int64_t* rowptr, int64_t M, int64_t* stats, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0, "negative size");
  PSA_REQUIRE(stats != nullptr && (M == 0 || rowptr != nullptr), "NULL pointer");
  hipStream_t s = psa::as_stream(stream);
  PSA_ZERO(stats, 4 * sizeof(int64_t), s);
  if (M == 0) return PSA_OK;
  int64_t blocks = psa::ceil_div(M, kThreads);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(csr_row_stats_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, rowptr, M,
                     reinterpret_cast<unsigned long long*>(stats));
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}
