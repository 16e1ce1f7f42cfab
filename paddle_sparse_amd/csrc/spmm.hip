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
#include <type_traits>

#include "common.h"
#include "lane_fold.h"
#include "long_rows.h"
#include "spmm_eb.h"
#include "vec_io.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

enum { R_SUM = 0, R_MIN = 1, R_MAX = 2 };

// Output rows may sit inside a wider matrix (a column slice of it): row r starts at p + r * ld.
struct OutView {
  float* p;
  int64_t ld;
};

using psa::load_vec;
using psa::load_vec_nt;
using psa::shfl_i64;
using psa::store_arg_nt;
using psa::store_vec;
using psa::store_vec_nt;
using psa::Vec;

// ---------------------------------------------------------------------------
// Long rows (power-law graphs).  One wave pulls only a few GB/s (8 gathers in
// flight), so a 40 000-edge row would keep its wave busy for milliseconds
// while the rest of the chip idles (R-MAT scale 21: 9.5 ms against 1.8 ms for
// a uniform graph of the same size).  Rows longer than kLongRow are therefore
// NOT reduced by their row wave: it records (row, first chunk) in a work list
// with ONE 64-bit atomic {rows << 32 | chunks} (so list order == chunk order),
// a second launch reduces every kLongChunk-edge chunk with its own wave into a
// partials buffer, and a third folds each row's partials IN CHUNK ORDER
// (deterministic, no float atomics; min/max keep the first winner).
// R-MAT scale 21, 19.5 M edges, K=128: 5.8 -> 2.5 ms; the chunk launch itself
// runs at 5.7 TB/s (11.7 M edges in 1.05 ms).  Cost when no row is long: three
// near-empty launches, ~5 us (visible only on sub-100-us problems).
// ---------------------------------------------------------------------------
using psa::find_long_entry;
using psa::kLongBlocks;
using psa::kLongChunk;
using psa::kLongRow;
using psa::LongEntry;
using psa::max_long_chunks;
using psa::max_long_rows;
using psa::push_long_row;

// MASK form of the reduction (grad of the dense operand of spmm_min/max, taken
// over the CSC view without atomics): edge j of "row" c is the CSR edge
// edge_id[j] = (r, c); its term w * grad[r, k] counts for output column k only
// where the forward's arg_out[r, k] named that edge.  bytes[r, k] holds
// arg_out[r, k] as an index local to row r (one byte instead of eight), tag[j]
// the local index of edge j in its row.  The byte is (index & 127), plus bit 7
// when row r has more than 128 edges: for short rows equal bytes ARE the hit; for
// long rows they name a candidate (the winner, or one of the ~1/128 of the row's
// edges whose index aliases it) that is then compared against arg_out itself.
// (A first encoding marked long rows with one reserved value and tested EVERY
// (edge, k) of such rows against arg_out: 8 K bytes more per edge — on an R-MAT
// graph, 40 % of whose edges sit in rows above 255, the pass took 8.3 ms.)  `val` is then the CSR-ordered value
// array (read through edge_id), `col` the CSR row of every CSC edge.
// With grad_value set, the same pass also forms grad_value[e] = sum over the
// hit columns k of mat[c, k] * grad[r, k]: the gathered grad row is already in
// registers, mat[c, :] is this wave's own row (mrow), and every CSR edge shows
// up exactly once in the CSC walk, so each grad_value element is stored once
// (in CSC order; the caller permutes it back with one gather through csc2csr).
// MODE of reduce_edge_range / spmm_fused_kernel:
//   M_PLAIN  the forward (and grad_mat of sum/mean with ready-made CSC weights);
//   M_MASK   min/max backward over the CSC view (byte test described above);
//   M_CSC    sum/mean backward over the CSC view: the same indirection (value
//            read through edge_id) and the same grad_value dot, every term on;
//   M_NOARG  min/max forward whose caller wants `out` only: the winners' edge
//            ids are not tracked (8 VGPRs and the 64-bit shuffles of the fold less).
enum { M_PLAIN = 0, M_MASK = 1, M_CSC = 2, M_NOARG = 3 };

constexpr int kFusedChunkBlocksDefault = 768;
// ... and of the CSC-view backward kernels (more registers, fewer resident workgroups): R-MAT 21, both
// gradients, 256 / 384 / 512 / 768 / 1024 workgroups: sum 2.63 / 2.21 / 2.31 / 2.46 / 2.51 ms, max 3.14 / 2.94 /
// 3.13 / 3.25 / 3.27 ms; no effect on a uniform graph (profiles/r02_rmat_backward.txt)
constexpr int kMaskedChunkBlocks = 384;
using psa::arg_local;
using psa::kByteExact;
using psa::kWordExact;
using psa::store_arg_local1;
using psa::store_arg_local4;

// where a forward leaves the row-local form of arg_out, and how wide its entries are
struct ArgLocal {
  uint8_t* p = nullptr;
  int width = 1;
};

struct MaskArgs {
  const uint8_t* bytes = nullptr;    // [M, K] at the lane's k0
  const uint8_t* tag = nullptr;      // [nnz], CSC order
  const int64_t* edge_id = nullptr;  // csr2csc
  const int64_t* arg = nullptr;      // arg_out [M, K] at the lane's k0
  const float* mat = nullptr;        // dense operand of the forward [N, K] (grad_value only)
  const float* mrow = nullptr;       // set by the kernel: mat[c, k0..] of the wave's column
  float* grad_value = nullptr;       // [nnz] in CSC order (position j <-> edge edge_id[j]), or NULL
  uint8_t* arg_bytes_out = nullptr;  // M_PLAIN min/max: also store arg_out as row-local bytes (see M_MASK)
  int arg_width = 1;                 // bytes per entry of bytes / tag / arg_bytes_out: 1 or 2 (vec_io.h)
  // M_MASK / M_CSC: ids >= hot_first in `col` name rows of a compact copy of the gathered operand
  // (hot, [h, K]) and of bytes (hot_bytes): the hub rows of a power-law graph, which otherwise
  // crowd a few memory channels
  const float* hot = nullptr;
  const uint8_t* hot_bytes = nullptr;
  int64_t hot_first = INT64_MAX;
  const float* row_scale = nullptr;  // M_CSC, mean: 1 / max(deg(r), 1) per CSR row, folded into both gradients
  // row role: inside every group of 8 consecutive row blocks, hand the blocks to the XCDs in an
  // order hashed from the group number.  Workgroups go to the 8 XCDs round-robin, so XCD x would
  // otherwise own the rows whose id has bits 2-4 equal to x; on a graph whose row lengths follow
  // the bits of the id (R-MAT as generated: a column of the CSC view is heavy when its id has few
  // bits set) XCD 0 then gets 32x the edges of XCD 7 and the launch waits for it
  // (profiles/r02_rmat_backward.txt: same requests, same wave-cycles, 1.4x the duration).
  int mix_xcds = 0;
  int xcd_rows = 0;                  // row role: give each XCD one contiguous eighth of the rows
  int temporal_out = 0;              // A/B hook (variant 17): ordinary instead of non-temporal output stores
  int chunk_blocks = kFusedChunkBlocksDefault;  // workgroups in the chunk role (variants 20-22 for A/B)
  int nt_gather = 0;                 // non-temporal loads for the gathered dense rows (operands >> Infinity Cache; variant 18 forces it)
};

// Reduce edges [s, e) of one row into acc/arg: LPR lanes x VEC floats cover the
// K tile at k0, the G = 64/LPR lane groups take different edges of a step, U
// steps are issued before any is consumed; the groups are folded at the end.
// AW (M_MASK): form of bytes / tag — 1: one byte per entry, candidates of long rows verified against
// arg_out (second phase); 0: one byte, no arg_out at hand (candidates of long rows count as no hit);
// 2: two bytes per entry, exact.  HOT: ids >= m.hot_first go to the compact copies
template <int VEC, int LPR, int RED, int U, int MODE = M_PLAIN, int AW = 1, bool HOT = false>
__device__ __forceinline__ void reduce_edge_range(
    const int64_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ matk, int64_t K, bool kact, int64_t s, int64_t e,
    int64_t nnz, int lane, float (&acc)[VEC], int64_t (&arg)[VEC],
    const MaskArgs& m = MaskArgs{}) {
  static_assert(MODE == M_PLAIN || MODE == M_NOARG || (RED == R_SUM && VEC == 4), "CSC forms: sum over float4 tiles");
  constexpr bool INDIRECT = MODE == M_MASK || MODE == M_CSC;
  constexpr bool TRACK = MODE != M_NOARG;
  constexpr bool MASK = MODE == M_MASK;
  constexpr int G = 64 / LPR;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int g = lane / LPR;
  // the winners are tracked as 32-bit indices local to `s` (4 VGPRs and 32-bit selects /
  // shuffles instead of 8 and 64-bit ones); INT32_MAX = none yet, so real ones order below it
  int32_t la[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc[i] = RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
    la[i] = INT32_MAX;
  }
  float mr[VEC];  // MASK + grad_value: this lane's slice of mat[c, :]
#pragma unroll
  for (int i = 0; i < VEC; ++i) mr[i] = 0.f;
  const bool want_gv = INDIRECT && m.grad_value != nullptr;  // wave-uniform
  if (want_gv && kact) {  // the column's own row of `mat`: read once, streamed past the caches
    if (m.temporal_out) load_vec<VEC>(m.mrow, mr);
    else load_vec_nt<VEC>(m.mrow, mr);
  }
  for (int64_t base = s; base < e; base += 64) {
    const int n = (e - base) < 64 ? static_cast<int>(e - base) : 64;
    int64_t c_l = 0;
    float v_l = 0.f;
    float gv_keep = 0.f;  // grad_value of this lane's edge, collected step by step
    int64_t id_l = 0;  // MASK: CSR edge id; its tag rides in the top byte
    float s_l = 1.f;   // M_CSC with row_scale: 1 / deg of this edge's CSR row
    if (lane < n) {
      c_l = col[base + lane];
      if (INDIRECT) {
        if (m.edge_id) {  // values in CSR order, read through the entry's CSR id
          id_l = m.edge_id[base + lane];
          v_l = val ? val[id_l] : 1.f;
        } else {  // values handed over in CSC order already (psa_permute_apply_u32): a stream
          v_l = val ? val[base + lane] : 1.f;
        }
        if (MODE == M_CSC && m.row_scale) s_l = m.row_scale[c_l];  // multiplied in where the weight is consumed
        if (MASK) {  // the tag rides in the top bits of the edge id (ids stay below 2^48)
          if (AW == 2) id_l |= static_cast<int64_t>(reinterpret_cast<const uint16_t*>(m.tag)[base + lane]) << 48;
          else id_l |= static_cast<int64_t>(m.tag[base + lane]) << 56;
        }
      } else {
        v_l = val ? val[base + lane] : 1.f;
      }
    }
    for (int j = 0; j < n; j += G * U) {
      float b[U][VEC];
      float w[U];
      bool ok[U];
      uint32_t mb[U], mb2[U];  // the (row, k0 .. k0 + 3) entries of bytes: 4 x 1 byte, or 4 x 2 bytes in two words
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * G + g;  // < 64 by the static_assert
        const int64_t c = shfl_i64(c_l, idx);
        // (the weight is fetched from its lane after the gathers are out: over the CSC view it comes
        // through a dependent read, val[edge_id], which would otherwise hold the gathers back one hop)
        if (!INDIRECT) w[u] = __shfl(v_l, idx);
        ok[u] = (idx < n) && kact;
#pragma unroll
        for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
        const bool is_hot = HOT && c >= m.hot_first;
        const int64_t off = (is_hot ? c - m.hot_first : c) * K;
        if (ok[u]) {
          const float* src = (is_hot ? m.hot : matk) + off;
          if (m.nt_gather) load_vec_nt<VEC>(src, b[u]);
          else load_vec<VEC>(src, b[u]);
        }
        if (MASK) {
          mb[u] = 0;
          mb2[u] = 0;
          if (ok[u]) {
            const uint8_t* bsrc = (is_hot ? m.hot_bytes : m.bytes) + off * (AW == 2 ? 2 : 1);
            if (AW == 2) {
              const uint2 w2 = *reinterpret_cast<const uint2*>(bsrc);
              mb[u] = w2.x;
              mb2[u] = w2.y;
            } else {
              mb[u] = *reinterpret_cast<const uint32_t*>(bsrc);
            }
          }
        }
      }
      if (MASK) {
        // Phase 1: byte compare.  hits bit i = (u, i) counts; need bit i = it is only a
        // candidate (row of more than 128 edges) and must be checked against arg_out.
        uint32_t hits[U], need[U];
        bool any_need = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // edge id / row are re-read from their lanes here rather than kept in
          // registers across the gathers (88 -> fewer VGPRs, one more wave per SIMD)
          const int idx = j + u * G + g;
          const uint64_t idw = static_cast<uint64_t>(shfl_i64(id_l, idx));
          hits[u] = 0;
          uint32_t tag;
          if (AW == 2) {  // exact: equal entries are the hits
            tag = static_cast<uint32_t>(idw >> 48);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              hits[u] |= static_cast<uint32_t>((((i < 2 ? mb[u] : mb2[u]) >> (16 * (i & 1))) & 0xffffu) == tag) << i;
            tag = 0;  // no candidates to verify
          } else {
            tag = static_cast<uint32_t>(idw >> 56);
#pragma unroll
            for (int i = 0; i < VEC; ++i) hits[u] |= static_cast<uint32_t>(((mb[u] >> (8 * i)) & 255u) == tag) << i;
          }
          need[u] = ((tag & 0x80u) && ok[u]) ? hits[u] : 0u;
          any_need |= need[u] != 0;
        }
        // Phase 2 (power-law graphs only): ALL exact tests of the step are requested
        // before any is consumed — one at a time, each a dependent memory round trip
        // inside the gather loop, they made this pass 1.5x slower than float atomics
        // on an R-MAT graph.  The low words decide: arg_out[r, k] is an edge of row r,
        // and two edges of one row differ in their low 32 bits.
        if (AW == 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) hits[u] &= ~need[u];
        }
        if (AW == 1 && __any(any_need)) {
          int32_t seen[U][VEC];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t r = shfl_i64(c_l, j + u * G + g);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              seen[u][i] = 0;
              if (((need[u] >> i) & 1u) && m.arg != nullptr)
                seen[u][i] = *reinterpret_cast<const int32_t*>(m.arg + r * K + i);
            }
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int32_t id_lo = static_cast<int32_t>(shfl_i64(id_l, j + u * G + g));
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              if (((need[u] >> i) & 1u) && (m.arg == nullptr || seen[u][i] != id_lo)) hits[u] &= ~(1u << i);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            if (!((hits[u] >> i) & 1u)) b[u][i] = 0.f;
          }
        }
      }
      if (INDIRECT) {
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __shfl(v_l, j + u * G + g);
        if (MODE == M_CSC && m.row_scale) {  // wave-uniform
#pragma unroll
          for (int u = 0; u < U; ++u) w[u] *= __shfl(s_l, j + u * G + g);
        }
        if (want_gv) {
          static_assert(!INDIRECT || ((U & (U - 1)) == 0 && U <= LPR), "U must be a power of two <= LPR");
          const int l = lane % LPR;
          float dot[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            dot[u] = 0.f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) dot[u] += b[u][i] * mr[i];
          }
          // fold the U partial dots of the lane group: lane l ends up with the whole dot of edge slot u = l % U
          // (lane_fold.h: partners 1, 2, 4, 8 lanes away as DPP operands of the add, 16 by ds_bpermute)
          psa::fold_group_dots<LPR, U>(dot, l);
          // Lane l < U of group g now holds the dot of edge slot j + l * G + g.  It is handed to
          // the lane that loaded that edge (lane == slot) and stored once per 64-edge batch, 256
          // contiguous bytes, instead of 8 floats per step (partial-line writes).  The store goes to
          // the edge's CSC position: contiguous per wave.  (Storing straight to the CSR position, a
          // 4-byte scatter, cost 0.8 ms more at 20 M edges than this store plus the caller's gather
          // through csc2csr.)
          const unsigned rel = static_cast<unsigned>(lane - j);  // this lane's edge belongs to the step iff rel < G * U
          const float got = __shfl(dot[0], static_cast<int>(((rel % G) * LPR + rel / G) & 63u));
          if (rel < static_cast<unsigned>(G * U)) gv_keep = got;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (RED == R_SUM) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += w[u] * b[u][i];
        } else if (ok[u]) {
          const int32_t eid = static_cast<int32_t>(base - s) + j + u * G + g;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float x = w[u] * b[u][i];
            const bool better = RED == R_MAX ? (x > acc[i]) : (x < acc[i]);
            if (better) {
              acc[i] = x;
              if (TRACK) la[i] = eid;
            }
          }
        }
      }
    }
    if (INDIRECT && want_gv && lane < n)
      __builtin_nontemporal_store(gv_keep * (MODE == M_CSC ? s_l : 1.f), m.grad_value + base + lane);
  }
  // Fold the G edge slots.
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float o = __shfl_xor(acc[i], off);
      if (RED == R_SUM) {
        acc[i] += o;
      } else if (!TRACK) {
        const bool better = RED == R_MAX ? (o > acc[i]) : (o < acc[i]);
        if (better) acc[i] = o;
      } else {
        const int32_t oa = __shfl_xor(la[i], off);
        // first winner in edge order: ties go to the smaller edge id
        const bool better = RED == R_MAX ? (o > acc[i]) : (o < acc[i]);
        if (better || (o == acc[i] && oa < la[i])) {
          acc[i] = o;
          la[i] = oa;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) arg[i] = (RED == R_SUM || !TRACK || la[i] == INT32_MAX) ? nnz : s + la[i];
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
                OutView out, int64_t* __restrict__ arg_out,
                int64_t M, int64_t K, int64_t nnz, int mean,
                unsigned long long* __restrict__ long_ctr,
                LongEntry* __restrict__ long_list) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  if (row >= M) return;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = static_cast<int64_t>(blockIdx.y) * (LPR * VEC) + l * VEC;
  const bool kact = k0 < K;  // K % VEC == 0 (dispatch guarantees it)

  const int64_t s = rowptr[row];
  const int64_t e = rowptr[row + 1];
  if (long_list && e - s > kLongRow) {  // wave-uniform
    if (lane == 0 && blockIdx.y == 0) push_long_row(long_ctr, long_list, row, e - s);
    return;
  }

  float acc[VEC];
  int64_t arg[VEC];
  reduce_edge_range<VEC, LPR, RED, U>(col, val, mat + k0, K, kact, s, e, nnz, lane, acc, arg);

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
      if (arg_out) store_arg_nt<VEC>(arg_out + row * K + k0, arg);
    }
    store_vec_nt<VEC>(out.p + row * out.ld + k0, acc);
  }
}

// ---------------------------------------------------------------------------
// Variant A2: one wavefront per R consecutive CSR rows.  The R+1 row pointers
// come from one coalesced load and the first 64 col/value entries of the
// wave's whole edge range from another, so rows that lie inside that window
// (short and empty rows: most rows of a power-law graph) pay neither their own
// pointer latency nor their own index load before their gathers can issue.
// ---------------------------------------------------------------------------
template <int VEC, int LPR, int RED, int U, int R>
__global__ void __launch_bounds__(kThreads)
spmm_rows_kernel(const int64_t* __restrict__ rowptr,
                 const int64_t* __restrict__ col,
                 const float* __restrict__ val, const float* __restrict__ mat,
                 OutView out, int64_t* __restrict__ arg_out,
                 int64_t M, int64_t K, int64_t nnz, int mean,
                 unsigned long long* __restrict__ long_ctr,
                 LongEntry* __restrict__ long_list) {
  constexpr int G = 64 / LPR;
  static_assert(64 % (G * U) == 0, "edge batch must divide the wave");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * kWaves + wave) * R;
  if (row0 >= M) return;
  const int nrows = (M - row0) < R ? static_cast<int>(M - row0) : R;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = static_cast<int64_t>(blockIdx.y) * (LPR * VEC) + l * VEC;
  const bool kact = k0 < K;
  const float* matk = mat + k0;

  const int64_t p = rowptr[row0 + (lane < nrows ? lane : nrows)];
  const int64_t win_s = shfl_i64(p, 0);
  const int64_t win_e_all = shfl_i64(p, nrows);
  int64_t c_w = 0;
  float v_w = 0.f;
  if (win_s + lane < win_e_all) {  // first 64 edges of the wave's range
    c_w = col[win_s + lane];
    v_w = val ? val[win_s + lane] : 1.f;
  }
  for (int rr = 0; rr < nrows; ++rr) {
    const int64_t row = row0 + rr;
    const int64_t s = shfl_i64(p, rr);
    const int64_t e = shfl_i64(p, rr + 1);
    if (long_list && e - s > kLongRow) {  // wave-uniform
      if (lane == 0 && blockIdx.y == 0) push_long_row(long_ctr, long_list, row, e - s);
      continue;
    }
    float acc[VEC];
    int64_t arg[VEC];
    if (e <= win_s + 64) {
      // the row lies inside the preloaded window: gather straight from it
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = RED == R_SUM ? 0.f : (RED == R_MAX ? -__FLT_MAX__ : __FLT_MAX__);
        arg[i] = nnz;
      }
      const int off = static_cast<int>(s - win_s);
      const int n = static_cast<int>(e - s);
      for (int j = 0; j < n; j += G * U) {
        float b[U][VEC];
        float w[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = j + u * G + g;
          const int64_t c = shfl_i64(c_w, (off + idx) & 63);
          w[u] = __shfl(v_w, (off + idx) & 63);
          ok[u] = (idx < n) && kact;
#pragma unroll
          for (int i = 0; i < VEC; ++i) b[u][i] = 0.f;
          if (ok[u]) load_vec<VEC>(matk + c * K, b[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (RED == R_SUM) {
            if (ok[u]) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[i] += w[u] * b[u][i];
            }
          } else if (ok[u]) {
            const int64_t eid = s + j + u * G + g;
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
#pragma unroll
      for (int off2 = LPR; off2 < 64; off2 <<= 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float o = __shfl_xor(acc[i], off2);
          if (RED == R_SUM) {
            acc[i] += o;
          } else {
            const int64_t oa = shfl_i64(arg[i], lane ^ off2);
            const bool better = RED == R_MAX ? (o > acc[i]) : (o < acc[i]);
            if (better || (o == acc[i] && oa < arg[i])) {
              acc[i] = o;
              arg[i] = oa;
            }
          }
        }
      }
    } else {
      reduce_edge_range<VEC, LPR, RED, U>(col, val, matk, K, kact, s, e, nnz, lane, acc, arg);
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
        if (arg_out) store_arg_nt<VEC>(arg_out + row * K + k0, arg);
      }
      store_vec_nt<VEC>(out.p + row * out.ld + k0, acc);
    }
  }
}

// One wave per kLongChunk-edge chunk of a long row (grid-stride over the chunk list):
// partial[chunk, :] (and the winning edge ids for min/max).
template <int VEC, int LPR, int RED, int U>
__global__ void __launch_bounds__(psa::kLongThreads)
spmm_long_chunk_kernel(const int64_t* __restrict__ rowptr,
                       const int64_t* __restrict__ col,
                       const float* __restrict__ val, const float* __restrict__ mat,
                       int64_t K, int64_t nnz,
                       const unsigned long long* __restrict__ long_ctr,
                       const LongEntry* __restrict__ long_list,
                       float* __restrict__ part_val, int64_t* __restrict__ part_arg) {
  const int lane = threadIdx.x & 63;
  const unsigned long long ctr = *long_ctr;
  const uint32_t total = static_cast<uint32_t>(ctr & 0xffffffffull);
  const int nrows = static_cast<int>(ctr >> 32);
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int ktiles = static_cast<int>((K + LPR * VEC - 1) / (LPR * VEC));
  const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t num_waves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t c = wave_id; c < total; c += num_waves) {
    const LongEntry ent = find_long_entry(long_list, nrows, c);
    const int64_t rs = rowptr[ent.row], re = rowptr[ent.row + 1];
    const int64_t s = rs + static_cast<int64_t>(c - ent.first_chunk) * kLongChunk;
    const int64_t e = s + kLongChunk < re ? s + kLongChunk : re;
    for (int t = 0; t < ktiles; ++t) {
      const int64_t k0 = static_cast<int64_t>(t) * (LPR * VEC) + l * VEC;
      const bool kact = k0 < K;
      float acc[VEC];
      int64_t arg[VEC];
      reduce_edge_range<VEC, LPR, RED, U>(col, val, mat + k0, K, kact, s, e, nnz, lane, acc, arg);
      if (g == 0 && kact) {
        store_vec<VEC>(part_val + static_cast<int64_t>(c) * K + k0, acc);
        if (RED != R_SUM) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) part_arg[static_cast<int64_t>(c) * K + k0 + i] = arg[i];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Fused form of the long-row path (one K tile, i.e. K <= LPR*VEC): the list is
// built by a pre-pass over rowptr, then ONE launch runs both roles — the first
// kFusedChunkBlocks workgroups reduce chunks of long rows (HBM-bound), all
// the others reduce ordinary rows (bound by per-row latency on power-law
// graphs, where most rows are empty or tiny) — so the two overlap instead of
// running back to back.  Roles are told apart by blockIdx only.
// ---------------------------------------------------------------------------
// (kFusedChunkBlocksDefault = 768, x 4 waves: ~3/8 of the chip's wave slots)


template <int VEC, int LPR, int RED, int U, int MODE = M_PLAIN, int AW = 1, bool HOT = false>
__global__ void __launch_bounds__(kThreads)
spmm_fused_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                  const float* __restrict__ val, const float* __restrict__ mat,
                  OutView out, int64_t* __restrict__ arg_out, int64_t M,
                  int64_t K, int64_t nnz, int mean,
                  const unsigned long long* __restrict__ long_ctr,
                  const LongEntry* __restrict__ long_list, float* __restrict__ part_val,
                  int64_t* __restrict__ part_arg, MaskArgs mask) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / LPR;
  const int l = lane % LPR;
  const int64_t k0 = (static_cast<int64_t>(blockIdx.y) * LPR + l) * VEC;  // K tiles of LPR * VEC floats over grid.y
  const bool kact = k0 < K;
  float acc[VEC];
  int64_t arg[VEC];
  if (MODE == M_MASK) {
    mask.bytes += k0 * (AW == 2 ? 2 : 1);
    mask.arg += k0;
    if (mask.hot_bytes) mask.hot_bytes += k0 * (AW == 2 ? 2 : 1);
  }
  if ((MODE == M_MASK || MODE == M_CSC) && mask.hot) mask.hot += k0;
  const unsigned kFusedChunkBlocks = static_cast<unsigned>(mask.chunk_blocks);
  if (blockIdx.x < kFusedChunkBlocks) {  // ---- chunk role ----
    const unsigned long long ctr = *long_ctr;
    const uint32_t total = static_cast<uint32_t>(ctr & 0xffffffffull);
    const int nrows = static_cast<int>(ctr >> 32);
    for (uint32_t c = blockIdx.x * kWaves + wave; c < total; c += kFusedChunkBlocks * kWaves) {
      const LongEntry ent = find_long_entry(long_list, nrows, c);
      const int64_t rs = rowptr[ent.row], re = rowptr[ent.row + 1];
      const int64_t s = rs + static_cast<int64_t>(c - ent.first_chunk) * kLongChunk;
      const int64_t e = s + kLongChunk < re ? s + kLongChunk : re;
      if ((MODE == M_MASK || MODE == M_CSC) && mask.mat) mask.mrow = mask.mat + ent.row * K + k0;
      reduce_edge_range<VEC, LPR, RED, U, MODE, AW, HOT>(col, val, mat + k0, K, kact, s, e, nnz, lane, acc, arg, mask);
      if (g == 0 && kact) {
        store_vec<VEC>(part_val + static_cast<int64_t>(c) * K + k0, acc);
        if (RED != R_SUM && MODE != M_NOARG) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) part_arg[static_cast<int64_t>(c) * K + k0 + i] = arg[i];
        }
      }
    }
    return;
  }
  // ---- row role ----
  // Workgroups go to the 8 XCDs round-robin (blockIdx % 8), each with its own
  // L2.  With xcd_rows the row blocks are renumbered so that XCD x owns the
  // contiguous eighth x of the rows: neighbouring rows of a graph with locality
  // (communities, banded orderings) then share their B rows through ONE L2
  // instead of spreading them over eight.  (The grid is a multiple of 8 blocks.)
  int64_t rb = static_cast<int64_t>(blockIdx.x) - kFusedChunkBlocks;
  if (mask.xcd_rows) {
    const int64_t per = (static_cast<int64_t>(gridDim.x) - kFusedChunkBlocks) / 8;
    rb = (rb % 8) * per + rb / 8;
  }
  if (mask.mix_xcds) rb ^= static_cast<int64_t>((static_cast<uint32_t>(rb >> 3) * 0x9E3779B1u) >> 29);  // grid: whole groups of 8
  const int64_t row = rb * kWaves + wave;
  if (row >= M) return;
  const int64_t s = rowptr[row];
  const int64_t e = rowptr[row + 1];
  if (e - s > kLongRow) return;  // on the list: chunk role + combine write it
  if ((MODE == M_MASK || MODE == M_CSC) && mask.mat) mask.mrow = mask.mat + row * K + k0;
  reduce_edge_range<VEC, LPR, RED, U, MODE, AW, HOT>(col, val, mat + k0, K, kact, s, e, nnz, lane, acc, arg, mask);
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
      if (MODE == M_NOARG || arg_out == nullptr) {
        // caller wants `out` (and maybe the byte form) only
      } else if (mask.temporal_out) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) arg_out[row * K + k0 + i] = arg[i];
      } else {
        store_arg_nt<VEC>(arg_out + row * K + k0, arg);
      }
      if constexpr (MODE != M_NOARG && VEC == 4) {
        if (mask.arg_bytes_out) {  // the backward's row-local form, for free while arg is in registers
          uint32_t f[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) f[i] = arg_local(arg[i] - s, deg, mask.arg_width);
          store_arg_local4(mask.arg_bytes_out, row * K + k0, f, mask.arg_width);
        }
      }
    }
    if (mask.temporal_out) store_vec<VEC>(out.p + row * out.ld + k0, acc);
    else store_vec_nt<VEC>(out.p + row * out.ld + k0, acc);
  }
}

// One wave per long row: fold its chunks' partials in chunk order.
template <int RED>
__global__ void __launch_bounds__(psa::kLongThreads)
spmm_long_combine_kernel(const int64_t* __restrict__ rowptr, int64_t K, int mean,
                         const unsigned long long* __restrict__ long_ctr,
                         const LongEntry* __restrict__ long_list,
                         const float* __restrict__ part_val,
                         const int64_t* __restrict__ part_arg,
                         OutView out, int64_t* __restrict__ arg_out,
                         uint8_t* __restrict__ arg_bytes, int arg_width) {
  const int lane = threadIdx.x & 63;
  const int nrows = static_cast<int>(*long_ctr >> 32);
  const int num_waves = gridDim.x * (blockDim.x >> 6);
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nrows; r += num_waves) {
    const LongEntry ent = long_list[r];
    const int64_t deg = rowptr[ent.row + 1] - rowptr[ent.row];
    // two K positions per lane and trip (k and k + 64): the chunk loop of a hub row
    // (a 40 000-edge row has > 300 chunks) is the kernel's critical path, and
    // K = 128 would otherwise walk it twice, one after the other
    for (int64_t kb = lane; kb < K; kb += 128) {
      const int64_t p0 = static_cast<int64_t>(ent.first_chunk) * K + kb;
      const bool two = kb + 64 < K;
      float acc[2];
      int64_t arg[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool on = t == 0 || two;
        acc[t] = on ? part_val[p0 + 64 * t] : 0.f;
        arg[t] = (RED == R_SUM || !on) ? 0 : part_arg[p0 + 64 * t];
      }
      // 8 partials per K position requested per step, folded in chunk order
      for (uint32_t c = 1; c < ent.num_chunks; c += 8) {
        float x[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            x[t][i] = (c + i < ent.num_chunks && (t == 0 || two))
                          ? part_val[p0 + 64 * t + static_cast<int64_t>(c + i) * K] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (c + i >= ent.num_chunks || (t == 1 && !two)) break;
            if (RED == R_SUM) {
              acc[t] += x[t][i];
            } else if (RED == R_MAX ? (x[t][i] > acc[t]) : (x[t][i] < acc[t])) {
              acc[t] = x[t][i];
              arg[t] = part_arg[p0 + 64 * t + static_cast<int64_t>(c + i) * K];
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
        const int64_t k = kb + 64 * t;
        if (RED == R_SUM) {
          if (mean) acc[t] = acc[t] / static_cast<float>(deg);
        } else {
          if (arg_out) __builtin_nontemporal_store(arg[t], arg_out + ent.row * K + k);
          if (arg_bytes) store_arg_local1(arg_bytes, ent.row * K + k, arg_local(arg[t] - rowptr[ent.row], deg, arg_width), arg_width);
        }
        __builtin_nontemporal_store(acc[t], out.p + ent.row * out.ld + k);
      }
    }
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
                     OutView out, int64_t* __restrict__ arg_out,
                     int64_t M, int64_t K, int64_t nnz, int mean,
                     unsigned long long* __restrict__ long_ctr,
                     LongEntry* __restrict__ long_list, uint8_t* __restrict__ arg_bytes, int arg_width) {
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
  if (long_list && e - s > kLongRow) {  // the whole lane group leaves together
    if (l == 0) push_long_row(long_ctr, long_list, row, e - s);
    return;
  }
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
    if (arg_out) store_arg_nt<VEC>(arg_out + row * K + k0, arg);
    if constexpr (VEC == 4) {
      if (arg_bytes) {  // row-local form for the one-pass backward (see MaskArgs)
        uint32_t f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = arg_local(arg[i] - s, deg, arg_width);
        store_arg_local4(arg_bytes, row * K + k0, f, arg_width);
      }
    }
  }
  store_vec_nt<VEC>(out.p + row * out.ld + k0, acc);
}

int g_variant = 0;

// Device scratch of the long-row path (carved out of the caller's workspace).
struct LongScratch {
  unsigned long long* ctr = nullptr;  // {rows << 32 | chunks}, zeroed every call
  LongEntry* list = nullptr;
  float* part_val = nullptr;
  int64_t* part_arg = nullptr;
};

using psa::align256;

size_t long_workspace_bytes(bool minmax, int64_t K, int64_t nnz) {
  const size_t chunks = static_cast<size_t>(max_long_chunks(nnz));
  return psa::long_list_bytes(nnz) + align256(chunks * K * sizeof(float)) +
         (minmax ? align256(chunks * K * sizeof(int64_t)) : 0);
}

LongScratch carve(void* workspace, bool minmax, int64_t K, int64_t nnz) {
  LongScratch w;
  char* p = static_cast<char*>(workspace);
  w.ctr = reinterpret_cast<unsigned long long*>(p);
  p += 256;
  w.list = reinterpret_cast<LongEntry*>(p);
  p += align256(sizeof(LongEntry) * static_cast<size_t>(max_long_rows(nnz)));
  w.part_val = reinterpret_cast<float*>(p);
  p += align256(static_cast<size_t>(max_long_chunks(nnz)) * K * sizeof(float));
  if (minmax) w.part_arg = reinterpret_cast<int64_t*>(p);
  return w;
}

// Second and third launch of the long-row path (no-ops when the list is empty).
template <int VEC, int LPR, int U>
int launch_long(int red, const int64_t* rowptr, const int64_t* col, const float* val,
                const float* mat, OutView out, int64_t* arg_out, int64_t K, int64_t nnz,
                int mean, const LongScratch& w, hipStream_t s, ArgLocal arg_bytes = {}) {
  const dim3 grid(kLongBlocks), block(psa::kLongThreads);
#define PSA_LONG(R)                                                                        \
  do {                                                                                     \
    hipLaunchKernelGGL((spmm_long_chunk_kernel<VEC, LPR, R, U>), grid, block, 0, s, rowptr, \
                       col, val, mat, K, nnz, w.ctr, w.list, w.part_val, w.part_arg);      \
    hipLaunchKernelGGL((spmm_long_combine_kernel<R>), grid, block, 0, s, rowptr, K,         \
                       mean, w.ctr, w.list, w.part_val, w.part_arg, out, arg_out,          \
                       arg_bytes.p, arg_bytes.width);                                      \
  } while (0)
  if (red == R_SUM) PSA_LONG(R_SUM);
  else if (red == R_MIN) PSA_LONG(R_MIN);
  else PSA_LONG(R_MAX);
#undef PSA_LONG
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

// Fused long-row path: pre-pass list, one launch with both roles, combine.
constexpr int64_t kNtGatherBytes = 6ll << 30;  // dense operand size from which its rows are gathered non-temporally

template <int VEC, int LPR, int U>
int launch_fused(int red, const int64_t* rowptr, const int64_t* col, const float* val,
                 const float* mat, OutView out, int64_t* arg_out, int64_t M, int64_t K,
                 int64_t nnz, int mean, const LongScratch& w, hipStream_t s,
                 ArgLocal arg_bytes = {}, bool nt_gather = false, int k_tiles = 1) {
  const int kFusedChunkBlocks = g_variant == 20 ? 512 : g_variant == 21 ? 1024 : g_variant == 22 ? 1536 : kFusedChunkBlocksDefault;
  const int64_t gx = psa::ceil_div(psa::ceil_div(M, kWaves), 8) * 8 + kFusedChunkBlocks;
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  PSA_REQUIRE(k_tiles >= 1 && k_tiles <= 65535, "too many K tiles");
  const dim3 block(kThreads), grid(static_cast<unsigned>(gx), static_cast<unsigned>(k_tiles));
  hipLaunchKernelGGL(psa::find_long_rows_kernel, dim3(static_cast<unsigned>(psa::ceil_div(M, psa::kFindThreads * psa::kFindIters))),
                     block, 0, s, rowptr, M, w.ctr, w.list);
  const dim3 cgrid(kLongBlocks), cblock(psa::kLongThreads);
  MaskArgs plain;
  plain.xcd_rows = g_variant == 16;
  plain.mix_xcds = g_variant == 28;  // A/B: XCD mixing of the row blocks in the forward too (tools/archive/mix_xcds_fwd.py)
  plain.temporal_out = g_variant == 17;
  plain.nt_gather = nt_gather;
  plain.chunk_blocks = kFusedChunkBlocks;
  plain.arg_bytes_out = arg_bytes.p;
  plain.arg_width = arg_bytes.width;
  // min/max with neither arg_out nor arg_bytes wanted: the instantiation that does
  // not track the winners' edge ids (the combine still folds chunk partials by
  // value; ids it reads there only break ties between equal values)
  const bool no_arg = red != R_SUM && arg_out == nullptr && arg_bytes.p == nullptr && g_variant != 19;
  auto launch = [&](auto red_tag) {
    constexpr int R = decltype(red_tag)::value;
    bool done = false;
    if constexpr (R != R_SUM) {
      if (no_arg) {
        hipLaunchKernelGGL((spmm_fused_kernel<VEC, LPR, R, U, M_NOARG>), grid, block, 0, s, rowptr, col, val, mat, out,
                           arg_out, M, K, nnz, mean, w.ctr, w.list, w.part_val, w.part_arg, plain);
        done = true;
      }
    }
    if (!done)
      hipLaunchKernelGGL((spmm_fused_kernel<VEC, LPR, R, U>), grid, block, 0, s, rowptr, col, val, mat, out, arg_out, M,
                         K, nnz, mean, w.ctr, w.list, w.part_val, w.part_arg, plain);
    hipLaunchKernelGGL((spmm_long_combine_kernel<R>), cgrid, cblock, 0, s, rowptr, K, mean, w.ctr, w.list, w.part_val,
                       w.part_arg, out, arg_out, plain.arg_bytes_out, plain.arg_width);
  };
  if (red == R_SUM) launch(std::integral_constant<int, R_SUM>{});
  else if (red == R_MIN) launch(std::integral_constant<int, R_MIN>{});
  else launch(std::integral_constant<int, R_MAX>{});
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

// The same three launches in the MASK form (sum over the CSC view; see MaskArgs).
template <int LPR, int U, int MODE>
int launch_fused_masked(const int64_t* colptr, const int64_t* row_csc, const float* value,
                        const float* grad, OutView out, int64_t N, int64_t K, int64_t nnz,
                        const MaskArgs& mask_in, const LongScratch& w, hipStream_t s) {
  const int kFusedChunkBlocks = g_variant == 20 ? 512 : g_variant == 21 ? 1024 : g_variant == 22 ? 1536 : g_variant == 23 ? 2048 : g_variant == 24 ? 256 : g_variant == 29 ? 768 : kMaskedChunkBlocks;
  const int64_t gx = psa::ceil_div(psa::ceil_div(N, kWaves), 8) * 8 + kFusedChunkBlocks;  // whole groups of 8 row blocks
  PSA_REQUIRE(gx <= 0x7fffffff, "N too large for one launch");
  const dim3 block(kThreads), grid(static_cast<unsigned>(gx));
  MaskArgs mask = mask_in;
  mask.mix_xcds = g_variant != 27;
  mask.chunk_blocks = kFusedChunkBlocks;
  hipLaunchKernelGGL(psa::find_long_rows_kernel, dim3(static_cast<unsigned>(psa::ceil_div(N, psa::kFindThreads * psa::kFindIters))),
                     block, 0, s, colptr, N, w.ctr, w.list);
  // instantiations by what the pass needs: the one-byte form keeps its second phase (exact test
  // against arg_out), the two-byte form and the hub-row copies have none of it — 76 VGPRs instead
  // of 96 where they are not needed, one more wave per SIMD
  const bool hot = mask.hot != nullptr;
#define PSA_MASKED(AW, HOT)                                                                                           \
  hipLaunchKernelGGL((spmm_fused_kernel<4, LPR, R_SUM, U, MODE, AW, HOT>), grid, block, 0, s, colptr, row_csc, value, \
                     grad, out, static_cast<int64_t*>(nullptr), N, K, nnz, 0, w.ctr, w.list, w.part_val, w.part_arg, mask)
  if constexpr (MODE == M_MASK) {
    if (mask.arg_width == 2) {
      if (hot) PSA_MASKED(2, true);
      else PSA_MASKED(2, false);
    } else {
      PSA_REQUIRE(!hot, "hub-row copies go with the two-byte row-local form");
      if (mask.arg != nullptr) PSA_MASKED(1, false);
      else PSA_MASKED(0, false);
    }
  } else {
    if (hot) PSA_MASKED(1, true);
    else PSA_MASKED(1, false);
  }
#undef PSA_MASKED
  hipLaunchKernelGGL((spmm_long_combine_kernel<R_SUM>), dim3(kLongBlocks), dim3(psa::kLongThreads), 0, s,
                     colptr, K, 0, w.ctr, w.list, w.part_val, w.part_arg, out,
                     static_cast<int64_t*>(nullptr), static_cast<uint8_t*>(nullptr), 1);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

// bytes[r, k] = arg_out[r, k] as an index local to row r, mod 128, bit 7 set for
// rows of more than 128 edges (empty rows: whatever, no edge ever asks).  Four
// elements per thread: 32 B in, 4 B out.
__global__ void __launch_bounds__(kThreads)
minmax_compress_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ arg_out,
                       int64_t M, int64_t K, uint8_t* __restrict__ bytes, int width) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;  // group of 4
  const int64_t kq = K / 4;
  if (q >= M * kq) return;
  const int64_t r = q / kq;
  const int64_t start = rowptr[r];
  const int64_t deg = rowptr[r + 1] - start;
  const longlong2 a01 = *reinterpret_cast<const longlong2*>(arg_out + 4 * q);
  const longlong2 a23 = *reinterpret_cast<const longlong2*>(arg_out + 4 * q + 2);
  const int64_t a[4] = {a01.x, a01.y, a23.x, a23.y};
  uint32_t f[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = arg_local(a[i] - start, deg, width);
  store_arg_local4(bytes, 4 * q, f, width);
}

// tag[j] = index of CSC edge j inside its CSR row, mod 128, bit 7 set for rows of more than 128 edges.
__global__ void __launch_bounds__(kThreads)
csc_edge_tags_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ row_csc,
                     const int64_t* __restrict__ csr2csc, int64_t nnz,
                     uint8_t* __restrict__ tag, int width) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (j >= nnz) return;
  const int64_t r = row_csc[j];
  const int64_t start = rowptr[r];
  store_arg_local1(tag, j, arg_local(csr2csc[j] - start, rowptr[r + 1] - start, width), width);
}

template <int VEC, int LPR, int U>
int launch_multirow(int red, const int64_t* rowptr, const int64_t* col,
                    const float* val, const float* mat, OutView out,
                    int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
                    const LongScratch& w, hipStream_t s, ArgLocal arg_bytes = {}) {
  const int64_t gx = psa::ceil_div(M, static_cast<int64_t>(kWaves) * (64 / LPR));
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx)), block(kThreads);
  if (red == R_SUM) {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_SUM, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list,
                       static_cast<uint8_t*>(nullptr), 1);
  } else if (red == R_MIN) {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_MIN, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list, arg_bytes.p, arg_bytes.width);
  } else {
    hipLaunchKernelGGL((spmm_multirow_kernel<VEC, LPR, R_MAX, U>), grid, block, 0, s,
                       rowptr, col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list, arg_bytes.p, arg_bytes.width);
  }
  PSA_LAUNCH_CHECK();
  // chunk waves split a row's edges over the 64/LPR lane groups: 8 edges per step
  constexpr int UL = (64 / LPR) >= 8 ? 1 : 8 / (64 / LPR);
  if (w.list)
    return launch_long<VEC, LPR, UL>(red, rowptr, col, val, mat, out, arg_out, K, nnz, mean, w, s, arg_bytes);
  return PSA_OK;
}

template <int VEC, int LPR, int U>
int launch_row(int red, const int64_t* rowptr, const int64_t* col,
               const float* val, const float* mat, OutView out,
               int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
               const LongScratch& w, hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, kWaves);
  const int64_t gy = psa::ceil_div(K, static_cast<int64_t>(LPR) * VEC);
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  PSA_REQUIRE(gy <= 65535, "K too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  const dim3 block(kThreads);
  if (red == R_SUM) {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_SUM, U>), grid, block, 0, s, rowptr, col,
                       val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  } else if (red == R_MIN) {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_MIN, U>), grid, block, 0, s, rowptr, col,
                       val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  } else {
    hipLaunchKernelGGL((spmm_row_kernel<VEC, LPR, R_MAX, U>), grid, block, 0, s, rowptr, col,
                       val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  }
  PSA_LAUNCH_CHECK();
  if (w.list) return launch_long<VEC, LPR, U>(red, rowptr, col, val, mat, out, arg_out, K, nnz, mean, w, s);
  return PSA_OK;
}

template <int VEC, int LPR, int U, int R>
int launch_rows(int red, const int64_t* rowptr, const int64_t* col,
                const float* val, const float* mat, OutView out,
                int64_t* arg_out, int64_t M, int64_t K, int64_t nnz, int mean,
                const LongScratch& w, hipStream_t s) {
  const int64_t gx = psa::ceil_div(M, static_cast<int64_t>(kWaves) * R);
  const int64_t gy = psa::ceil_div(K, static_cast<int64_t>(LPR) * VEC);
  PSA_REQUIRE(gx <= 0x7fffffff, "M too large for one launch");
  PSA_REQUIRE(gy <= 65535, "K too large for one launch");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  const dim3 block(kThreads);
  if (red == R_SUM) {
    hipLaunchKernelGGL((spmm_rows_kernel<VEC, LPR, R_SUM, U, R>), grid, block, 0, s, rowptr,
                       col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  } else if (red == R_MIN) {
    hipLaunchKernelGGL((spmm_rows_kernel<VEC, LPR, R_MIN, U, R>), grid, block, 0, s, rowptr,
                       col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  } else {
    hipLaunchKernelGGL((spmm_rows_kernel<VEC, LPR, R_MAX, U, R>), grid, block, 0, s, rowptr,
                       col, val, mat, out, arg_out, M, K, nnz, mean, w.ctr, w.list);
  }
  PSA_LAUNCH_CHECK();
  if (w.list) return launch_long<VEC, LPR, U>(red, rowptr, col, val, mat, out, arg_out, K, nnz, mean, w, s);
  return PSA_OK;
}

}  // namespace

extern "C" {

int psa_spmm_set_variant(int variant) {
  const int prev = g_variant;
  g_variant = variant;
  return prev;
}

int psa_csc_edge_tags(const int64_t* rowptr, const int64_t* row_csc, const int64_t* csr2csc,
                      int64_t nnz, void* tag, int width, psa_stream_t stream) {
  PSA_REQUIRE(nnz >= 0, "negative size");
  PSA_REQUIRE(width == 1 || width == 2, "width must be 1 or 2");
  if (nnz == 0) return PSA_OK;
  PSA_REQUIRE(rowptr && row_csc && csr2csc && tag, "NULL pointer");
  const int64_t blocks = psa::ceil_div(nnz, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "nnz too large for one launch");
  hipLaunchKernelGGL(csc_edge_tags_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     psa::as_stream(stream), rowptr, row_csc, csr2csc, nnz, static_cast<uint8_t*>(tag), width);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

size_t psa_spmm_minmax_bw_csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
  if (M <= 0 || K <= 0) return 256;
  return align256(2 * static_cast<size_t>(M) * K) + long_workspace_bytes(false, K, nnz > 0 ? nnz : 1);
}

int psa_spmm_minmax_bw_csc(const int64_t* rowptr, const int64_t* colptr,
                           const int64_t* row_csc, const int64_t* csr2csc,
                           const void* tag, const float* value, const float* mat,
                           const float* grad, const int64_t* arg_out, const void* arg_bytes, int arg_width,
                           const float* hot_grad, const void* hot_bytes, int64_t num_hot,
                           int64_t M, int64_t N, int64_t K, int64_t nnz, float* grad_value,
                           float* grad_mat, void* workspace, size_t workspace_bytes,
                           psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (N == 0 || K == 0) return PSA_OK;
  if (K % 4 != 0 || K > 256 || !psa::aligned(grad, 16) || !psa::aligned(grad_mat, 16) ||
      !psa::aligned(arg_out, 16) || !psa::aligned(mat, 16)) {
    psa::set_error("psa_spmm_minmax_bw_csc: needs K % 4 == 0, K <= 256 and 16-byte aligned "
                   "operands (use psa_spmm_minmax_bw)");
    return PSA_ERR_UNSUPPORTED;
  }
  PSA_REQUIRE(colptr && grad_mat, "NULL pointer");
  PSA_REQUIRE(nnz == 0 || (rowptr && row_csc && tag && grad), "NULL pointer");
  PSA_REQUIRE(nnz == 0 || csr2csc != nullptr || arg_out == nullptr,
              "csr2csc = NULL (value in CSC order) goes with the exact arg_bytes forms only: arg_out is compared with CSR edge ids");
  PSA_REQUIRE(nnz == 0 || arg_out != nullptr || arg_bytes != nullptr, "arg_out and arg_bytes are both NULL");
  PSA_REQUIRE(arg_width == 1 || arg_width == 2, "arg_width must be 1 or 2");
  PSA_REQUIRE(arg_bytes == nullptr || psa::aligned(arg_bytes, 4 * arg_width), "arg_bytes alignment");
  PSA_REQUIRE(num_hot >= 0 && (num_hot == 0 || (hot_grad != nullptr && hot_bytes != nullptr)), "hot rows are NULL");
  PSA_REQUIRE(num_hot == 0 || (arg_bytes != nullptr && arg_out == nullptr && arg_width == 2 && psa::aligned(hot_grad, 16) &&
                               psa::aligned(hot_bytes, 8)),
              "hot rows go with the two-byte arg_bytes (no arg_out) and 16-byte aligned copies");
  PSA_REQUIRE(grad_value == nullptr || mat != nullptr || nnz == 0, "grad_value needs mat");
  PSA_REQUIRE(max_long_chunks(nnz) < (1ll << 32), "too many chunks");
  if (workspace == nullptr || workspace_bytes < psa_spmm_minmax_bw_csc_workspace_bytes(M, K, nnz)) {
    psa::set_error("psa_spmm_minmax_bw_csc: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
  hipStream_t s = psa::as_stream(stream);
  uint8_t* bytes = static_cast<uint8_t*>(workspace);
  const LongScratch w = carve(bytes + align256(2 * static_cast<size_t>(M) * K), false, K, nnz > 0 ? nnz : 1);
  PSA_ZERO(w.ctr, 8, s);
  if (arg_bytes == nullptr && M > 0 && nnz > 0) {  // the forward did not leave the row-local form behind
    const int64_t blocks = psa::ceil_div(M * (K / 4), kThreads);
    PSA_REQUIRE(blocks <= 0x7fffffff, "M*K too large for one launch");
    hipLaunchKernelGGL(minmax_compress_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                       s, rowptr, arg_out, M, K, bytes, arg_width);
  }
  MaskArgs mask;
  mask.temporal_out = g_variant == 17;
  mask.bytes = arg_bytes != nullptr ? static_cast<const uint8_t*>(arg_bytes) : bytes;
  mask.arg_width = arg_width;
  mask.tag = static_cast<const uint8_t*>(tag);
  if (num_hot > 0) {
    mask.hot = hot_grad;
    mask.hot_bytes = static_cast<const uint8_t*>(hot_bytes);
    mask.hot_first = M;
  }
  mask.edge_id = csr2csc;
  mask.arg = arg_out;
  if (grad_value != nullptr && nnz > 0) {
    mask.mat = mat;
    mask.grad_value = grad_value;
  }
  const int64_t q = K / 4;
  if (q <= 4) return launch_fused_masked<4, 1, M_MASK>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 8) return launch_fused_masked<8, 2, M_MASK>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 16) return launch_fused_masked<16, 4, M_MASK>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 32) return launch_fused_masked<32, 4, M_MASK>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  return launch_fused_masked<64, 8, M_MASK>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
}

size_t psa_spmm_minmax_bw_eb_workspace_bytes(int64_t K, int64_t nnz) {
  if (K <= 0 || nnz <= 0 || K % 4 != 0) return 256;
  return psa::eb_workspace_bytes(false, K, nnz);
}

int psa_spmm_minmax_bw_eb(const int64_t* colptr, const int64_t* col_csc, const int64_t* row_csc, const void* tag,
                          const float* weight_csc, const float* grad, const void* arg_bytes, int arg_width,
                          const float* hot_grad, const void* hot_bytes, int64_t num_hot, int64_t M, int64_t N, int64_t K,
                          int64_t nnz, float* grad_mat, void* workspace, size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  PSA_REQUIRE(arg_width == 1 || arg_width == 2, "arg_width must be 1 or 2");
  if (N == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(colptr != nullptr && grad_mat != nullptr, "NULL pointer");
  if (nnz == 0) {  // no entries: grad_mat = 0
    return psa::zero_async(grad_mat, sizeof(float) * static_cast<size_t>(N) * K, psa::as_stream(stream));
  }
  if (K % 4 != 0 || M >= (1ll << 31) || !psa::aligned(grad, 16) || !psa::aligned(grad_mat, 16) || !psa::eb_supported(N, K, nnz)) {
    psa::set_error("psa_spmm_minmax_bw_eb: needs K % 4 == 0, 31-bit ids and 16-byte aligned operands (use psa_spmm_minmax_bw_csc)");
    return PSA_ERR_UNSUPPORTED;
  }
  PSA_REQUIRE(row_csc && tag && grad && arg_bytes, "NULL pointer");
  PSA_REQUIRE(num_hot >= 0 && (num_hot == 0 || (hot_grad != nullptr && hot_bytes != nullptr)), "hot rows are NULL");
  const psa::EbMask mask{arg_bytes, hot_bytes, tag, arg_width};
  // the CSC view as a CSR matrix: rows = columns of A (N of them), gathered operand = grad ([M, K])
  return psa::launch_spmm_eb(R_SUM, 0, colptr, col_csc, row_csc, weight_csc, grad, grad_mat, K, nullptr, nullptr, 1, N, M, K,
                             nnz, hot_grad, num_hot, workspace, workspace_bytes, false, 0, 0, psa::as_stream(stream), 0, &mask);
}

size_t psa_spmm_sum_bw_csc_workspace_bytes(int64_t K, int64_t nnz) {
  return long_workspace_bytes(false, K > 0 ? K : 1, nnz > 0 ? nnz : 1);
}

int psa_spmm_sum_bw_csc(const int64_t* colptr, const int64_t* row_csc, const int64_t* csr2csc,
                        const float* value, const float* row_scale, const float* mat,
                        const float* grad, const float* hot_grad, int64_t num_hot, int64_t M,
                        int64_t N, int64_t K, int64_t nnz,
                        float* grad_value, float* grad_mat, void* workspace,
                        size_t workspace_bytes, psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  PSA_REQUIRE(num_hot >= 0 && (num_hot == 0 || (hot_grad != nullptr && psa::aligned(hot_grad, 16))), "hot_grad");
  if (N == 0 || K == 0) return PSA_OK;
  if (K % 4 != 0 || K > 256 || !psa::aligned(grad, 16) || !psa::aligned(grad_mat, 16) ||
      !psa::aligned(mat, 16)) {
    psa::set_error("psa_spmm_sum_bw_csc: needs K % 4 == 0, K <= 256 and 16-byte aligned operands "
                   "(use psa_spmm_value_bw + psa_transpose_weights + psa_spmm)");
    return PSA_ERR_UNSUPPORTED;
  }
  PSA_REQUIRE(colptr && grad_mat, "NULL pointer");
  PSA_REQUIRE(nnz == 0 || (row_csc && grad), "NULL pointer");  // csr2csc = NULL: value is in CSC order already
  PSA_REQUIRE(grad_value == nullptr || mat != nullptr || nnz == 0, "grad_value needs mat");
  PSA_REQUIRE(max_long_chunks(nnz) < (1ll << 32), "too many chunks");
  if (workspace == nullptr || workspace_bytes < psa_spmm_sum_bw_csc_workspace_bytes(K, nnz)) {
    psa::set_error("psa_spmm_sum_bw_csc: workspace too small");
    return PSA_ERR_WORKSPACE;
  }
  PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
  hipStream_t s = psa::as_stream(stream);
  const LongScratch w = carve(workspace, false, K, nnz > 0 ? nnz : 1);
  PSA_ZERO(w.ctr, 8, s);
  MaskArgs mask;
  mask.temporal_out = g_variant == 17;
  mask.edge_id = csr2csc;
  mask.row_scale = row_scale;
  if (num_hot > 0) {
    mask.hot = hot_grad;
    mask.hot_first = M;
  }
  if (grad_value != nullptr && nnz > 0) {
    mask.mat = mat;
    mask.grad_value = grad_value;
  }
  const int64_t q = K / 4;
  if (q <= 4) return launch_fused_masked<4, 1, M_CSC>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 8) return launch_fused_masked<8, 2, M_CSC>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 16) return launch_fused_masked<16, 4, M_CSC>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  if (q <= 32) return launch_fused_masked<32, 4, M_CSC>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
  return launch_fused_masked<64, 8, M_CSC>(colptr, row_csc, value, grad, OutView{grad_mat, K}, N, K, nnz, mask, w, s);
}

size_t psa_spmm_workspace_bytes(int reduce, int64_t K, int64_t nnz) {
  if (K <= 0 || nnz <= 0) return 0;
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  const size_t eb = K % 4 == 0 ? psa::eb_workspace_bytes(minmax, K, nnz) : 0;
  const size_t lr = nnz > kLongRow ? long_workspace_bytes(minmax, K, nnz) : 0;  // else no row can be long
  return eb > lr ? eb : lr;
}

}  // extern "C"

namespace {

// psa_spmm proper; *bytes_done tells whether the kernel that ran also stored
// arg_bytes (only the fused-roles path does, the caller compresses otherwise).
int spmm_dispatch(int reduce, const int64_t* rowptr, const int64_t* row, const int64_t* col,
                  const float* value, const float* mat, int64_t M, int64_t N,
                  int64_t K, int64_t nnz, OutView out, int64_t* arg_out,
                  void* workspace, size_t workspace_bytes, hipStream_t s,
                  ArgLocal arg_bytes, bool* bytes_done, int algo, const float* hot_rows = nullptr,
                  int64_t num_hot = 0) {
  PSA_REQUIRE(reduce >= PSA_SUM && reduce <= PSA_MAX, "bad reduce");
  PSA_REQUIRE(num_hot >= 0 && (num_hot == 0 || hot_rows != nullptr), "hot_rows is NULL");
  PSA_REQUIRE(algo >= PSA_SPMM_AUTO && algo <= PSA_SPMM_EDGE_RANGES, "bad algo");
  PSA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nnz >= 0, "negative size");
  if (M == 0 || K == 0) return PSA_OK;
  PSA_REQUIRE(rowptr != nullptr, "rowptr is NULL");
  PSA_REQUIRE(out.p != nullptr, "out is NULL");
  PSA_REQUIRE(out.ld >= K, "ldo must be at least K");
  PSA_REQUIRE(nnz == 0 || (col != nullptr && mat != nullptr),
              "col/mat is NULL");
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  const int red = reduce == PSA_MIN ? R_MIN : (reduce == PSA_MAX ? R_MAX : R_SUM);
  const int mean = reduce == PSA_MEAN;

  // Edge-balanced path (spmm_eb.hip): ranges of consecutive edges instead of rows.
  // Variants 30-33 force it (31-33 with ranges of 128 / 512 / 1024 edges).
  const bool eb_forced = (g_variant >= 30 && g_variant <= 36) || (algo == PSA_SPMM_EDGE_RANGES && g_variant == 0);
  if (eb_forced && workspace != nullptr && N < (1ll << 31) && nnz > 0 && K % 4 == 0 && psa::aligned(mat, 16) &&
      psa::aligned(out.p, 16) && out.ld % 4 == 0 && psa::eb_supported(M, K, nnz)) {
    *bytes_done = arg_bytes.p != nullptr && minmax;
    const bool nt_gather = N * K * 4 >= kNtGatherBytes;
    const int range_len = g_variant == 31 ? 128 : g_variant == 32 ? 512 : g_variant == 33 ? 1024 : 0;
    return psa::launch_spmm_eb(red, mean, rowptr, row, col, value, mat, out.p, out.ld, arg_out,
                               minmax ? arg_bytes.p : nullptr, arg_bytes.width, M, N, K, nnz, hot_rows, num_hot, workspace,
                               workspace_bytes,
                               nt_gather, range_len, g_variant == 34 ? 1 : g_variant == 35 ? 2 : g_variant == 36 ? 4 : 0, s);
  }

  if (num_hot > 0) {
    psa::set_error("psa_spmm_coo: hot_rows is served by the edge-range kernels only (algo = PSA_SPMM_EDGE_RANGES, "
                   "K % 4 == 0, 16-byte aligned operands, a workspace)");
    return PSA_ERR_UNSUPPORTED;
  }
  // Long-row path: on when the caller brings a workspace (NULL keeps every row
  // on its own wave: same results, slow on power-law graphs).
  LongScratch w;
  if (workspace != nullptr && nnz > kLongRow && g_variant != 10) {
    if (workspace_bytes < long_workspace_bytes(minmax, K, nnz)) {
      psa::set_error("psa_spmm: workspace too small");
      return PSA_ERR_WORKSPACE;
    }
    PSA_REQUIRE(psa::aligned(workspace, 16), "workspace must be 16-byte aligned");
    PSA_REQUIRE(max_long_chunks(nnz) < (1ll << 32), "too many chunks");
    w = carve(workspace, minmax, K, nnz);
    PSA_ZERO(w.ctr, 8, s);
  }

#define PSA_ROW(VEC, LPR, U)                                                 \
  return launch_row<VEC, LPR, U>(red, rowptr, col, value, mat, out, arg_out, \
                                 M, K, nnz, mean, w, s)

  const bool v4 = (K % 4 == 0) && psa::aligned(mat, 16) && psa::aligned(out.p, 16) && out.ld % 4 == 0;
  if (v4) {
    const int64_t q = K / 4;  // float4 per row
    if (g_variant == 2 && K % 128 == 0 && psa::aligned(mat, 8)) PSA_ROW(2, 64, 8);
    if (g_variant == 3 && q >= 32) PSA_ROW(4, 32, 8);
    if (g_variant == 4 && K % 128 == 0) PSA_ROW(2, 64, 16);
#define PSA_MULTI(VEC, LPR, U)                                                     \
  do {                                                                             \
    *bytes_done = arg_bytes.p != nullptr && minmax;                                  \
    return launch_multirow<VEC, LPR, U>(red, rowptr, col, value, mat, out, arg_out, \
                                        M, K, nnz, mean, w, s,                     \
                                        minmax ? arg_bytes : ArgLocal{});           \
  } while (0)
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
#define PSA_ROWS(VEC, LPR, U, R)                                                   \
  return launch_rows<VEC, LPR, U, R>(red, rowptr, col, value, mat, out, arg_out, \
                                     M, K, nnz, mean, w, s)
    // 64 < K <= 256 with a workspace: fused roles (R-MAT scale 21: 2.55 -> 2.34 ms
    // against chunk and row launches back to back; uniform graphs unchanged);
    // variant 15 forces the separate launches
    if ((g_variant == 0 || g_variant == 14 || g_variant == 16 || g_variant == 17 || g_variant == 18 || g_variant == 19 || (g_variant >= 20 && g_variant <= 22) || g_variant == 25 || g_variant == 26 || g_variant == 28) && w.list && q > 16 &&
        (q <= 64 || g_variant != 25)) {
      *bytes_done = arg_bytes.p != nullptr && minmax;
      // A dense operand far beyond the 256 MiB Infinity Cache is gathered with
      // non-temporal loads: nothing of it will be hit again, and not allocating
      // the lines is worth 4-5 % at 8-16 GiB (tools/archive/nt_gather_sweep.py: break-even
      // at ~4 GiB, 0.85x at 1 GiB where a quarter of B does stay cached).
      const bool nt_gather = g_variant == 18 || ((g_variant == 0 || g_variant == 25) && N * K * 4 >= kNtGatherBytes);
      // K >= 192 runs as ceil(K / 128) tiles of the K = 128 form (32 lanes x float4,
      // two edges per gather instruction) over grid.y; measured against one tile of
      // 64 lanes x float4 (K <= 256) or the row kernel (K > 256), 2 M rows / 20 M edges:
      // K = 192 2.67 -> 2.58 ms, 224 3.22 -> 2.88, 256 3.66 -> 3.30 (0.78 -> 0.86 of
      // peak), 320 4.79 -> 4.20, 448 6.37 -> 5.85, 512 7.49 -> 6.75; K = 160 is better
      // off as one tile (2.12 vs 2.45 ms).  col / value are read once per tile (+1 % of
      // the bytes).  Variant 25 keeps the old choices, 26 tiles every K > 128.
      const bool tiled = (K >= 192 && g_variant != 25) || (K > 128 && g_variant == 26);
      if (q <= 32 || tiled)
        return launch_fused<4, 32, 4>(red, rowptr, col, value, mat, out, arg_out, M, K, nnz, mean, w, s,
                                      minmax ? arg_bytes : ArgLocal{}, nt_gather,
                                      static_cast<int>(psa::ceil_div(K, 128)));
      return launch_fused<4, 64, 8>(red, rowptr, col, value, mat, out, arg_out, M, K, nnz, mean, w, s,
                                    minmax ? arg_bytes : ArgLocal{}, nt_gather);
    }
    if (g_variant == 11) { if (q <= 32) PSA_ROWS(4, 32, 4, 4); PSA_ROWS(4, 64, 8, 4); }
    if (g_variant == 12) { if (q <= 32) PSA_ROWS(4, 32, 4, 8); PSA_ROWS(4, 64, 8, 8); }
    if (g_variant == 13) { if (q <= 32) PSA_ROWS(4, 32, 4, 2); PSA_ROWS(4, 64, 8, 2); }
#undef PSA_ROWS
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

}  // namespace

extern "C" {

int psa_spmm(int reduce, const int64_t* rowptr, const int64_t* col,
             const float* value, const float* mat, int64_t M, int64_t N,
             int64_t K, int64_t nnz, float* out, int64_t* arg_out,
             uint8_t* arg_bytes, void* workspace, size_t workspace_bytes,
             psa_stream_t stream) {
  return psa_spmm_coo(reduce, rowptr, nullptr, col, value, mat, nullptr, 0, M, N, K, nnz, out, 0, arg_out, arg_bytes, 1,
                      PSA_SPMM_AUTO, workspace, workspace_bytes, stream);
}

int psa_spmm_coo(int reduce, const int64_t* rowptr, const int64_t* row, const int64_t* col,
                 const float* value, const float* mat, const float* hot_rows, int64_t num_hot, int64_t M, int64_t N,
                 int64_t K, int64_t nnz, float* out, int64_t ldo, int64_t* arg_out,
                 void* arg_bytes_v, int arg_width, int algo, void* workspace, size_t workspace_bytes,
                 psa_stream_t stream) {
  const bool minmax = reduce == PSA_MIN || reduce == PSA_MAX;
  uint8_t* arg_bytes = static_cast<uint8_t*>(arg_bytes_v);
  PSA_REQUIRE(arg_bytes == nullptr || arg_width == 1 || arg_width == 2, "arg_width must be 1 or 2");
  if (arg_bytes != nullptr && minmax && K % 4 != 0) {
    psa::set_error("psa_spmm: arg_bytes needs K % 4 == 0");
    return PSA_ERR_UNSUPPORTED;
  }
  hipStream_t s = psa::as_stream(stream);
  bool bytes_done = false;
  const int st = spmm_dispatch(reduce, rowptr, row, col, value, mat, M, N, K, nnz, OutView{out, ldo > 0 ? ldo : K}, arg_out, workspace,
                               workspace_bytes, s, ArgLocal{arg_bytes, arg_width}, &bytes_done, algo, hot_rows, num_hot);
  if (st != PSA_OK || arg_bytes == nullptr || !minmax || bytes_done || M == 0 || K == 0) return st;
  // the kernel that ran keeps arg_out only: one more pass turns it into bytes
  PSA_REQUIRE(arg_out != nullptr, "arg_bytes without arg_out needs a K tile whose kernel writes the bytes itself (K % 4 == 0, K <= 256)");
  PSA_REQUIRE(psa::aligned(arg_out, 16) && psa::aligned(arg_bytes, 4 * arg_width), "arg_out / arg_bytes alignment");
  const int64_t blocks = psa::ceil_div(M * (K / 4), kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "M*K too large for one launch");
  hipLaunchKernelGGL(minmax_compress_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s,
                     rowptr, arg_out, M, K, arg_bytes, arg_width);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
