// Cross-lane sums for the SpMM backward kernels (gfx950): DPP row operations — fused into the add, v_add_f32_dpp — for
// partners 1, 2, 4 and 8 lanes away, and a choice between ds_bpermute and the v_permlane16/32_swap pair for 16 and 32.
// The DPP form replaces {ds_bpermute, v_add} by one v_add_f32_dpp: never more VALU work, less on the LDS pipe, fewer
// registers (half-width pass over the CSC view 78 -> 70 VGPRs, 6 -> 7 waves per SIMD, 1.27 -> 1.15 ms at config 3; fp32
// sum pass on R-MAT 21 1.89 -> 1.78 ms, profiles/r04_spmm_dpp_ab.txt).  The permlane pair is NOT free: see add_xor16.
// A first form that put EVERY fold of the fp32 passes into the VALU — permlane swaps for 16 / 32 and for the lane groups'
// sums at the end of every row — took the sum pass on R-MAT 21 from 1.91 to 2.72 ms (profiles/r04_fold_ab.txt, commit
// 0290f97 with a -DPSA_SHFL_FOLDS build as the other arm): with the hub rows in cache-resident copies that pass is bound
// by VALU issue, and most of its rows are empty or tiny, so the per-row fold is what it executes most.
#pragma once

#include <hip/hip_runtime.h>

namespace psa {

template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// the value lane ^ BIT holds, for BIT = 1, 2 (quad permutes), 8 (row of 16 rotated by 8)
template <int BIT>
__device__ __forceinline__ float lane_xor(float x) {
  static_assert(BIT == 1 || BIT == 2 || BIT == 8, "exact partners in the VALU: 1, 2, 8");
  if constexpr (BIT == 1) return dpp_move<0xB1>(x);   // quad_perm [1, 0, 3, 2]
  if constexpr (BIT == 2) return dpp_move<0x4E>(x);   // quad_perm [2, 3, 0, 1]
  return dpp_move<0x128>(x);                          // row_ror:8
}

// x + x(lane ^ 16) / x + x(lane ^ 32), two ways.  SWAP = false: ds_bpermute.  SWAP = true: v_permlane16/32_swap, which
// exchanges the halves of TWO registers in place — a sum costs two copies, the swap and the add, four VALU instructions
// where the shuffle costs one (the add) plus one on the LDS pipe.  Which is faster depends on the kernel's busiest unit
// (same box, profiles/r04_fold_hybrid_ab.txt): the half-width kernels are bound by VALU issue and lose with the swap
// (pass over the CSC view 1.21 vs 1.15 ms; the forward went from 422 M to 468 M VALU instructions per launch for no change
// in time), spmm_value_bw waits on memory with few registers and gains (1.63 vs 1.77 ms).  DPP moves fuse into the add
// (v_add_f32_dpp) and are used either way.
template <bool SWAP>
__device__ __forceinline__ float add_xor16(float x) {
  if constexpr (SWAP) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    return x + __shfl_xor(x, 16);
  }
}
template <bool SWAP>
__device__ __forceinline__ float add_xor32(float x) {
  if constexpr (SWAP) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    return x + __shfl_xor(x, 32);
  }
}

// x + x(lane ^ OFF) with the bits of x + __shfl_xor(x, OFF) (float addition commutes): the fold of the
// 64 / LPR lane groups of a wave at the end of a row.  OFF = 8 is a DPP row rotation.
template <int OFF, bool SWAP = false>
__device__ __forceinline__ float add_xor(float x) {
  if constexpr (OFF == 8) return x + lane_xor<8>(x);
  else if constexpr (OFF == 16) return add_xor16<SWAP>(x);
  else if constexpr (OFF == 32) return add_xor32<SWAP>(x);
  else return x + __shfl_xor(x, OFF);
}

// acc[i] += acc[i] of the other lane groups, for all groups: offsets LPR, 2 LPR, ... 32
template <int LPR, int N>
__device__ __forceinline__ void fold_lane_groups(float (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float x = acc[i];
    if constexpr (LPR <= 1) x = add_xor<1>(x);
    if constexpr (LPR <= 2) x = add_xor<2>(x);
    if constexpr (LPR <= 4) x = add_xor<4>(x);
    if constexpr (LPR <= 8) x = add_xor<8>(x);
    if constexpr (LPR <= 16) x = add_xor<16>(x);
    if constexpr (LPR <= 32) x = add_xor<32>(x);
    acc[i] = x;
  }
}

// One step of a sum over the lanes of an LPR-lane group (LPR a power of two, groups aligned) that differ in
// bit BIT, the lower bits >= LOW having been summed already and bits < LOW naming DIFFERENT quantities (the
// transposing fold below keeps one partial per lane residue mod LOW): every lane ends up with the sum over its
// own residue class.  BIT = 4 has no exact partner in DPP: a rotation by 4 inside the row of 16 pairs lane l
// with l + 4 — right as long as the step for BIT = 8 follows (LPR >= 16: both rotations stay inside the group
// and together cover the row); in an 8-lane group the mirror of the half row (l <-> 7 - l) serves when the quad
// holds one quantity (LOW == 1).  Anything else falls back to ds_bpermute.
template <int BIT, int LPR, int LOW, bool SWAP = false>
__device__ __forceinline__ float group_sum_step(float x) {
  if constexpr (BIT == 1 || BIT == 2) return x + lane_xor<BIT>(x);
  else if constexpr (BIT == 4 && LPR >= 16) return x + dpp_move<0x124>(x);        // row_ror:4 (completed by BIT = 8)
  else if constexpr (BIT == 4 && LPR == 8 && LOW == 1) return x + dpp_move<0x141>(x);  // row_half_mirror
  else if constexpr (BIT == 8) return x + lane_xor<8>(x);
  else if constexpr (BIT == 16) return add_xor16<SWAP>(x);
  else if constexpr (BIT == 32) return add_xor32<SWAP>(x);
  else return x + __shfl_xor(x, BIT);
}

// Fold the U partial dots every lane of a group holds (dot[u]: this lane's share of edge slot u) into whole
// dots, transposing as it goes: after log2(U) exchange steps lane l keeps ONE partial, of slot u = l % U; the
// remaining bits add up.  Lane l (< U) of the group ends with the whole dot of slot l in dot[0].
// U - 1 + log2(LPR / U) cross-lane moves for U dots instead of U * log2(LPR).
template <int LPR, int U, bool SWAP = false>
__device__ __forceinline__ void fold_group_dots(float (&dot)[U], int l) {
  static_assert((U & (U - 1)) == 0 && U <= LPR, "U must be a power of two <= LPR");
  int cnt = U;
#pragma unroll
  for (int bit = 1; bit < U; bit <<= 1, cnt >>= 1) {
    const bool up = (l & bit) != 0;
#pragma unroll
    for (int i = 0; i < cnt / 2; ++i) {
      const float keep = up ? dot[2 * i + 1] : dot[2 * i];
      const float send = up ? dot[2 * i] : dot[2 * i + 1];
      // the partner l ^ bit exactly: quad permutes for 1 and 2, ds_bpermute above
      dot[i] = keep + (bit == 1 ? lane_xor<1>(send) : bit == 2 ? lane_xor<2>(send) : __shfl_xor(send, bit));
    }
  }
  float x = dot[0];
  if constexpr (U <= 1 && LPR > 1) x = group_sum_step<1, LPR, U>(x);
  if constexpr (U <= 2 && LPR > 2) x = group_sum_step<2, LPR, U>(x);
  if constexpr (U <= 4 && LPR > 4) x = group_sum_step<4, LPR, U>(x);
  if constexpr (U <= 8 && LPR > 8) x = group_sum_step<8, LPR, U>(x);
  if constexpr (U <= 16 && LPR > 16) x = group_sum_step<16, LPR, U, SWAP>(x);
  if constexpr (U <= 32 && LPR > 32) x = group_sum_step<32, LPR, U, SWAP>(x);
  dot[0] = x;
}

}  // namespace psa
