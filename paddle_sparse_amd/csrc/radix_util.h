// Pieces of the LSD radix sort shared by sort.hip (multi-workgroup passes) and
// chain.hip (the one-workgroup, LDS-resident sort of the small coalesce).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psa {

constexpr int kRadix = 256;

__device__ __forceinline__ unsigned digit_of(uint64_t key, int shift) {
  return static_cast<unsigned>(key >> shift) & (kRadix - 1);
}

// Stable rank of this lane's key among the wave's keys with the same digit,
// continuing the wave's running count in wcnt (LDS, 256 counters).
//
// Split in two so that the 8 ballot chains of all items (pure VALU/SALU, no
// memory) can interleave, and only the short counter update is serial:
//   wave_match : (#peers below this lane) | (#peers << 8), 0xffff.. if invalid
//   wave_rank  : rank from the packed match word + the LDS running count
__device__ __forceinline__ uint32_t wave_match(unsigned d, bool valid) {
  // lanes whose bit b differs from mine = ballot(bit b) ^ (my bit ? ~0 : 0);
  // OR the 8 difference masks, the complement are my peers.  Written on the
  // 32-bit halves so it lowers to v_bfe_i32 + v_cmp + 2 v_xor + 2 v_or per bit.
  const unsigned long long vm = __ballot(valid);
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t neg = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(d), b, 1));
    const unsigned long long m = __ballot(neg != 0u);
    lo |= static_cast<uint32_t>(m) ^ neg;
    hi |= static_cast<uint32_t>(m >> 32) ^ neg;
  }
  const uint32_t plo = ~lo & static_cast<uint32_t>(vm);
  const uint32_t phi = ~hi & static_cast<uint32_t>(vm >> 32);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
  return below | (static_cast<uint32_t>(__popc(plo) + __popc(phi)) << 8);
}

__device__ __forceinline__ uint32_t wave_rank(uint32_t* wcnt, unsigned d, bool valid,
                                              uint32_t match) {
  const uint32_t below = match & 0xffu;
  const uint32_t old = wcnt[d];
  if (valid && below == 0u)  // lowest lane of the peer set publishes the new count
    wcnt[d] = old + (match >> 8);
  return old + below;
}

}  // namespace psa
