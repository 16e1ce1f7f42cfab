// ind2ptr / ptr2ind for gfx950.
//
// Behaviour follows the reference CPU text csrc/cpu/convert_cpu.cpp:6-48
// (bit-exact int64 results); the launch shapes are new.  The reference's
// CUDA kernels (csrc/cuda/convert_cuda.cu) give one thread a data-dependent
// serial loop (gap of empty rows / whole row of edges), which leaves 63 of 64
// lanes idle on skewed inputs and scatters the stores.  Here every long run
// is written by the whole wavefront with contiguous 512-B stores.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// One lane per boundary t in [0, numel]: rows r in (ind[t-1], ind[t]] get
// out[r] = t.  Runs longer than kShortRun are handed to the whole wave.
constexpr int kShortRun = 4;

__global__ void __launch_bounds__(kThreads)
ind2ptr_kernel(const int64_t* __restrict__ ind, int64_t* __restrict__ out,
               int64_t M, int64_t numel) {
  const int lane = threadIdx.x & 63;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  int64_t lo = 0, hi = 0;
  if (t <= numel) {
    lo = (t == 0) ? 0 : ind[t - 1] + 1;
    hi = (t == numel) ? M + 1 : ind[t] + 1;
    // The reference does not validate index values (UB when out of range);
    // clamp so a bad input cannot turn into an out-of-bounds store.
    lo = lo < 0 ? 0 : lo;
    hi = hi > M + 1 ? M + 1 : hi;
  }
  const int64_t len = hi - lo;
  if (len > 0 && len <= kShortRun) {
    for (int64_t r = lo; r < hi; ++r) out[r] = t;
  }
  unsigned long long pending = __ballot(len > kShortRun);
  while (pending) {
    const int src = __ffsll(static_cast<long long>(pending)) - 1;
    pending &= pending - 1;
    const int64_t l = __shfl(lo, src);
    const int64_t h = __shfl(hi, src);
    const int64_t tv = t - lane + src;
    for (int64_t r = l + lane; r < h; r += 64) out[r] = tv;
  }
}

// The same with two boundaries per lane, so that `ind` is read with one 16-byte
// load per lane (8-byte accesses reach only ~0.6 of the 16-byte rate on this
// chip); the entry before the pair comes from the previous lane.  Needs a
// 16-byte aligned `ind`.
__global__ void __launch_bounds__(kThreads)
ind2ptr_pair_kernel(const int64_t* __restrict__ ind, int64_t* __restrict__ out,
                    int64_t M, int64_t numel) {
  const int lane = threadIdx.x & 63;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const int64_t t0 = 2 * i;  // this lane's boundaries: t0 and t0 + 1
  int64_t a = 0, b = 0;      // ind[t0], ind[t0 + 1]
  if (t0 + 1 < numel) {
    const longlong2 p = *reinterpret_cast<const longlong2*>(ind + t0);
    a = p.x;
    b = p.y;
  } else if (t0 < numel) {
    a = ind[t0];
  }
  int64_t prev = __shfl_up(static_cast<long long>(b), 1);  // ind[t0 - 1]
  if (lane == 0) prev = (t0 > 0 && t0 <= numel) ? ind[t0 - 1] : 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t t = t0 + h;
    int64_t lo = 0, hi = 0;
    if (t <= numel) {
      lo = (t == 0) ? 0 : (h == 0 ? prev : a) + 1;
      hi = (t == numel) ? M + 1 : (h == 0 ? a : b) + 1;
      lo = lo < 0 ? 0 : lo;  // as in ind2ptr_kernel: no out-of-bounds store on bad input
      hi = hi > M + 1 ? M + 1 : hi;
    }
    const int64_t len = hi - lo;
    if (len > 0 && len <= kShortRun) {
      for (int64_t r = lo; r < hi; ++r) out[r] = t;
    }
    unsigned long long pending = __ballot(len > kShortRun);
    while (pending) {
      const int src = __ffsll(static_cast<long long>(pending)) - 1;
      pending &= pending - 1;
      const int64_t l = __shfl(lo, src);
      const int64_t hh = __shfl(hi, src);
      const int64_t tv = t + 2 * (src - lane);
      for (int64_t r = l + lane; r < hh; r += 64) out[r] = tv;
    }
  }
}

// One wave per 64 consecutive rows.  The wave's edges [ptr[r0], ptr[r0+64))
// are contiguous, so it sweeps them 64 at a time (coalesced stores) and each
// lane finds its row by a 6-step search over the 64 pointers kept in LDS.
template <bool PAIRS>
__global__ void __launch_bounds__(kThreads)
ptr2ind_kernel(const int64_t* __restrict__ ptr, int64_t* __restrict__ out,
               int64_t M, int64_t E) {
  __shared__ int64_t sptr[kThreads / 64][65];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t r0 =
      (static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + wave) * 64;
  const bool live = r0 < M;
  int64_t nrows = 0;
  if (live) {
    nrows = M - r0 < 64 ? M - r0 : 64;
    const int64_t i = lane < nrows ? lane : nrows;
    sptr[wave][lane] = ptr[r0 + i];
    if (lane == 0) sptr[wave][64] = ptr[r0 + nrows];
  }
  __syncthreads();
  if (!live) return;
  int64_t e_begin = sptr[wave][0];
  int64_t e_end = sptr[wave][64];
  e_begin = e_begin < 0 ? 0 : e_begin;
  e_end = e_end > E ? E : e_end;
  auto row_of = [&](int64_t e) {
    int pos = 0;  // largest i in [0, 63] with sptr[i] <= e
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      if (sptr[wave][pos + step] <= e) pos += step;
    }
    return r0 + pos;
  };
  // A block that holds far more entries than rows (hub rows of a power-law graph: block 0
  // of R-MAT scale 21 has ~200 k entries, one row alone 41 677) is filled ROW BY ROW: every
  // row's range is a plain coalesced fill, no search at all.  With the per-lane search below
  // such a wave ran the 6-step LDS search 1 600 times in a row and ptr2ind took 1.0 ms on
  // that graph against 0.03 ms on a uniform one.
  if (e_end - e_begin > 64 * 64) {
    for (int r = 0; r < nrows; ++r) {  // wave-uniform bounds
      int64_t rs = sptr[wave][r], re = sptr[wave][r + 1];
      rs = rs < e_begin ? e_begin : rs;
      re = re > e_end ? e_end : re;
      const int64_t id = r0 + r;
      if (PAIRS) {
        int64_t e = rs;
        if ((e & 1) && e < re) {  // odd head
          if (lane == 0) out[e] = id;
          ++e;
        }
        longlong2 v;
        v.x = v.y = id;
        for (int64_t q = e + 2 * lane; q + 1 < re; q += 128) *reinterpret_cast<longlong2*>(out + q) = v;
        if (((re - e) & 1) && re > e && lane == 0) out[re - 1] = id;  // odd tail
      } else {
        for (int64_t q = rs + lane; q < re; q += 64) out[q] = id;
      }
    }
    return;
  }
  if (!PAIRS) {
    for (int64_t e = e_begin + lane; e < e_end; e += 64) out[e] = row_of(e);
    return;
  }
  // two edges per lane, one 16-byte store (out 16-byte aligned): pairs start at
  // even e; the odd ends of the wave's range fall back to 8-byte stores
  for (int64_t e = (e_begin & ~int64_t{1}) + 2 * lane; e < e_end; e += 128) {
    const bool first = e >= e_begin, second = e + 1 < e_end;
    longlong2 v;
    v.x = first ? row_of(e) : 0;
    v.y = second ? row_of(e + 1) : 0;
    if (first && second) *reinterpret_cast<longlong2*>(out + e) = v;
    else if (first) out[e] = v.x;
    else if (second) out[e + 1] = v.y;
  }
}

}  // namespace

extern "C" {

int psa_ind2ptr(const int64_t* ind, int64_t numel, int64_t M, int64_t* out,
                psa_stream_t stream) {
  PSA_REQUIRE(numel >= 0 && M >= 0, "negative size");
  PSA_REQUIRE(out != nullptr, "out is NULL");
  PSA_REQUIRE(numel == 0 || ind != nullptr, "ind is NULL");
  hipStream_t s = psa::as_stream(stream);
  if (numel == 0) {  // csrc/cpu/convert_cpu.cpp:9-11
    PSA_ZERO(out, sizeof(int64_t) * (M + 1), s);
    return PSA_OK;
  }
  if (psa::aligned(ind, 16)) {
    const int64_t blocks = psa::ceil_div(numel / 2 + 1, kThreads);
    PSA_REQUIRE(blocks <= 0x7fffffff, "numel too large for one launch");
    hipLaunchKernelGGL(ind2ptr_pair_kernel, dim3(static_cast<unsigned>(blocks)),
                       dim3(kThreads), 0, s, ind, out, M, numel);
    PSA_LAUNCH_CHECK();
    return PSA_OK;
  }
  const int64_t blocks = psa::ceil_div(numel + 1, kThreads);
  PSA_REQUIRE(blocks <= 0x7fffffff, "numel too large for one launch");
  hipLaunchKernelGGL(ind2ptr_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, ind, out, M, numel);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

int psa_ptr2ind(const int64_t* ptr, int64_t M, int64_t E, int64_t* out,
                psa_stream_t stream) {
  PSA_REQUIRE(M >= 0 && E >= 0, "negative size");
  PSA_REQUIRE(ptr != nullptr, "ptr is NULL");
  PSA_REQUIRE(E == 0 || out != nullptr, "out is NULL");
  if (M == 0 || E == 0) return PSA_OK;
  hipStream_t s = psa::as_stream(stream);
  const int64_t blocks = psa::ceil_div(M, 64 * (kThreads / 64));
  PSA_REQUIRE(blocks <= 0x7fffffff, "M too large for one launch");
  if (psa::aligned(out, 16)) {
    hipLaunchKernelGGL(ptr2ind_kernel<true>, dim3(static_cast<unsigned>(blocks)),
                       dim3(kThreads), 0, s, ptr, out, M, E);
  } else {
    hipLaunchKernelGGL(ptr2ind_kernel<false>, dim3(static_cast<unsigned>(blocks)),
                       dim3(kThreads), 0, s, ptr, out, M, E);
  }
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
