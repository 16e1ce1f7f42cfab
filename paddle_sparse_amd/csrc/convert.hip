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

// One wave per 64 consecutive rows.  The wave's edges [ptr[r0], ptr[r0+64))
// are contiguous, so it sweeps them 64 at a time (coalesced stores) and each
// lane finds its row by a 6-step search over the 64 pointers kept in LDS.
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
  for (int64_t e = e_begin + lane; e < e_end; e += 64) {
    int pos = 0;  // largest i in [0, 63] with sptr[i] <= e
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      if (sptr[wave][pos + step] <= e) pos += step;
    }
    out[e] = r0 + pos;
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
  hipLaunchKernelGGL(ptr2ind_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kThreads), 0, s, ptr, out, M, E);
  PSA_LAUNCH_CHECK();
  return PSA_OK;
}

}  // extern "C"
