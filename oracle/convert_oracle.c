/*
 * TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.
 *
 * Restates the reference's single-threaded CPU index conversions
 *   ind2ptr_cpu  /root/reference/csrc/cpu/convert_cpu.cpp:6-30
 *   ptr2ind_cpu  /root/reference/csrc/cpu/convert_cpu.cpp:32-48
 * on raw int64 buffers (the reference versions take paddle::Tensor, which
 * cannot be compiled here: no <paddle/extension.h> in this image).
 * Pinned by the reference's known answers test/test_storage.py:20-32
 * (tests/test_oracle.py).
 */
#include <stdint.h>

/* out[M+1]; ind sorted ascending with values in [0, M). */
void oracle_ind2ptr(const int64_t* ind, int64_t numel, int64_t M,
                    int64_t* out) {
  if (numel == 0) { /* convert_cpu.cpp:9-11: zeros(M + 1) */
    for (int64_t r = 0; r <= M; ++r) out[r] = 0;
    return;
  }
  /* rows up to and including the first index start at edge 0 (:19) */
  int64_t r = 0;
  for (; r <= ind[0]; ++r) out[r] = 0;
  /* each step to a larger index closes the rows in between (:21-25) */
  int64_t cur = ind[0];
  for (int64_t e = 1; e < numel; ++e) {
    const int64_t nxt = ind[e];
    while (cur < nxt) {
      out[cur + 1] = e;
      ++cur;
    }
  }
  /* rows after the last index end at numel (:27) */
  for (r = ind[numel - 1] + 1; r <= M; ++r) out[r] = numel;
}

/* ptr[M+1] non-decreasing; out[E]; entries outside [ptr[0], ptr[M]) are not
 * written, as in the reference (:40-45). */
void oracle_ptr2ind(const int64_t* ptr, int64_t M, int64_t E, int64_t* out) {
  (void)E;
  int64_t lo = ptr[0];
  for (int64_t r = 0; r < M; ++r) {
    const int64_t hi = ptr[r + 1];
    for (int64_t e = lo; e < hi; ++e) out[e] = r;
    lo = hi;
  }
}
