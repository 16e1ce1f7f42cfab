/* TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.
 *
 * C restatement of the reference's sort + coalesce path, used as the timed
 * CPU baseline of the sort / coalesce / transpose rows
 * and cross-checked against the numpy restatement (storage_oracle.py) in
 * tests/test_oracle.py:
 *
 *   key = row * N + col, sortedness test, permutation applied to row/col/value
 *       /root/reference/paddle_sparse/storage.py:158-171
 *   index_sort (argsort of the key)   /root/reference/paddle_sparse/utils.py:14-23
 *   head flags against a -1 sentinel, ptr = positions of heads, segment_csr
 *       /root/reference/paddle_sparse/storage.py:449-486
 *   coalesce(index, value, m, n, op)  /root/reference/paddle_sparse/coalesce.py:25-29
 *
 * The reference sorts with paddle.argsort (a comparison sort, order of equal
 * keys unspecified); this restatement uses a STABLE LSD radix sort, the same
 * convention as storage_oracle.index_sort and the HIP path.  Values of one
 * segment are accumulated in segment order in fp32 (separate roundings:
 * built with -ffp-contract=off).
 *
 * threads <= 1 runs every loop sequentially (the reference's CPU ops have no
 * OpenMP pragmas); threads > 1 is the all-cores baseline.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { R_SUM = 0, R_MEAN = 1, R_MIN = 2, R_MAX = 3 };

static int radix_passes(int64_t max_value) {
    int bits = 0;
    uint64_t top = max_value > 1 ? (uint64_t)(max_value - 1) : 1;
    while (top) { ++bits; top >>= 1; }
    return (bits + 7) / 8;
}

/* Stable LSD radix sort of (key, idx) pairs on the low 8*passes bits.
 * Returns 0 when the result is in (k0, i0), 1 when it is in (k1, i1). */
static int radix_sort_pairs(int64_t* k0, int64_t* i0, int64_t* k1, int64_t* i1,
                            int64_t n, int passes, int threads) {
    if (threads < 1) threads = 1;
    int64_t* hist = (int64_t*)malloc(sizeof(int64_t) * 256 * (size_t)threads);
    int64_t *sk = k0, *si = i0, *dk = k1, *di = i1;
    int where = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        memset(hist, 0, sizeof(int64_t) * 256 * (size_t)threads);
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            const int64_t lo = n * t / T, hi = n * (t + 1) / T;
            int64_t* h = hist + 256 * t;
            for (int64_t e = lo; e < hi; ++e) ++h[((uint64_t)sk[e] >> shift) & 255];
#pragma omp barrier
#pragma omp single
            {
                /* digit-major, thread-minor exclusive scan keeps the sort stable */
                int64_t run = 0;
                for (int d = 0; d < 256; ++d)
                    for (int u = 0; u < T; ++u) {
                        const int64_t c = hist[256 * u + d];
                        hist[256 * u + d] = run;
                        run += c;
                    }
            }
            for (int64_t e = lo; e < hi; ++e) {
                const int64_t pos = h[((uint64_t)sk[e] >> shift) & 255]++;
                dk[pos] = sk[e];
                di[pos] = si[e];
            }
        }
        int64_t* tk = sk; sk = dk; dk = tk;
        int64_t* ti = si; si = di; di = ti;
        where ^= 1;
    }
    free(hist);
    return where;
}

/* utils.py:14-23: sorted keys and the (stable) permutation. */
void oracle_index_sort(const int64_t* keys, int64_t n, int64_t max_value,
                       int64_t* sorted_out, int64_t* perm_out, int threads) {
    if (n <= 0) return;
    int64_t* k1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* i1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(sorted_out, keys, sizeof(int64_t) * (size_t)n);
#pragma omp parallel for num_threads(threads > 1 ? threads : 1) schedule(static)
    for (int64_t e = 0; e < n; ++e) perm_out[e] = e;
    if (radix_sort_pairs(sorted_out, perm_out, k1, i1, n, radix_passes(max_value), threads)) {
        memcpy(sorted_out, k1, sizeof(int64_t) * (size_t)n);
        memcpy(perm_out, i1, sizeof(int64_t) * (size_t)n);
    }
    free(k1);
    free(i1);
}

/* coalesce.py:25-29 on fp32 values of D columns (D = 0: no values).
 * out_row/out_col/out_value must hold nnz entries; returns nnz'. */
int64_t oracle_coalesce_f32(const int64_t* row, const int64_t* col, const float* value,
                            int64_t D, int64_t nnz, int64_t M, int64_t N, int reduce,
                            int64_t* out_row, int64_t* out_col, float* out_value, int threads) {
    if (nnz <= 0) return 0;
    if (threads < 1) threads = 1;
    int64_t* key = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnz);
    int64_t* perm = NULL;
    int unsorted = 0;
    /* storage.py:158-163 */
#pragma omp parallel for num_threads(threads) schedule(static) reduction(| : unsorted)
    for (int64_t e = 0; e < nnz; ++e) {
        key[e] = row[e] * N + col[e];
        if (e > 0 && row[e] * N + col[e] < row[e - 1] * N + col[e - 1]) unsorted |= 1;
    }
    if (unsorted) { /* storage.py:164-171 */
        int64_t* k1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnz);
        int64_t* i0 = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnz);
        int64_t* i1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)nnz);
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t e = 0; e < nnz; ++e) i0[e] = e;
        if (radix_sort_pairs(key, i0, k1, i1, nnz, radix_passes(M * N), threads)) {
            free(key); free(i0);
            key = k1; perm = i1;
        } else {
            free(k1); free(i1);
            perm = i0;
        }
    }
    /* storage.py:458-466: heads = key > previous key (sentinel -1), ptr = their positions */
    int64_t* ptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    int64_t count = 0;
    int64_t prev = -1;
    for (int64_t e = 0; e < nnz; ++e) {
        if (key[e] > prev) ptr[count++] = e;
        prev = key[e];
    }
    ptr[count] = nnz;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t s = 0; s < count; ++s) {
        const int64_t k = key[ptr[s]];
        out_row[s] = k / N;
        out_col[s] = k % N;
    }
    /* storage.py:468-471: segment_csr(value, ptr, reduce) along dim 0 */
    if (value != NULL && D > 0) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4096)
        for (int64_t s = 0; s < count; ++s) {
            const int64_t lo = ptr[s], hi = ptr[s + 1];
            float* o = out_value + s * D;
            const float* v0 = value + (perm ? perm[lo] : lo) * D;
            for (int64_t d = 0; d < D; ++d) o[d] = v0[d];
            for (int64_t e = lo + 1; e < hi; ++e) {
                const float* v = value + (perm ? perm[e] : e) * D;
                for (int64_t d = 0; d < D; ++d) {
                    if (reduce == R_MIN) o[d] = v[d] < o[d] ? v[d] : o[d];
                    else if (reduce == R_MAX) o[d] = v[d] > o[d] ? v[d] : o[d];
                    else o[d] = o[d] + v[d];
                }
            }
            if (reduce == R_MEAN)
                for (int64_t d = 0; d < D; ++d) o[d] = o[d] / (float)(hi - lo);
        }
    }
    free(ptr);
    free(key);
    free(perm);
    return count;
}
