/* TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.
 *
 * Sparse x sparse product C = A @ B for the `spspmm` the reference documents
 * (/root/reference/README.md:308-353) without shipping a kernel: upstream
 * pytorch_sparse's CPU path hands the product to a row-by-row (Gustavson)
 * CSR product.  Restated here as the textbook algorithm: for every row i of A,
 * walk its entries in storage order, add a * B[c, :] into a dense accumulator
 * in B's storage order, then emit the touched columns in increasing order.
 * fp32 sums are taken in exactly that order (separate roundings).
 *
 * Pinned by the README known answer (tests/golden/reference_kats.json
 * "spspmm") and cross-checked against scipy.sparse in tests/test_oracle.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_i64(const void* a, const void* b) {
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

/* Pass out_* = NULL to count only.  Returns nnz(C).  A, B in CSR (valA / valB
 * NULL = ones). */
int64_t oracle_spspmm_f32(const int64_t* rowptrA, const int64_t* colA, const float* valA,
                          const int64_t* rowptrB, const int64_t* colB, const float* valB,
                          int64_t m, int64_t n, int64_t* out_row, int64_t* out_col,
                          float* out_val) {
    float* acc = (float*)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    unsigned char* seen = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t* touched = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t nnz = 0;
    for (int64_t i = 0; i < m; ++i) {
        int64_t cnt = 0;
        for (int64_t e = rowptrA[i]; e < rowptrA[i + 1]; ++e) {
            const int64_t c = colA[e];
            const float a = valA ? valA[e] : 1.0f;
            for (int64_t q = rowptrB[c]; q < rowptrB[c + 1]; ++q) {
                const int64_t j = colB[q];
                const float prod = a * (valB ? valB[q] : 1.0f);
                if (!seen[j]) {
                    seen[j] = 1;
                    touched[cnt++] = j;
                    acc[j] = prod;
                } else {
                    acc[j] = acc[j] + prod;
                }
            }
        }
        qsort(touched, (size_t)cnt, sizeof(int64_t), cmp_i64);
        for (int64_t t = 0; t < cnt; ++t) {
            const int64_t j = touched[t];
            if (out_row) {
                out_row[nnz] = i;
                out_col[nnz] = j;
                if (out_val) out_val[nnz] = acc[j];
            }
            seen[j] = 0;
            ++nnz;
        }
    }
    free(acc);
    free(seen);
    free(touched);
    return nnz;
}
