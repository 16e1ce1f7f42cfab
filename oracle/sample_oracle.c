/* TEST INFRASTRUCTURE — CPU oracle, never part of the shipped path.
 *
 * sample_adj, following /root/reference/csrc/cpu/sample_cpu.cpp:9-148 step by
 * step: a map from node id to new id seeded with the subset (:31-39, a later
 * duplicate overwrites an earlier one), rows walked in subset order, unseen
 * columns numbered in first-seen order (:54-58, :77-81, :111-115), the three
 * selection branches (:45-63 all neighbours, :66-88 with replacement, :90-121
 * without replacement by the Floyd-style loop of :100-106), rows sorted by new
 * column id (:131-141), outputs rowptr / col / n_id / e_id (:124-147).
 *
 * Two things the reference leaves open are fixed here the way the HIP path
 * fixes them (include/paddle_sparse_hip.h, "sample_adj on the GPU"):
 *   - random draws: uniform_randint (cpu/utils.h:22-34, the framework's global
 *     generator) becomes randint(seed, i, t, n), a pure function of the subset
 *     row i and the draw number t;
 *   - order: picks of one row are taken in draw order (the reference iterates
 *     an unordered_set at :108) and the per-row sort is stable.
 * With num_neighbors < 0 there is no draw and no set: that branch is the
 * reference's result exactly (pinned by test/test_sample.py:17-30).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int64_t randint(uint64_t seed, int64_t i, int64_t t, int64_t n) {
    const uint64_t r = mix64(mix64(seed ^ mix64((uint64_t)i)) + (uint64_t)t);
    return (int64_t)(((unsigned __int128)r * (uint64_t)n) >> 64);
}

typedef struct { int64_t col, e, seq; } pick_t;

/* by new column id, draw order among equals (= a stable sort) */
static int cmp_pick(const void* a, const void* b) {
    const pick_t *x = (const pick_t*)a, *y = (const pick_t*)b;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return (x->seq > y->seq) - (x->seq < y->seq);
}

/* out_rowptr: int64[S+1].  out_col / out_e_id: capacity `cap` picks; n_id:
 * capacity S + cap.  num_nodes bounds every node id (rows and columns).
 * Returns the number of nodes in n_id, or -1 if cap is too small. */
int64_t oracle_sample_adj(const int64_t* rowptr, const int64_t* col, const int64_t* idx,
                          int64_t S, int64_t num_nodes, int64_t num_neighbors, int replace,
                          uint64_t seed, int64_t cap, int64_t* out_rowptr, int64_t* out_col,
                          int64_t* n_id, int64_t* out_e_id) {
    int64_t* map = (int64_t*)malloc(sizeof(int64_t) * (size_t)(num_nodes > 0 ? num_nodes : 1));
    for (int64_t v = 0; v < num_nodes; ++v) map[v] = -1;
    pick_t* picks = (pick_t*)malloc(sizeof(pick_t) * (size_t)(cap > 0 ? cap : 1));
    int64_t n_nodes = 0, E = 0;
    for (int64_t i = 0; i < S; ++i) { /* :35-39 */
        map[idx[i]] = i;
        n_id[n_nodes++] = idx[i];
    }
    out_rowptr[0] = 0;
    for (int64_t i = 0; i < S; ++i) {
        const int64_t n = idx[i];
        const int64_t start = rowptr[n], deg = rowptr[n + 1] - start;
        int64_t cnt;
        if (num_neighbors < 0) cnt = deg;
        else if (replace) cnt = deg > 0 ? num_neighbors : 0;
        else cnt = deg < num_neighbors ? deg : num_neighbors;
        if (E + cnt > cap) { free(map); free(picks); return -1; }
        pick_t* mine = picks + E;
        for (int64_t t = 0; t < cnt; ++t) {
            int64_t e;
            if (num_neighbors < 0 || (!replace && deg <= num_neighbors)) {
                e = start + t;                                   /* :50-51, :97-98 */
            } else if (replace) {
                e = start + randint(seed, i, t, deg);             /* :74 */
            } else {                                             /* :100-106 */
                const int64_t j = deg - num_neighbors + t;
                e = start + randint(seed, i, t, j);
                for (int64_t u = 0; u < t; ++u)
                    if (mine[u].e == e) { e = start + j; break; }
            }
            mine[t].e = e;
        }
        for (int64_t t = 0; t < cnt; ++t) {                       /* :52-59 */
            const int64_t c = col[mine[t].e];
            if (map[c] < 0) {
                map[c] = n_nodes;
                n_id[n_nodes++] = c;
            }
            mine[t].col = map[c];
            mine[t].seq = t;
        }
        qsort(mine, (size_t)cnt, sizeof(pick_t), cmp_pick);                                    /* :131-141 */
        E += cnt;
        out_rowptr[i + 1] = E;
    }
    for (int64_t p = 0; p < E; ++p) {
        out_col[p] = picks[p].col;
        out_e_id[p] = picks[p].e;
    }
    free(map);
    free(picks);
    return n_nodes;
}
