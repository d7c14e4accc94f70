/*
 * oracle/nq_oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY.
 *
 * This file is NOT part of the product.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it; the product
 * library (constraint-solver_amd/csrc) never links or falls back to it.
 *
 * It restates, in plain C, the N-queens half of the reference's hot path
 * (asimihsan/constraint-solver; the reference is Rust and cannot be built in
 * this environment: no cargo/rustc, no vendored crates, so there is no
 * oracle/_ref/):
 *
 *   get_col_scores                 examples/nqueens/src/lib.rs:74-87
 *   get_scored_solution            examples/nqueens/src/lib.rs:126-140
 *   MoveIterator (candidate order) examples/nqueens/src/lib.rs:214-234
 *   neighbourhood pipeline         local-search/src/local_search.rs:315-325
 *   tabu == {current}              local-search/src/local_search.rs:155-199
 *   ScoredSolution ordering        local-search/src/local_search.rs:29-37
 *
 * Pinning: the scoring arithmetic is pinned by the reference's own known-answer
 * tests (examples/nqueens/src/lib.rs:93-119: rows=[0,0,0,0] -> [3,3,3,3];
 * rows=[1,3,0,2] -> all 0), checked in tests/test_oracle_nqueens.py.  The
 * reference holds no vector that pins the neighbourhood argmin itself; it is a
 * literal restatement cross-checked by two independent formulations below.
 *
 * Integer type: i64 everywhere, as the reference (`type Integer = i64`,
 * examples/nqueens/src/lib.rs:13).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SKIPPED INT64_MIN /* marker in scores_out for candidates never scored */

/* examples/nqueens/src/lib.rs:74-87 — literal double loop. */
void orc_nq_get_col_scores(const int64_t *rows, int64_t n, int64_t *result)
{
    for (int64_t i = 0; i < n; i++) result[i] = 0;
    for (int64_t col1 = 0; col1 < n; col1++) {
        const int64_t row1 = rows[col1];
        for (int64_t col2 = col1 + 1; col2 < n; col2++) {
            const int64_t row_diff = rows[col2] - row1;
            const int64_t column_diff = col2 - col1;
            const int64_t ard = row_diff < 0 ? -row_diff : row_diff;
            const int64_t acd = column_diff < 0 ? -column_diff : column_diff;
            if (row_diff == 0 || ard == acd) {
                result[col1] += 1;
                result[col2] += 1;
            }
        }
    }
}

/* examples/nqueens/src/lib.rs:130-139 — score = sum of the column scores.
 * Same pair loop as above without the per-column vector (the sum is what the
 * reference keeps). */
int64_t orc_nq_score(const int64_t *rows, int64_t n)
{
    int64_t total = 0;
    for (int64_t col1 = 0; col1 < n; col1++) {
        const int64_t row1 = rows[col1];
        int64_t hits = 0;
        for (int64_t col2 = col1 + 1; col2 < n; col2++) {
            const int64_t row_diff = rows[col2] - row1;
            const int64_t column_diff = col2 - col1;
            const int64_t ard = row_diff < 0 ? -row_diff : row_diff;
            hits += (row_diff == 0) | (ard == column_diff);
        }
        total += 2 * hits;
    }
    return total;
}

static int lexcmp(const int64_t *a, const int64_t *b, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

/*
 * LITERAL formulation of local_search.rs:315-325 for the N-queens plug-in.
 *
 *   iter_local_moves      candidate k = clone(start) with rows[cols[k/n]] = k%n
 *                         (nqueens lib.rs:217-234, cols in proposer order)
 *   .filter(!tabu)        tabu set is {current} (local_search.rs:155-199 with
 *                         every configured expiry >= 1) => drop candidate == current
 *   .map(score)           full O(n^2) rescoring (nqueens lib.rs:130-139)
 *   .take(window)         stop pulling once `window` candidates were scored
 *   sort(); first()       minimum by (score, solution vector), first of equals
 *
 * scores_out (nullable): n_cols*n entries indexed by unfiltered position k;
 * ORC_SKIPPED where the candidate was dropped or never reached.
 * best_index = unfiltered position k of the winner (-1 if none).
 */
int orc_nq_neighbourhood_literal(const int64_t *rows, int64_t n,
                                 const int64_t *cols, int64_t n_cols,
                                 int64_t window,
                                 int64_t *scores_out,
                                 int64_t *n_scored, int64_t *best_index,
                                 int64_t *best_score)
{
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *best = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!cand || !best) { free(cand); free(best); return -1; }
    int64_t scored = 0, bi = -1, bs = 0;
    const int64_t total = n_cols * n;
    if (scores_out)
        for (int64_t k = 0; k < total; k++) scores_out[k] = ORC_SKIPPED;
    for (int64_t k = 0; k < total && scored < window; k++) {
        const int64_t col = cols[k / n];
        const int64_t value = k % n;
        memcpy(cand, rows, sizeof(int64_t) * (size_t)n); /* self.solution.clone() */
        cand[col] = value;
        if (lexcmp(cand, rows, n) == 0) continue;         /* is_solution_tabu */
        const int64_t s = orc_nq_score(cand, n);          /* get_scored_solution */
        if (scores_out) scores_out[k] = s;
        scored++;
        if (bi < 0 || s < bs || (s == bs && lexcmp(cand, best, n) < 0)) {
            bi = k; bs = s;
            memcpy(best, cand, sizeof(int64_t) * (size_t)n);
        }
    }
    *n_scored = scored; *best_index = bi; *best_score = bs;
    free(cand); free(best);
    return 0;
}

/*
 * The same pipeline for a GENERAL tabu set (History with all_solution_iteration_expiry == 0 keeps every visited
 * solution, local-search/src/local_search.rs:155-199): `tabu` holds n_tabu boards of n entries each; a candidate is
 * dropped iff it equals one of them (is_solution_tabu, :197-199: a HashSet lookup of the whole vector).  The current
 * board is NOT added implicitly: the caller passes the History's set, which always contains it after seen_solution.
 */
int orc_nq_neighbourhood_literal_tabu(const int64_t *rows, int64_t n,
                                      const int64_t *cols, int64_t n_cols, int64_t window,
                                      const int64_t *tabu, int64_t n_tabu,
                                      int64_t *scores_out,
                                      int64_t *n_scored, int64_t *best_index, int64_t *best_score)
{
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *best = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!cand || !best) { free(cand); free(best); return -1; }
    int64_t scored = 0, bi = -1, bs = 0;
    const int64_t total = n_cols * n;
    if (scores_out)
        for (int64_t k = 0; k < total; k++) scores_out[k] = ORC_SKIPPED;
    for (int64_t k = 0; k < total && scored < window; k++) {
        memcpy(cand, rows, sizeof(int64_t) * (size_t)n);
        cand[cols[k / n]] = k % n;
        int is_tabu = 0;
        for (int64_t t = 0; t < n_tabu && !is_tabu; t++)
            is_tabu = lexcmp(cand, tabu + t * n, n) == 0;
        if (is_tabu) continue;
        const int64_t s = orc_nq_score(cand, n);
        if (scores_out) scores_out[k] = s;
        scored++;
        if (bi < 0 || s < bs || (s == bs && lexcmp(cand, best, n) < 0)) {
            bi = k; bs = s;
            memcpy(best, cand, sizeof(int64_t) * (size_t)n);
        }
    }
    *n_scored = scored; *best_index = bi; *best_score = bs;
    free(cand); free(best);
    return 0;
}

/* conf(c, r) = number of other columns j attacking square (c, r):
 * rows[j]==r or |rows[j]-r| == |j-c|   (the pair test of lib.rs:78-80). */
static int64_t conf_scan(const int64_t *rows, int64_t n, int64_t c, int64_t r)
{
    int64_t hits = 0;
    for (int64_t j = 0; j < n; j++) {
        if (j == c) continue;
        const int64_t rd = rows[j] - r;
        const int64_t ard = rd < 0 ? -rd : rd;
        const int64_t cd = j < c ? c - j : j - c;
        hits += (rd == 0) | (ard == cd);
    }
    return hits;
}

/* Lexicographic comparison of two single-position edits (c1,v1), (c2,v2) of
 * the same base vector, done by walking the (at most two) positions where
 * either differs from the base — NOT the closed-form key the GPU uses. */
static int lexcmp_edits(const int64_t *rows, int64_t c1, int64_t v1, int64_t c2, int64_t v2)
{
    int64_t p[2]; int np = 0;
    if (c1 == c2) { p[np++] = c1; }
    else if (c1 < c2) { p[np++] = c1; p[np++] = c2; }
    else { p[np++] = c2; p[np++] = c1; }
    for (int i = 0; i < np; i++) {
        const int64_t a = (p[i] == c1) ? v1 : rows[p[i]];
        const int64_t b = (p[i] == c2) ? v2 : rows[p[i]];
        if (a < b) return -1;
        if (a > b) return 1;
    }
    return 0;
}

/*
 * DELTA formulation (second, independent): score(candidate) =
 * score(current) - 2*conf(c, rows[c]) + 2*conf(c, v), conf by an O(n) scan.
 * Identity checked against the literal formulation in tests.  cur_score is
 * computed here from scratch (not trusted from the caller).
 */
int orc_nq_neighbourhood_delta(const int64_t *rows, int64_t n,
                               const int64_t *cols, int64_t n_cols,
                               int64_t window,
                               int64_t *scores_out,
                               int64_t *n_scored, int64_t *best_index,
                               int64_t *best_score)
{
    const int64_t cur = orc_nq_score(rows, n);
    int64_t scored = 0, bi = -1, bs = 0, bc = 0, bv = 0;
    const int64_t total = n_cols * n;
    if (scores_out)
        for (int64_t k = 0; k < total; k++) scores_out[k] = ORC_SKIPPED;
    for (int64_t slot = 0; slot < n_cols && scored < window; slot++) {
        const int64_t c = cols[slot];
        const int64_t old = conf_scan(rows, n, c, rows[c]);
        for (int64_t v = 0; v < n && scored < window; v++) {
            if (v == rows[c]) continue;
            const int64_t s = cur - 2 * old + 2 * conf_scan(rows, n, c, v);
            const int64_t k = slot * n + v;
            if (scores_out) scores_out[k] = s;
            scored++;
            if (bi < 0 || s < bs || (s == bs && lexcmp_edits(rows, c, v, bc, bv) < 0)) {
                bi = k; bs = s; bc = c; bv = v;
            }
        }
    }
    *n_scored = scored; *best_index = bi; *best_score = bs;
    return 0;
}

/*
 * COUNT-TABLE formulation (third, independent): row / diagonal occupancy
 * histograms; conf(c, v) for v != rows[c] is cntR[v] + cntD1[v-c] + cntD2[v+c].
 * O(1) per candidate — used to produce expected results for the full
 * n x n neighbourhood at n = 4096 in seconds.
 */
int orc_nq_neighbourhood_table(const int64_t *rows, int64_t n,
                               const int64_t *cols, int64_t n_cols,
                               int64_t window,
                               int64_t *scores_out,
                               int64_t *n_scored, int64_t *best_index,
                               int64_t *best_score)
{
    int64_t *cr = (int64_t *)calloc((size_t)(5 * n + 5), sizeof(int64_t));
    if (!cr) return -1;
    int64_t *d1 = cr + n;          /* index r - j + n   in [1, 2n-1] */
    int64_t *d2 = d1 + 2 * n + 1;  /* index r + j       in [0, 2n-2] */
    int64_t cur = 0;
    for (int64_t j = 0; j < n; j++) {
        cr[rows[j]]++; d1[rows[j] - j + n]++; d2[rows[j] + j]++;
    }
    for (int64_t j = 0; j < n; j++)
        cur += cr[rows[j]] + d1[rows[j] - j + n] + d2[rows[j] + j] - 3;
    int64_t scored = 0, bi = -1, bs = 0, bc = 0, bv = 0;
    const int64_t total = n_cols * n;
    if (scores_out)
        for (int64_t k = 0; k < total; k++) scores_out[k] = ORC_SKIPPED;
    for (int64_t slot = 0; slot < n_cols && scored < window; slot++) {
        const int64_t c = cols[slot], rc = rows[c];
        const int64_t old = cr[rc] + d1[rc - c + n] + d2[rc + c] - 3;
        for (int64_t v = 0; v < n && scored < window; v++) {
            if (v == rc) continue;
            const int64_t s = cur - 2 * old + 2 * (cr[v] + d1[v - c + n] + d2[v + c]);
            const int64_t k = slot * n + v;
            if (scores_out) scores_out[k] = s;
            scored++;
            if (bi < 0 || s < bs || (s == bs && lexcmp_edits(rows, c, v, bc, bv) < 0)) {
                bi = k; bs = s; bc = c; bv = v;
            }
        }
    }
    *n_scored = scored; *best_index = bi; *best_score = bs;
    free(cr);
    return 0;
}

/* Column-score vector by count tables (cross-check of orc_nq_get_col_scores
 * and the expected output of the GPU col-score kernel). */
void orc_nq_col_scores_table(const int64_t *rows, int64_t n, int64_t *result)
{
    int64_t *cr = (int64_t *)calloc((size_t)(5 * n + 5), sizeof(int64_t));
    int64_t *d1 = cr + n, *d2 = d1 + 2 * n + 1;
    for (int64_t j = 0; j < n; j++) {
        cr[rows[j]]++; d1[rows[j] - j + n]++; d2[rows[j] + j]++;
    }
    for (int64_t j = 0; j < n; j++)
        result[j] = cr[rows[j]] + d1[rows[j] - j + n] + d2[rows[j] + j] - 3;
    free(cr);
}

/*
 * CPU baseline leg for bench.py: the reference-shaped per-candidate work
 * (clone + tabu compare + full O(n^2) rescoring + running (score, vector)
 * minimum) over the first `max_candidates` non-tabu candidates.  Returns the
 * number scored; the caller times it.
 */
int64_t orc_nq_baseline_run(const int64_t *rows, int64_t n,
                            const int64_t *cols, int64_t n_cols,
                            int64_t max_candidates, int64_t *best_score_out)
{
    int64_t ns = 0, bi = -1, bs = 0;
    orc_nq_neighbourhood_literal(rows, n, cols, n_cols, max_candidates, NULL, &ns, &bi, &bs);
    if (best_score_out) *best_score_out = bs;
    return ns;
}
