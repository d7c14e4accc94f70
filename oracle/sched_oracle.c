/*
 * oracle/sched_oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (see
 * nq_oracle.c for the rules: never linked, loaded or called by the product).
 *
 * Plain-C restatement of the employee-scheduling half of the reference's hot
 * path (reference is Rust, unbuildable here):
 *
 *   get_date_index / get_employee_for_date   examples/employee-scheduling/src/lib.rs:149-168
 *   get_employees_to_days                    lib.rs:170-179
 *   get_days_to_employees                    lib.rs:181-192
 *   get_weekday_to_employee_counts_score     lib.rs:194-218
 *   is_weekend                               lib.rs:220-222
 *   ScheduleSolutionScoreCalculator::get_scored_solution   lib.rs:261-375
 *   ScheduleRandomMoveProposer moves (ChangeDay/SwapDays)  lib.rs:464-479
 *   ScheduleMoveProposer (exhaustive ring)   lib.rs:526-550
 *   neighbourhood pipeline                   local-search/src/local_search.rs:315-325
 *
 * PARITY UNPINNED by the reference: examples/employee-scheduling has no
 * #[test] and no fixture, so no reference-held vector pins these scores.  The
 * restatement is cross-checked against (a) an independent Python/Counter
 * formulation in tests/test_oracle_sched.py and (b) the hand-derivable cases
 * of SURVEY.md 8(c).
 *
 * Scores are small non-negative integer counts; the reference accumulates them
 * in f64 (lib.rs:269-270), exactly representable, so int64 here and the
 * caller converts to double for the ScheduleScore{hard,soft} view.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SKIPPED INT64_MIN

/* Days since 1970-01-01 of a proleptic Gregorian date (chrono NaiveDate). */
static int64_t days_from_civil(int64_t y, int64_t m, int64_t d)
{
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const int64_t yoe = y - era * 400;
    const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + doe - 719468;
}

/* chrono Weekday numbering from Monday: Mon=0 .. Sun=6 (1970-01-01 = Thu). */
int64_t orc_weekday_from_monday(int64_t y, int64_t m, int64_t d)
{
    int64_t z = days_from_civil(y, m, d);
    int64_t w = (z + 3) % 7;
    return w < 0 ? w + 7 : w;
}

int64_t orc_days_between(int64_t y0, int64_t m0, int64_t d0, int64_t y1, int64_t m1, int64_t d1)
{
    return days_from_civil(y1, m1, d1) - days_from_civil(y0, m0, d0);
}

typedef struct {
    int64_t n_scored_days;   /* len(get_days_to_employees) = end - start + 1, lib.rs:181-192 */
    int64_t start_weekday;   /* Monday = 0 */
    /* employee_to_holidays flattened: (employee id, day index) pairs, lib.rs:251-253 */
    const int64_t *hol_emp;
    const int64_t *hol_day;
    int64_t n_holidays;
} orc_sched_instance;

static int is_weekend_idx(const orc_sched_instance *in, int64_t idx)
{
    const int64_t wd = (in->start_weekday + idx) % 7;
    return wd == 5 || wd == 6; /* Sat || Sun, lib.rs:220-222 */
}

/* number of distinct values in w[0..len) whose multiplicity is > thresh
 * (itertools .counts().filter(count > thresh).count(), lib.rs:319-325). */
static int64_t count_over(const int64_t *w, int64_t len, int64_t thresh)
{
    int64_t viol = 0;
    for (int64_t p = 0; p < len; p++) {
        int first = 1;
        for (int64_t q = 0; q < p; q++) if (w[q] == w[p]) { first = 0; break; }
        if (!first) continue;
        int64_t cnt = 0;
        for (int64_t q = 0; q < len; q++) cnt += (w[q] == w[p]);
        viol += cnt > thresh;
    }
    return viol;
}

/*
 * get_scored_solution, lib.rs:265-374.  e = date_to_employee (stored length may
 * exceed n_scored_days by one, lib.rs:407-411; only the first n_scored_days
 * entries are scored).  terms_out (nullable): H1..H4, S1..S4.
 */
int orc_sched_score(const orc_sched_instance *in, const int64_t *e,
                    int64_t *hard_out, int64_t *soft_out, int64_t *terms_out)
{
    const int64_t D = in->n_scored_days;
    int64_t h1 = 0, h2 = 0, h3 = 0, h4 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;

    /* H1 holidays, lib.rs:273-280 */
    for (int64_t i = 0; i < in->n_holidays; i++) {
        const int64_t idx = in->hol_day[i];
        if (idx < 0 || idx >= D) return -2; /* reference: unwrap() panic */
        if (e[idx] == in->hol_emp[i]) h1 += 1;
    }
    /* H2 consecutive days, windows(2), lib.rs:286-292 */
    for (int64_t w = 0; w + 2 <= D; w++)
        if (e[w] == e[w + 1]) h2 += 1;
    /* H3 consecutive weekends, windows(9), lib.rs:295-315 */
    for (int64_t w = 0; w + 9 <= D; w++) {
        if (!(is_weekend_idx(in, w) && is_weekend_idx(in, w + 1))) continue;
        if (e[w] == e[w + 7]) h3 += 1;
        if (e[w] == e[w + 8]) h3 += 1;
        if (e[w + 1] == e[w + 7]) h3 += 1;
        if (e[w + 1] == e[w + 8]) h3 += 1;
    }
    /* H4 no more than 3 per 14 days, windows(14), lib.rs:318-327 */
    for (int64_t w = 0; w + 14 <= D; w++) h4 += count_over(e + w, 14, 3);
    /* S1 no more than 2 per 7 days, windows(7), lib.rs:330-339 */
    for (int64_t w = 0; w + 7 <= D; w++) s1 += count_over(e + w, 7, 2);
    /* S2 weekday consistency, lib.rs:194-218 */
    for (int64_t wd = 0; wd < 5; wd++) {
        /* per-employee counts among the scored days falling on weekday wd */
        int64_t distinct = 0, mn = 0;
        for (int64_t p = 0; p < D; p++) {
            if ((in->start_weekday + p) % 7 != wd) continue;
            int first = 1;
            for (int64_t q = 0; q < p; q++)
                if ((in->start_weekday + q) % 7 == wd && e[q] == e[p]) { first = 0; break; }
            if (!first) continue;
            int64_t cnt = 0;
            for (int64_t q = 0; q < D; q++)
                if ((in->start_weekday + q) % 7 == wd && e[q] == e[p]) cnt++;
            if (distinct == 0 || cnt < mn) mn = cnt;
            distinct++;
        }
        if (distinct <= 1) continue;       /* employee_count.len() <= 1, lib.rs:206 */
        s2 += mn;                          /* MinMax(min, _), lib.rs:212-214 */
    }
    /* S3 / S4 over employees present in the scored days, lib.rs:345-365 */
    {
        int64_t distinct = 0, mn = 0, mx = 0, wmn = 0, wmx = 0;
        for (int64_t p = 0; p < D; p++) {
            int first = 1;
            for (int64_t q = 0; q < p; q++) if (e[q] == e[p]) { first = 0; break; }
            if (!first) continue;
            int64_t cnt = 0, wcnt = 0;
            for (int64_t q = 0; q < D; q++)
                if (e[q] == e[p]) { cnt++; wcnt += is_weekend_idx(in, q); }
            if (distinct == 0) { mn = mx = cnt; wmn = wmx = wcnt; }
            else {
                if (cnt < mn) mn = cnt;
                if (cnt > mx) mx = cnt;
                if (wcnt < wmn) wmn = wcnt;
                if (wcnt > wmx) wmx = wcnt;
            }
            distinct++;
        }
        if (distinct >= 2) { s3 = mx - mn; s4 = wmx - wmn; } /* MinMaxResult::MinMax only */
    }
    *hard_out = h1 + h2 + h3 + h4;
    *soft_out = s1 + s2 + s3 + s4;
    if (terms_out) {
        terms_out[0] = h1; terms_out[1] = h2; terms_out[2] = h3; terms_out[3] = h4;
        terms_out[4] = s1; terms_out[5] = s2; terms_out[6] = s3; terms_out[7] = s4;
    }
    return 0;
}

/* A move in proposer-stream order.
 * type 0 = ChangeDay: day a <- employee id b            (lib.rs:466-470)
 * type 1 = SwapDays : days a and b exchange employees   (lib.rs:471-478) */
typedef struct { int64_t type, a, b; } orc_sched_move;

/*
 * Neighbourhood pipeline local_search.rs:315-325 over an explicit move list:
 * candidate = clone + move; tabu == {current} drops candidates equal to the
 * current vector (over ALL stored entries, lib.rs:140); full rescoring;
 * take(window); minimum by (hard, soft, date_to_employee vector), first of
 * equals.  n_stored = len(date_to_employee).
 */
int orc_sched_neighbourhood(const orc_sched_instance *in, const int64_t *e, int64_t n_stored,
                            const orc_sched_move *moves, int64_t n_moves, int64_t window,
                            int64_t *hard_out, int64_t *soft_out, /* nullable, per move */
                            int64_t *n_scored, int64_t *best_index,
                            int64_t *best_hard, int64_t *best_soft)
{
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_stored);
    int64_t *best = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_stored);
    if (!cand || !best) { free(cand); free(best); return -1; }
    int64_t scored = 0, bi = -1, bh = 0, bs = 0;
    for (int64_t k = 0; k < n_moves; k++) {
        if (hard_out) hard_out[k] = ORC_SKIPPED;
        if (soft_out) soft_out[k] = ORC_SKIPPED;
    }
    for (int64_t k = 0; k < n_moves && scored < window; k++) {
        memcpy(cand, e, sizeof(int64_t) * (size_t)n_stored);
        if (moves[k].type == 0) {
            cand[moves[k].a] = moves[k].b;
        } else {
            cand[moves[k].a] = e[moves[k].b];
            cand[moves[k].b] = e[moves[k].a];
        }
        if (memcmp(cand, e, sizeof(int64_t) * (size_t)n_stored) == 0) continue; /* tabu */
        int64_t h, s;
        int rc = orc_sched_score(in, cand, &h, &s, NULL);
        if (rc) { free(cand); free(best); return rc; }
        if (hard_out) hard_out[k] = h;
        if (soft_out) soft_out[k] = s;
        scored++;
        int better = 0;
        if (bi < 0) better = 1;
        else if (h != bh) better = h < bh;
        else if (s != bs) better = s < bs;
        else {
            for (int64_t i = 0; i < n_stored; i++) {
                if (cand[i] != best[i]) { better = cand[i] < best[i]; break; }
            }
        }
        if (better) { bi = k; bh = h; bs = s; memcpy(best, cand, sizeof(int64_t) * (size_t)n_stored); }
    }
    *n_scored = scored; *best_index = bi; *best_hard = bh; *best_soft = bs;
    free(cand); free(best);
    return 0;
}

/* The same pipeline for a GENERAL tabu set (see orc_nq_neighbourhood_literal_tabu): `tabu` holds n_tabu stored
 * vectors; a candidate is dropped iff it equals one of them.  The base vector is not added implicitly. */
int orc_sched_neighbourhood_tabu(const orc_sched_instance *in, const int64_t *e, int64_t n_stored,
                                 const orc_sched_move *moves, int64_t n_moves, int64_t window,
                                 const int64_t *tabu, int64_t n_tabu,
                                 int64_t *hard_out, int64_t *soft_out,
                                 int64_t *n_scored, int64_t *best_index,
                                 int64_t *best_hard, int64_t *best_soft)
{
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_stored);
    int64_t *best = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_stored);
    if (!cand || !best) { free(cand); free(best); return -1; }
    int64_t scored = 0, bi = -1, bh = 0, bs = 0;
    for (int64_t k = 0; k < n_moves; k++) {
        if (hard_out) hard_out[k] = ORC_SKIPPED;
        if (soft_out) soft_out[k] = ORC_SKIPPED;
    }
    for (int64_t k = 0; k < n_moves && scored < window; k++) {
        memcpy(cand, e, sizeof(int64_t) * (size_t)n_stored);
        if (moves[k].type == 0) {
            cand[moves[k].a] = moves[k].b;
        } else {
            cand[moves[k].a] = e[moves[k].b];
            cand[moves[k].b] = e[moves[k].a];
        }
        int is_tabu = 0;
        for (int64_t t = 0; t < n_tabu && !is_tabu; t++)
            is_tabu = memcmp(cand, tabu + t * n_stored, sizeof(int64_t) * (size_t)n_stored) == 0;
        if (is_tabu) continue;
        int64_t h, sft;
        int rc = orc_sched_score(in, cand, &h, &sft, NULL);
        if (rc) { free(cand); free(best); return rc; }
        if (hard_out) hard_out[k] = h;
        if (soft_out) soft_out[k] = sft;
        scored++;
        int better = 0;
        if (bi < 0) better = 1;
        else if (h != bh) better = h < bh;
        else if (sft != bs) better = sft < bs;
        else {
            for (int64_t i = 0; i < n_stored; i++) {
                if (cand[i] != best[i]) { better = cand[i] < best[i]; break; }
            }
        }
        if (better) { bi = k; bh = h; bs = sft; memcpy(best, cand, sizeof(int64_t) * (size_t)n_stored); }
    }
    *n_scored = scored; *best_index = bi; *best_hard = bh; *best_soft = bs;
    free(cand); free(best);
    return 0;
}

/*
 * Exhaustive proposer order, lib.rs:526-550: for each STORED day index, walk
 * the ring next_employees (ids ascending, wrapping) starting after the day's
 * current employee; the last emitted candidate of a day equals the base.
 * employees: sorted ascending ids.  Emits ChangeDay moves into `out`
 * (capacity n_stored * n_employees); returns the count.  Employees not in the
 * ring would panic in the reference (unwrap) -> returns -2.
 */
int64_t orc_sched_exhaustive_moves(const int64_t *e, int64_t n_stored,
                                   const int64_t *employees, int64_t n_employees,
                                   orc_sched_move *out)
{
    int64_t cnt = 0;
    for (int64_t day = 0; day < n_stored; day++) {
        int64_t pos = -1;
        for (int64_t i = 0; i < n_employees; i++) if (employees[i] == e[day]) { pos = i; break; }
        if (pos < 0) return -2;
        for (int64_t step = 1; step <= n_employees; step++) {
            const int64_t next = employees[(pos + step) % n_employees];
            out[cnt].type = 0; out[cnt].a = day; out[cnt].b = next; cnt++;
        }
    }
    return cnt;
}
