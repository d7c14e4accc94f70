/* The reference's `employee-scheduling` binary (examples/employee-scheduling/src/main.rs:8-64) in plain C99 on the
 * C ABI of libls_amd.so: the built-in instance (2022-05-09 + 30 days = 31 scored days, employees 0..6, no holidays),
 * seed "42", window 100, LS max 1000, ILS 250 rounds, best set 64, no-improvement limit 20; rounds run until
 * is_finished (main.rs:48-50).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/employee_scheduling_c.c -o employee_scheduling_c \
 *       -Lconstraint-solver_amd/lib -lls_amd -Wl,-rpath,$PWD/constraint-solver_amd/lib
 */
#include <stdio.h>

#include "ls_amd.h"

enum { N_DAYS_SCORED = 31, N_EMPLOYEES = 7, START_WEEKDAY = 0 /* 2022-05-09 is a Monday */ };

static int fail(const char *what, ls_ctx *ctx)
{
    fprintf(stderr, "%s: %s\n", what, ls_last_error(ctx));
    return 1;
}

int main(int argc, char **argv)
{
    const char *seed = argc > 1 ? argv[1] : "42";
    static const char *weekday[7] = {"Mon", "Tue", "Wed", "Thu", "Fri", "Sat", "Sun"};
    printf("employee scheduling local search example\n");
    ls_ctx *ctx = NULL;
    if (ls_ctx_create(0, &ctx) != LS_OK) return fail("ls_ctx_create", NULL);

    ls_solver_params p;                       /* main.rs:22-31 */
    p.local_search_max_iterations = 1000;
    p.window_size = 100;
    p.best_solutions_capacity = 64;
    p.all_solutions_capacity = 100000;
    p.all_solution_iteration_expiry = 1000;
    p.iterated_local_search_max_iterations = 250;
    p.max_allow_no_improvement_for = 20;
    uint64_t holiday_bits[N_EMPLOYEES] = {0}; /* employee_to_holidays = HashMap::new() */

    ls_solver *s = NULL;
    if (ls_solver_create_sched(ctx, N_DAYS_SCORED, START_WEEKDAY, N_EMPLOYEES, holiday_bits, seed, &p, &s) != LS_OK)
        return fail("ls_solver_create_sched", ctx);
    int finished = 0;
    while (!finished) {
        if (ls_solver_execute_round(s) != LS_OK) return fail("ls_solver_execute_round", ctx);
        if (ls_solver_is_finished(s, &finished) != LS_OK) return fail("ls_solver_is_finished", ctx);
    }
    int64_t score[2] = {0, 0}, sol[N_DAYS_SCORED + 1];
    int32_t len = 0;
    int have = 0;
    if (ls_solver_get_best_solution(s, score, sol, N_DAYS_SCORED + 1, &len, &have) != LS_OK || !have)
        return fail("ls_solver_get_best_solution", ctx);
    printf("result.solution:\n[");
    for (int32_t d = 0; d < len; d++) printf("%s%lld", d ? ", " : "", (long long)sol[d]);
    printf("]\nresult.score: hard %lld, soft %lld\n---\n", (long long)score[0], (long long)score[1]);
    for (int e = 0; e < N_EMPLOYEES; e++) {   /* get_employees_to_days over the SCORED days, main.rs:57-63 */
        printf("employee: %d\n", e);
        for (int d = 0; d < N_DAYS_SCORED; d++)
            if (sol[d] == e) printf("%s - day %d\n", weekday[(START_WEEKDAY + d) % 7], d);
    }
    uint64_t it = 0, scored = 0, launches = 0;
    ls_solver_get_stats(s, &it, &scored, &launches);
    fprintf(stderr, "%llu LS iterations, %llu candidates scored on the GPU, %llu launches\n", (unsigned long long)it,
            (unsigned long long)scored, (unsigned long long)launches);
    ls_solver_free(s);
    ls_ctx_destroy(ctx);
    return 0;
}
