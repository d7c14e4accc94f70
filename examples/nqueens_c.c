/* The reference's `nqueens` binary (examples/nqueens/src/main.rs:95-151) in plain C99 on the C ABI of libls_amd.so:
 * same defaults (window = 5 * board_size, LS / ILS max 10 000 iterations, best set 32, no-improvement limit 5),
 * same stopping rule (until finished or a score-0 solution is in the best set), same output shape.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/nqueens_c.c -o nqueens_c \
 *       -Lconstraint-solver_amd/lib -lls_amd -Wl,-rpath,$PWD/constraint-solver_amd/lib
 *   ./nqueens_c [board_size = 8] [seed = 42]
 */
#include <stdio.h>
#include <stdlib.h>

#include "ls_amd.h"

static int fail(const char *what, ls_ctx *ctx)
{
    fprintf(stderr, "%s: %s\n", what, ls_last_error(ctx));
    return 1;
}

int main(int argc, char **argv)
{
    const uint64_t board_size = argc > 1 ? strtoull(argv[1], NULL, 10) : 8;
    const char *seed = argc > 2 ? argv[2] : "42";
    ls_ctx *ctx = NULL;
    if (ls_ctx_create(0, &ctx) != LS_OK) return fail("ls_ctx_create", NULL);

    ls_solver_params p;                       /* main.rs:129-135 */
    p.local_search_max_iterations = 10000;
    p.window_size = board_size * 5;
    p.best_solutions_capacity = 32;
    p.all_solutions_capacity = 100000;
    p.all_solution_iteration_expiry = 10000;
    p.iterated_local_search_max_iterations = 10000;
    p.max_allow_no_improvement_for = 5;

    ls_solver *s = NULL;
    if (ls_solver_create_nqueens(ctx, board_size, seed, &p, &s) != LS_OK) return fail("ls_solver_create_nqueens", ctx);
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)board_size);
    int64_t score[2] = {0, 0};
    int32_t len = 0;
    int have = 0, finished = 0;
    while (!finished) {                       /* main.rs:77-86: run rounds until done or solved */
        if (ls_solver_execute_round(s) != LS_OK) return fail("ls_solver_execute_round", ctx);
        if (ls_solver_get_best_solution(s, score, rows, (int32_t)board_size, &len, &have) != LS_OK)
            return fail("ls_solver_get_best_solution", ctx);
        if (have && score[0] == 0) break;
        if (ls_solver_is_finished(s, &finished) != LS_OK) return fail("ls_solver_is_finished", ctx);
    }
    uint64_t it = 0, scored = 0, launches = 0;
    ls_solver_get_stats(s, &it, &scored, &launches);
    printf("result.solution:\n[");
    for (int32_t i = 0; i < len; i++) printf("%s%lld", i ? ", " : "", (long long)rows[i]);
    printf("]\nresult.score: %lld\n", have ? (long long)score[0] : -1LL);
    fprintf(stderr, "%llu LS iterations, %llu candidates scored on the GPU, %llu launches\n", (unsigned long long)it,
            (unsigned long long)scored, (unsigned long long)launches);
    free(rows);
    ls_solver_free(s);
    ls_ctx_destroy(ctx);
    return have && score[0] == 0 ? 0 : 2;
}
