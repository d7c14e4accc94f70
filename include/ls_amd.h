/*
 * ls_amd.h — C ABI of the MI355X (gfx950) local-search neighbourhood evaluator.
 *
 * Drop-in boundary for ONE hot path of asimihsan/constraint-solver: the
 * neighbourhood pipeline of LocalSearch::execute
 * (local-search/src/local_search.rs:315-325)
 *
 *     move_proposer.iter_local_moves(&current, rng)      :316-317
 *         .filter(|s| !history.is_solution_tabu(s))      :319   (tabu == {current}, :155-199)
 *         .map(|s| calculator.get_scored_solution(s))    :320
 *         .take(window_size).collect();  sort(); first() :321-325
 *
 * for the two plug-ins the reference ships: N-queens
 * (examples/nqueens/src/lib.rs) and employee scheduling
 * (examples/employee-scheduling/src/lib.rs).  The reference has no FFI on this
 * path (it is monomorphised Rust); the only FFI precedent in the repository is
 * the wasm-bindgen opaque-context pattern
 * (web/employee-scheduling-wasm-bindgen/src/lib.rs:13-84), which this ABI
 * mirrors: create / call / query / destroy on an opaque handle.
 *
 * One call evaluates one WHOLE neighbourhood (or a batch of them, one per
 * independent restart) on the GPU and returns what `first()` would return:
 * the minimum by (score, solution-vector) plus the number of candidates
 * scored.  Move selection, acceptance, RNG and history stay with the caller.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns an int status
 *    (LS_OK == 0); nothing throws or aborts across the boundary (the reference
 *    panics via unwrap(); see ls_last_error()).
 *  - the caller owns every host buffer for the duration of the call; no
 *    pointer is retained after return.  Device buffers passed to the *_dev
 *    entry points stay owned by the caller as well.
 *  - a context is used by one host thread at a time; one context per GPU;
 *    different contexts may be driven concurrently from different threads
 *    (reference: Solution/Score are Send, nothing is Sync,
 *    local-search/src/local_search.rs:16-23).
 *  - There is NO CPU fallback: without a usable HIP device ls_ctx_create
 *    fails with LS_ERR_NO_DEVICE.
 */
#ifndef LS_AMD_H
#define LS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS_ABI_VERSION 1

/* status codes */
#define LS_OK               0
#define LS_ERR_INVALID      1   /* bad argument (null pointer, size, out-of-range value) */
#define LS_ERR_NO_DEVICE    2   /* no HIP device / device id out of range */
#define LS_ERR_HIP          3   /* a HIP runtime call failed; text via ls_last_error() */
#define LS_ERR_UNSUPPORTED  4   /* size outside what the kernels support */
#define LS_ERR_NOMEM        5
#define LS_ERR_STATE        6   /* call order (e.g. schedule eval before instance upload) */

/* N-queens kernel families (ls_ctx_set_nq_algo) */
#define LS_NQ_ALGO_SCAN   0  /* one wavefront per candidate scans the packed state: the reference's O(n) pair checks */
#define LS_NQ_ALGO_TABLE  1  /* occupancy count tables, O(1) per candidate */
#define LS_NQ_ALGO_AUTO   2  /* default: the fastest family for the launch (today: TABLE, fused LS step when it fits) */

/* limits */
#define LS_NQ_MAX_N        32768
#define LS_SCHED_MAX_DAYS  64

typedef struct ls_ctx ls_ctx;

/* ------------------------------------------------------------------ context */

int ls_abi_version(void);

/* Number of HIP devices visible (0 when none / no driver). */
int ls_device_count(void);

/* Opaque context bound to one GPU; owns a HIP stream and persistent device
 * workspaces.  Mirrors create_solver(..) -> SolverContext
 * (web/employee-scheduling-wasm-bindgen/src/lib.rs:18-53). */
int ls_ctx_create(int device_id, ls_ctx **out);

/* Mirrors the JS-side solver.free() (web/employee-scheduling/src/worker.ts:9-17). */
int ls_ctx_destroy(ls_ctx *ctx);

/* Text of the last error on this context ("" if none); owned by the context.
 * ls_last_error(NULL) returns the last error of a failed ls_ctx_create. */
const char *ls_last_error(const ls_ctx *ctx);

/* Launch on a caller-owned hipStream_t (e.g. PyTorch's current stream) instead
 * of the context's own stream; NULL restores the own stream. */
int ls_ctx_set_stream(ls_ctx *ctx, void *hip_stream);

/* Block until everything queued on the context's stream has finished. */
int ls_ctx_synchronize(ls_ctx *ctx);

int ls_ctx_set_nq_algo(ls_ctx *ctx, int algo);
int ls_ctx_get_nq_algo(const ls_ctx *ctx, int *algo);

/* Counters of the latency-critical host-pointer path since the context was created (ls_multi_solver_get_counters: of
 * the last run).  The reference has no counterpart (its loop, local_search.rs:301-342, never leaves the thread); they
 * exist so that the slow branches of this boundary are visible instead of silent:
 *   [0] requests served by the resident LS-step servers (N-queens: one launch whose workgroups are the slots; scheduling:
 *       one workgroup behind ls_sched_eval_resident / ls_sched_score for lists of <= 512 moves; a one-restart
 *       ls_nqueens_eval / ls_nqueens_eval_batch[_next] / ls_nqueens_col_scores call is a 64-byte mailbox write and a
 *       32-byte answer, no kernel launch), [1] launches of that server (it leaves when no slot has been served for 2 ms
 *       -- LS_NQ_SERVER_IDLE_US -- and is started again on demand), [2] waits that found the launch drained under their
 *       request, [3] boards uploaded to server slots (a board that continues from the previous winner is not),
 *   [4] completion-word waits of launched steps that fell back to hipStreamSynchronize after 2 ms,
 *   [5] boards re-uploaded by the launch-per-call route (LS_NQ_STEP=3) because they were not the resident one,
 *   [6] (ls_multi_solver only) restarts that an idle host thread took over from a busy one (slots beyond the cores are
 *       cooperative fibres of one pool). */
#define LS_N_COUNTERS 7
int ls_ctx_get_counters(const ls_ctx *ctx, uint64_t out[LS_N_COUNTERS]);

/* Device properties the harness reports (CU count, wavefront size, clock kHz, name). */
int ls_ctx_device_info(const ls_ctx *ctx, int *n_cus, int *wave_size, int *clock_khz,
                       char *name, int name_cap);

/* ----------------------------------------------------------------- N-queens
 *
 * State: rows[col] = row, packed int32 (reference: Vec<i64>,
 * examples/nqueens/src/lib.rs:13-21), 0 <= rows[col] < n, 1 <= n <= LS_NQ_MAX_N.
 *
 * Candidate order (MoveIterator, examples/nqueens/src/lib.rs:214-234): for
 * cols[0..n_cols) in the proposer's order, values 0..n-1; candidate
 * k = (col = cols[k / n], value = k % n) is the current board with
 * rows[col] = value.  Candidates equal to the current board are tabu and
 * neither scored nor counted (local_search.rs:319); only the first `window`
 * remaining candidates are scored (:321).  Score = sum of per-column conflict
 * counts (lib.rs:126-140).  Winner = minimum by (score, board vector), the
 * derived Ord of ScoredSolution (local_search.rs:29-37).
 */
typedef struct ls_nq_result {
    int64_t n_scored;    /* candidates scored (after tabu filter and window)            */
    int64_t best_index;  /* winner's position k in the UNFILTERED order; -1 if none    */
    int64_t best_score;  /* its score (NQueensScore)                                    */
    int32_t best_col;    /* cols[best_index / n]                                        */
    int32_t best_value;  /* best_index % n                                              */
} ls_nq_result;

/* Replaces local_search.rs:315-325 for ONE neighbourhood.  cur_score must be
 * the score of `rows` (the caller always has it: local_search.rs:306/:335).
 * scores_out (nullable, n_cols*n entries, parity/debug mode): per-candidate
 * scores indexed by k; INT64_MIN where the candidate was tabu or beyond the
 * window. */
int ls_nqueens_eval(ls_ctx *ctx, int32_t n, const int32_t *rows, int64_t cur_score,
                    const int32_t *cols, int32_t n_cols, int64_t window,
                    ls_nq_result *out, int64_t *scores_out);

/* Batched over independent restarts (one neighbourhood each, one launch):
 * rows[n_restarts][n], cur_scores[n_restarts], cols[n_restarts][max_cols]
 * (only the first n_cols[r] of each row are read), results[n_restarts],
 * scores_out (nullable) [n_restarts][max_cols*n]. */
int ls_nqueens_eval_batch(ls_ctx *ctx, int32_t n_restarts, int32_t n,
                          const int32_t *rows, const int64_t *cur_scores,
                          const int32_t *cols, const int32_t *n_cols, int32_t max_cols,
                          int64_t window, ls_nq_result *results, int64_t *scores_out);

/* Same, every pointer in DEVICE memory; asynchronous on the context's stream
 * (no host synchronisation, graph-capturable).  cols are NOT validated: a
 * column outside [0, n) contributes no candidates. */
int ls_nqueens_eval_batch_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n,
                              const int32_t *rows_d, const int64_t *cur_scores_d,
                              const int32_t *cols_d, const int32_t *n_cols_d, int32_t max_cols,
                              int64_t window, ls_nq_result *results_d, int64_t *scores_out_d);

/* Neighbourhood AND the proposer's input for the next LS iteration in one round trip: besides the
 * results, next_col_scores[n_restarts][n] receives get_col_scores (lib.rs:74-87) of each restart's
 * winner board (rows with rows[best_col] = best_value; of rows itself when nothing was scored) —
 * what iter_local_moves computes first when the loop continues from the winner
 * (local_search.rs:335, nqueens lib.rs:182).  The winner's total is results[r].best_score. */
int ls_nqueens_eval_batch_next(ls_ctx *ctx, int32_t n_restarts, int32_t n,
                               const int32_t *rows, const int64_t *cur_scores,
                               const int32_t *cols, const int32_t *n_cols, int32_t max_cols,
                               int64_t window, ls_nq_result *results, int64_t *next_col_scores);
/* Device-pointer flavour (asynchronous): next_rows_d[n_restarts][n] receives the winner boards,
 * next_col_scores_d[n_restarts][n] their column scores, next_scores_d[n_restarts] (nullable) the sums. */
int ls_nqueens_eval_batch_next_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n,
                                   const int32_t *rows_d, const int64_t *cur_scores_d,
                                   const int32_t *cols_d, const int32_t *n_cols_d, int32_t max_cols,
                                   int64_t window, ls_nq_result *results_d, int32_t *next_rows_d,
                                   int64_t *next_col_scores_d, int64_t *next_scores_d);

/* ls_nqueens_eval for a History whose tabu set holds more than the current board (all_solution_iteration_expiry == 0:
 * the reference's set then keeps every visited solution, local-search/src/local_search.rs:155-199; no configuration the
 * reference ships does this).  tabu_positions: the unfiltered positions k = slot * n + value of the candidates that equal
 * a tabu board OTHER than `rows` itself (the no-ops are always tabu and need not be listed), strictly ascending, each a
 * real candidate (value != rows[cols[slot]]).  They are neither scored nor counted: candidates behind them move up in
 * the window exactly as after the reference's filter (:319-321).  n_tabu == 0 is ls_nqueens_eval. */
int ls_nqueens_eval_tabu(ls_ctx *ctx, int32_t n, const int32_t *rows, int64_t cur_score, const int32_t *cols,
                         int32_t n_cols, int64_t window, const int64_t *tabu_positions, int32_t n_tabu,
                         ls_nq_result *out, int64_t *scores_out);

/* get_col_scores + its sum (examples/nqueens/src/lib.rs:74-87,130-139) for a
 * batch of boards, by occupancy count tables: col_scores[n_restarts][n]
 * (nullable), scores[n_restarts] (nullable).  This is the vector the move
 * proposer needs per LS iteration (lib.rs:182-187). */
int ls_nqueens_col_scores(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows,
                          int64_t *col_scores, int64_t *scores);
int ls_nqueens_col_scores_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows_d,
                              int64_t *col_scores_d, int64_t *scores_d);

/* ------------------------------------------------------ employee scheduling
 *
 * State: date_to_employee as int32 employee INDICES into the instance's
 * ascending `employees` vector (reference: Vec<Employee{id:i64}>,
 * examples/employee-scheduling/src/lib.rs:119-146; ascending ids <-> ascending
 * indices, so the derived vector order is preserved).  n_days_stored entries
 * (the reference stores one more than it scores, lib.rs:407-411);
 * the first n_days_scored are scored (lib.rs:181-192).
 */
#define LS_SCHED_MOVE_CHANGE_DAY 0
#define LS_SCHED_MOVE_SWAP_DAYS  1
#define LS_SCHED_MOVE_SKIP       2  /* not a candidate: the caller found the solution this move leads to in its tabu set
                                       (a History holding more than the current solution, local_search.rs:155-199);
                                       neither scored nor counted, exactly like a move that reproduces the current state */
typedef struct ls_sched_move {
    int32_t type; /* 0 = ChangeDay: day a <- employee index b   (lib.rs:466-470)
                     1 = SwapDays : days a and b swap employees (lib.rs:471-478)
                     2 = skip (a, b ignored) */
    int32_t a;
    int32_t b;
} ls_sched_move;

typedef struct ls_sched_result {
    int64_t n_scored;
    int64_t best_index;  /* position in the move list; -1 if none */
    int64_t best_hard;   /* ScheduleScore.hard_score (integer-valued f64 in the reference) */
    int64_t best_soft;   /* ScheduleScore.soft_score */
} ls_sched_result;

/* Instance = what ScheduleSolutionScoreCalculator::new + the solution's date
 * range hold (lib.rs:251-259,129-146): holiday_bits[e] bit d set <=> employee
 * index e has a holiday on day index d (d < n_days_scored).
 * start_weekday: Monday = 0 .. Sunday = 6 (chrono::Weekday). */
int ls_sched_instance_upload(ls_ctx *ctx, int32_t n_days_scored, int32_t n_days_stored,
                             int32_t start_weekday, int32_t n_employees,
                             const uint64_t *holiday_bits);

/* get_scored_solution (lib.rs:261-375) for n_states stored vectors:
 * states[n_states][n_days_stored] -> hard[n_states], soft[n_states];
 * terms (nullable) [n_states][8] = H1..H4, S1..S4. */
int ls_sched_score(ls_ctx *ctx, int32_t n_states, const int32_t *states,
                   int64_t *hard, int64_t *soft, int64_t *terms);

/* Replaces local_search.rs:315-325 for one neighbourhood given the proposer's
 * move stream (ScheduleRandomMoveProposer lib.rs:444-490 or ScheduleMoveProposer
 * :515-558, both expressible as ls_sched_move lists).  Candidates equal to the
 * current state are tabu; the first `window` others are scored; winner =
 * minimum by (hard, soft, date_to_employee vector).  hard_out/soft_out
 * (nullable, n_moves entries): per-candidate scores, INT64_MIN where skipped. */
int ls_sched_eval(ls_ctx *ctx, const int32_t *state, const ls_sched_move *moves,
                  int64_t n_moves, int64_t window, ls_sched_result *out,
                  int64_t *hard_out, int64_t *soft_out);

/* The random proposer's move stream never changes during a run (its iterator owns a CLONE of the
 * generator, lib.rs:488, and nothing advances the original), so it can stay on the device:
 * ls_sched_moves_upload replaces the context's resident list; ls_sched_eval_resident evaluates its
 * first n_moves entries against `state` exactly like ls_sched_eval. */
int ls_sched_moves_upload(ls_ctx *ctx, const ls_sched_move *moves, int64_t n_moves);

/* ScheduleMoveProposer::iter_local_moves (examples/employee-scheduling/src/lib.rs:493-559), the exhaustive
 * proposer (constructed nowhere in the reference, :59), as a move list in its order: for every STORED day d
 * (:530), ChangeDay(d, e) for e = next(current), next(next(current)), ... in id order with wrap-around
 * (next_employees, :499-508), the day's current employee last (:541-543: a no-op, dropped by the tabu filter).
 * moves: [n_restarts][n_days_stored * n_employees], generated on the device from the states.
 *   ls_sched_exhaustive_moves      host states in, host moves out (synchronous)
 *   ls_sched_exhaustive_moves_dev  device pointers, asynchronous on the context's stream
 *   ls_sched_eval_exhaustive       lines :315-325 over that neighbourhood: generate + score + argmin on the device,
 *                                  only the states (in) and the result records (out) cross the boundary;
 *                                  best_index is the position in the list above */
int ls_sched_exhaustive_moves(ls_ctx *ctx, int32_t n_restarts, const int32_t *states, ls_sched_move *moves_out);
int ls_sched_exhaustive_moves_dev(ls_ctx *ctx, int32_t n_restarts, const int32_t *states_d, ls_sched_move *moves_d);
int ls_sched_eval_exhaustive(ls_ctx *ctx, int32_t n_restarts, const int32_t *states, int64_t window,
                             ls_sched_result *results);
int ls_sched_eval_resident(ls_ctx *ctx, const int32_t *state, int64_t n_moves, int64_t window, ls_sched_result *out);

/* Batched over restarts: states[n_restarts][n_days_stored],
 * moves[n_restarts][max_moves] (first n_moves[r] used), results[n_restarts]. */
int ls_sched_eval_batch(ls_ctx *ctx, int32_t n_restarts, const int32_t *states,
                        const ls_sched_move *moves, const int64_t *n_moves, int64_t max_moves,
                        int64_t window, ls_sched_result *results,
                        int64_t *hard_out, int64_t *soft_out);

/* Device-pointer flavour, asynchronous on the context's stream. */
int ls_sched_eval_batch_dev(ls_ctx *ctx, int32_t n_restarts, const int32_t *states_d,
                            const ls_sched_move *moves_d, const int64_t *n_moves_d,
                            int64_t max_moves, int64_t window, ls_sched_result *results_d,
                            int64_t *hard_out_d, int64_t *soft_out_d);

/* ------------------------------------------------------------ whole solver
 *
 * Host-side mirror of the reference's driver around the GPU path, shaped like the only FFI
 * the reference has (web/employee-scheduling-wasm-bindgen/src/lib.rs:18-84: create_solver /
 * execute_solver_round / get_iteration_info / is_solver_finished / get_best_solution / free).
 * IteratedLocalSearch::execute_round (iterated_local_search.rs:173-202), LocalSearch::execute
 * (local_search.rs:301-342), History, AcceptanceCriterion, the plug-ins' proposers,
 * perturbations and initial solutions run on the host exactly as in the reference; every score
 * and every neighbourhood (local_search.rs:315-325) runs on the GPU of `ctx`.
 * Seeds: seed string -> Blake2b-256 -> two identically seeded ChaCha20 generators
 * (examples/nqueens/src/main.rs:28-39,66).  The rand/rand_chacha algorithms are restated from
 * their published sources; equality with the Rust binaries' streams is unverified (no Rust
 * toolchain in the build environment).
 */
typedef struct ls_solver ls_solver;

typedef struct ls_solver_params {        /* MainArgs, examples/nqueens/src/main.rs:16-26 */
    uint64_t local_search_max_iterations;
    uint64_t window_size;
    uint64_t best_solutions_capacity;
    uint64_t all_solutions_capacity;
    uint64_t all_solution_iteration_expiry;   /* >= 1 (every shipped configuration): tabu == {current}, SURVEY 3.4-1;
                                                 0: the tabu set grows (served through ls_nqueens_eval_tabu / LS_SCHED_MOVE_SKIP) */
    uint64_t iterated_local_search_max_iterations;
    uint64_t max_allow_no_improvement_for;
} ls_solver_params;

/* get_solution's set-up (examples/nqueens/src/main.rs:35-87).  The context must outlive the solver. */
int ls_solver_create_nqueens(ls_ctx *ctx, uint64_t board_size, const char *seed, const ls_solver_params *params,
                             ls_solver **out);
/* get_ils (examples/employee-scheduling/src/lib.rs:57-117).  Employees are indices 0..n_employees-1
 * in ascending id order; uploads the instance to `ctx`.  n_days_stored = n_days_scored + 1
 * (the generator's off-by-one, lib.rs:407-411). */
int ls_solver_create_sched(ls_ctx *ctx, int32_t n_days_scored, int32_t start_weekday, int32_t n_employees,
                           const uint64_t *holiday_bits, const char *seed, const ls_solver_params *params,
                           ls_solver **out);
int ls_solver_execute_round(ls_solver *s);                                   /* execute_solver_round */
int ls_solver_is_finished(const ls_solver *s, int *finished);                 /* is_solver_finished */
int ls_solver_get_iteration_info(const ls_solver *s, uint64_t *current, uint64_t *total); /* get_iteration_info */
/* get_best_solution / the current local minimum.  score[0] = NQueensScore or hard, score[1] = 0 or soft;
 * solution: rows (N-queens) or employee indices (scheduling), `cap` entries available, *len written.
 * *have = 0 when the best set is still empty (the reference would panic on unwrap, :165-167). */
int ls_solver_get_best_solution(const ls_solver *s, int64_t score[2], int64_t *solution, int32_t cap, int32_t *len,
                                int *have);
int ls_solver_get_current_solution(const ls_solver *s, int64_t score[2], int64_t *solution, int32_t cap, int32_t *len);
/* counters: LS iterations run, candidates scored on the GPU, GPU neighbourhood launches */
int ls_solver_get_stats(const ls_solver *s, uint64_t *ls_iterations, uint64_t *neighbours_scored, uint64_t *gpu_evals);
int ls_solver_free(ls_solver *s);

/* Many independent restarts on one GPU (the reference's only multi-seed convention is
 * consecutive seed strings, examples/nqueens/src/main.rs:167).  Each restart is a complete, unchanged solver
 * on a host thread (or fibre); n_slots of them are in flight at once and a finished restart's slot takes the next seed of
 * the list, so the GPU stays fed while restarts need very different numbers of iterations.  The result of
 * every restart is identical to running it alone with ls_solver_*.  Two ways to serve the slots' GPU requests:
 *   LS_MULTI_FREE_RUNNING  (default) every slot is one workgroup of a single resident LS-step server launch on the
 *                          device (board and count tables stay in that workgroup's LDS; an LS iteration is a 64-byte
 *                          request and a 32-byte answer, no kernel launch); no slot ever waits for another.
 *                          (LS_NQ_STEP=3, LS_NQ_ALGO_SCAN or n > 8192: a context per slot issuing one-restart launches);
 *   LS_MULTI_LOCKSTEP      the slots' requests of one step are merged into ONE batched launch on the caller's
 *                          context (ls_nqueens_eval_batch_next / ls_nqueens_col_scores).
 * ls_multi_solver_run drives all restarts until each is finished, has found a best-possible solution, or a
 * limit is reached (restarts not started before max_seconds elapsed report have = 0). */
#define LS_MULTI_LOCKSTEP      0
#define LS_MULTI_FREE_RUNNING  1
typedef struct ls_multi_solver ls_multi_solver;
int ls_multi_solver_create_nqueens(ls_ctx *ctx, uint64_t board_size, int32_t n_restarts, const char *const *seeds,
                                   const ls_solver_params *params, ls_multi_solver **out);
/* The same for the scheduling plug-in: n_restarts complete, unchanged get_ils drivers (examples/employee-scheduling/src/lib.rs:57-117;
 * instance as for ls_solver_create_sched, seeds by the convention of examples/nqueens/src/main.rs:167) on one GPU.  The slots'
 * GPU requests go to ONE resident launch shared by all of them: slot t is workgroup t of the scheduling LS-step server (move
 * lists of at most 512 entries: the reference example's window 100, examples/employee-scheduling/src/main.rs:26) or sub-grid t
 * of the resident grid (longer lists: a window of 65 536 over 10 000 employees) -- no kernel launch per LS iteration, no slot
 * waits for another.  Free-running only (LS_MULTI_LOCKSTEP is refused by ls_multi_solver_configure's caller contract: the
 * mode argument is ignored for scheduling).  ls_multi_solver_get_best returns score[0] = hard, score[1] = soft and the employee
 * indices of the n_days_scored + 1 stored days; every restart's result equals its ls_solver_create_sched run. */
int ls_multi_solver_create_sched(ls_ctx *ctx, int32_t n_days_scored, int32_t start_weekday, int32_t n_employees,
                                 const uint64_t *holiday_bits, int32_t n_restarts, const char *const *seeds,
                                 const ls_solver_params *params, ls_multi_solver **out);
/* n_slots in [1, 256], or 0 for the mode's default (default: four per core the process may use -- eight for a scheduling solver --; free-running slots beyond the cores share host
 * threads as cooperative fibres: a slot whose launch is in flight lends its thread to a slot with host work to do;
 * lockstep slots are one host thread each), mode as above; call before ls_multi_solver_run.  For a scheduling solver the call also
 * makes what the slots share -- the resident launch's host object, a light context per slot -- which the first run would make otherwise
 * (set-up, like ls_solver_create_sched's, not part of a run). */
int ls_multi_solver_configure(ls_multi_solver *m, int32_t n_slots, int32_t mode);
int ls_multi_solver_run(ls_multi_solver *m, uint64_t max_rounds_per_restart, double max_seconds);
int ls_multi_solver_get_best(const ls_multi_solver *m, int32_t restart, int64_t score[2], int64_t *solution,
                             int32_t cap, int32_t *len, int *have);
/* totals over all restarts: LS iterations, candidates scored, batched GPU launches, rounds executed */
int ls_multi_solver_get_stats(const ls_multi_solver *m, uint64_t *ls_iterations, uint64_t *neighbours_scored,
                              uint64_t *gpu_launches, uint64_t *rounds);
/* one restart of the last run (the same figures ls_solver_get_stats reports for a restart run alone: "same seed => same
 * result", examples/nqueens/src/main.rs:156-200, extends to the iteration counts) */
int ls_multi_solver_get_restart_stats(const ls_multi_solver *m, int32_t restart, uint64_t *ls_iterations,
                                      uint64_t *neighbours_scored, uint64_t *rounds);
int ls_multi_solver_get_counters(const ls_multi_solver *m, uint64_t out[LS_N_COUNTERS]); /* see ls_ctx_get_counters */
int ls_multi_solver_free(ls_multi_solver *m);

/* Host RNG primitives, exported so the tests can pin them (RFC 7693 / RFC 7539 vectors, and
 * the oracle's independent restatement).  ls_host_rng_script runs `n_ops` operations
 * (op, a, b) on a ChaCha20Rng::from_seed(seed32) and appends every produced value to `out`:
 *   0 next_u32 | 1 next_u64 | 2 gen_range_u64 a..=b | 3 gen_range_u32 a..=b | 4 Uniform<u64>(0,a) |
 *   5 Uniform<u32>(0,a) | 6 shuffle of 0..a (a values) | 7 index::sample(a, b) (b values) |
 *   8 gen::<f64>() bit pattern | 9 index::sample_weighted(weights[0..a], b) (b values) */
int ls_host_hash_seed(const char *seed, uint8_t out32[32]);
int ls_host_chacha20_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                           uint32_t out16[16]);
int ls_host_rng_script(const uint8_t seed32[32], const int64_t *ops, int64_t n_ops, const double *weights,
                       uint64_t *out, int64_t cap, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* LS_AMD_H */
