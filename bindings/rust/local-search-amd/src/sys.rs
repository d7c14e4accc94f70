//! Raw bindings of include/ls_amd.h (ABI version 1).  One declaration per C function.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)] pub struct ls_ctx { _private: [u8; 0] }
#[repr(C)] pub struct ls_solver { _private: [u8; 0] }
#[repr(C)] pub struct ls_multi_solver { _private: [u8; 0] }

pub const LS_OK: c_int = 0;
pub const LS_NQ_ALGO_SCAN: c_int = 0;
pub const LS_NQ_ALGO_TABLE: c_int = 1;
pub const LS_NQ_ALGO_AUTO: c_int = 2;

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct ls_nq_result { pub n_scored: i64, pub best_index: i64, pub best_score: i64, pub best_col: i32, pub best_value: i32 }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct ls_sched_move { pub r#type: i32, pub a: i32, pub b: i32 }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct ls_sched_result { pub n_scored: i64, pub best_index: i64, pub best_hard: i64, pub best_soft: i64 }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct ls_solver_params {
    pub local_search_max_iterations: u64, pub window_size: u64, pub best_solutions_capacity: u64,
    pub all_solutions_capacity: u64, pub all_solution_iteration_expiry: u64,
    pub iterated_local_search_max_iterations: u64, pub max_allow_no_improvement_for: u64,
}

extern "C" {
    pub fn ls_abi_version() -> c_int;
    pub fn ls_device_count() -> c_int;
    pub fn ls_ctx_create(device_id: c_int, out: *mut *mut ls_ctx) -> c_int;
    pub fn ls_ctx_destroy(ctx: *mut ls_ctx) -> c_int;
    pub fn ls_last_error(ctx: *const ls_ctx) -> *const c_char;
    pub fn ls_ctx_set_stream(ctx: *mut ls_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn ls_ctx_synchronize(ctx: *mut ls_ctx) -> c_int;
    pub fn ls_ctx_set_nq_algo(ctx: *mut ls_ctx, algo: c_int) -> c_int;
    pub fn ls_ctx_get_nq_algo(ctx: *const ls_ctx, algo: *mut c_int) -> c_int;
    pub fn ls_ctx_get_counters(ctx: *const ls_ctx, out: *mut u64) -> c_int;
    pub fn ls_ctx_device_info(ctx: *const ls_ctx, n_cus: *mut c_int, wave_size: *mut c_int, clock_khz: *mut c_int,
                              name: *mut c_char, name_cap: c_int) -> c_int;
    pub fn ls_nqueens_eval(ctx: *mut ls_ctx, n: i32, rows: *const i32, cur_score: i64, cols: *const i32, n_cols: i32,
                           window: i64, out: *mut ls_nq_result, scores_out: *mut i64) -> c_int;
    pub fn ls_nqueens_eval_tabu(ctx: *mut ls_ctx, n: i32, rows: *const i32, cur_score: i64, cols: *const i32, n_cols: i32,
                                window: i64, tabu_positions: *const i64, n_tabu: i32, out: *mut ls_nq_result,
                                scores_out: *mut i64) -> c_int;
    pub fn ls_nqueens_eval_batch(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows: *const i32, cur_scores: *const i64,
                                 cols: *const i32, n_cols: *const i32, max_cols: i32, window: i64,
                                 results: *mut ls_nq_result, scores_out: *mut i64) -> c_int;
    pub fn ls_nqueens_eval_batch_dev(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows_d: *const i32,
                                     cur_scores_d: *const i64, cols_d: *const i32, n_cols_d: *const i32, max_cols: i32,
                                     window: i64, results_d: *mut ls_nq_result, scores_out_d: *mut i64) -> c_int;
    pub fn ls_nqueens_eval_batch_next(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows: *const i32, cur_scores: *const i64,
                                      cols: *const i32, n_cols: *const i32, max_cols: i32, window: i64,
                                      results: *mut ls_nq_result, next_col_scores: *mut i64) -> c_int;
    pub fn ls_nqueens_eval_batch_next_dev(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows_d: *const i32,
                                          cur_scores_d: *const i64, cols_d: *const i32, n_cols_d: *const i32,
                                          max_cols: i32, window: i64, results_d: *mut ls_nq_result,
                                          next_rows_d: *mut i32, next_col_scores_d: *mut i64,
                                          next_scores_d: *mut i64) -> c_int;
    pub fn ls_nqueens_col_scores(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows: *const i32, col_scores: *mut i64,
                                 scores: *mut i64) -> c_int;
    pub fn ls_nqueens_col_scores_dev(ctx: *mut ls_ctx, n_restarts: i32, n: i32, rows_d: *const i32,
                                     col_scores_d: *mut i64, scores_d: *mut i64) -> c_int;
    pub fn ls_sched_instance_upload(ctx: *mut ls_ctx, n_days_scored: i32, n_days_stored: i32, start_weekday: i32,
                                    n_employees: i32, holiday_bits: *const u64) -> c_int;
    pub fn ls_sched_score(ctx: *mut ls_ctx, n_states: i32, states: *const i32, hard: *mut i64, soft: *mut i64,
                          terms: *mut i64) -> c_int;
    pub fn ls_sched_eval(ctx: *mut ls_ctx, state: *const i32, moves: *const ls_sched_move, n_moves: i64, window: i64,
                         out: *mut ls_sched_result, hard_out: *mut i64, soft_out: *mut i64) -> c_int;
    pub fn ls_sched_moves_upload(ctx: *mut ls_ctx, moves: *const ls_sched_move, n_moves: i64) -> c_int;
    pub fn ls_sched_exhaustive_moves(ctx: *mut ls_ctx, n_restarts: i32, states: *const i32,
                                     moves_out: *mut ls_sched_move) -> c_int;
    pub fn ls_sched_exhaustive_moves_dev(ctx: *mut ls_ctx, n_restarts: i32, states_d: *const i32,
                                         moves_d: *mut ls_sched_move) -> c_int;
    pub fn ls_sched_eval_exhaustive(ctx: *mut ls_ctx, n_restarts: i32, states: *const i32, window: i64,
                                    results: *mut ls_sched_result) -> c_int;
    pub fn ls_sched_eval_resident(ctx: *mut ls_ctx, state: *const i32, n_moves: i64, window: i64,
                                  out: *mut ls_sched_result) -> c_int;
    pub fn ls_sched_eval_batch(ctx: *mut ls_ctx, n_restarts: i32, states: *const i32, moves: *const ls_sched_move,
                               n_moves: *const i64, max_moves: i64, window: i64, results: *mut ls_sched_result,
                               hard_out: *mut i64, soft_out: *mut i64) -> c_int;
    pub fn ls_sched_eval_batch_dev(ctx: *mut ls_ctx, n_restarts: i32, states_d: *const i32,
                                   moves_d: *const ls_sched_move, n_moves_d: *const i64, max_moves: i64, window: i64,
                                   results_d: *mut ls_sched_result, hard_out_d: *mut i64, soft_out_d: *mut i64) -> c_int;
    pub fn ls_solver_create_nqueens(ctx: *mut ls_ctx, board_size: u64, seed: *const c_char,
                                    params: *const ls_solver_params, out: *mut *mut ls_solver) -> c_int;
    pub fn ls_solver_create_sched(ctx: *mut ls_ctx, n_days_scored: i32, start_weekday: i32, n_employees: i32,
                                  holiday_bits: *const u64, seed: *const c_char, params: *const ls_solver_params,
                                  out: *mut *mut ls_solver) -> c_int;
    pub fn ls_solver_execute_round(s: *mut ls_solver) -> c_int;
    pub fn ls_solver_is_finished(s: *const ls_solver, finished: *mut c_int) -> c_int;
    pub fn ls_solver_get_iteration_info(s: *const ls_solver, current: *mut u64, total: *mut u64) -> c_int;
    pub fn ls_solver_get_best_solution(s: *const ls_solver, score: *mut i64, solution: *mut i64, cap: i32,
                                       len: *mut i32, have: *mut c_int) -> c_int;
    pub fn ls_solver_get_current_solution(s: *const ls_solver, score: *mut i64, solution: *mut i64, cap: i32,
                                          len: *mut i32) -> c_int;
    pub fn ls_solver_get_stats(s: *const ls_solver, ls_iterations: *mut u64, neighbours_scored: *mut u64,
                               gpu_evals: *mut u64) -> c_int;
    pub fn ls_solver_free(s: *mut ls_solver) -> c_int;
    pub fn ls_multi_solver_create_nqueens(ctx: *mut ls_ctx, board_size: u64, n_restarts: i32,
                                          seeds: *const *const c_char, params: *const ls_solver_params,
                                          out: *mut *mut ls_multi_solver) -> c_int;
    pub fn ls_multi_solver_create_sched(ctx: *mut ls_ctx, n_days_scored: i32, start_weekday: i32, n_employees: i32,
                                        holiday_bits: *const u64, n_restarts: i32, seeds: *const *const c_char,
                                        params: *const ls_solver_params, out: *mut *mut ls_multi_solver) -> c_int;
    pub fn ls_multi_solver_configure(m: *mut ls_multi_solver, n_slots: i32, mode: i32) -> c_int;
    pub fn ls_multi_solver_run(m: *mut ls_multi_solver, max_rounds_per_restart: u64, max_seconds: c_double) -> c_int;
    pub fn ls_multi_solver_get_best(m: *const ls_multi_solver, restart: i32, score: *mut i64, solution: *mut i64,
                                    cap: i32, len: *mut i32, have: *mut c_int) -> c_int;
    pub fn ls_multi_solver_get_stats(m: *const ls_multi_solver, ls_iterations: *mut u64, neighbours_scored: *mut u64,
                                     gpu_launches: *mut u64, rounds: *mut u64) -> c_int;
    pub fn ls_multi_solver_get_restart_stats(m: *const ls_multi_solver, restart: i32, ls_iterations: *mut u64,
                                             neighbours_scored: *mut u64, rounds: *mut u64) -> c_int;
    pub fn ls_multi_solver_get_counters(m: *const ls_multi_solver, out: *mut u64) -> c_int;
    pub fn ls_multi_solver_free(m: *mut ls_multi_solver) -> c_int;
    pub fn ls_host_hash_seed(seed: *const c_char, out32: *mut u8) -> c_int;
    pub fn ls_host_chacha20_block(key: *const u32, w12: u32, w13: u32, w14: u32, w15: u32, out16: *mut u32) -> c_int;
    pub fn ls_host_rng_script(seed32: *const u8, ops: *const i64, n_ops: i64, weights: *const f64, out: *mut u64,
                              cap: i64, n_out: *mut i64) -> c_int;
}
