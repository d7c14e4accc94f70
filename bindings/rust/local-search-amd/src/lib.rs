//! Safe wrapper over `libls_amd.so` (include/ls_amd.h) for the reference's plug-in seam.
//!
//! The reference's hot path is `LocalSearch::execute`'s neighbourhood pipeline
//! (local-search/src/local_search.rs:315-325).  This crate does NOT touch the `local-search` crate: the GPU is
//! reached through the traits it already has.  A plug-in crate (examples/nqueens, examples/employee-scheduling)
//! adds a `MoveProposer` whose `iter_local_moves` asks the GPU for the neighbourhood's winner and yields ONLY that
//! solution, plus a `SolutionScoreCalculator` that answers from the same call (the
//! adaptors are `bindings/rust/examples/*/gpu.rs`).  `execute` then runs unchanged:
//! `.filter(!tabu)` keeps the winner (the tabu set is `{current}`, and the winner differs from it),
//! `.map(get_scored_solution)` is a cache hit, `.take(window).collect()`, `sort()`, `first()` return it.
//!
//! SOURCE ONLY — the build environment of this repository has no Rust toolchain, so this crate has not been
//! compiled.  `tests/test_abi.py` checks that `sys.rs` names every function of the header and that this file
//! wraps every entry point the adaptors need.
pub mod sys;

use std::ffi::CStr;

pub struct GpuContext {
    raw: *mut sys::ls_ctx,
}

// One context is driven by one thread at a time (the reference's Solution / Score are Send, nothing is Sync:
// local-search/src/local_search.rs:16-23); it may move between threads.
unsafe impl Send for GpuContext {}

#[derive(Debug)]
pub struct LsError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for LsError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "ls_amd error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for LsError {}

/// What `first()` of the sorted neighbourhood would be, plus how many candidates were scored.
#[derive(Clone, Copy, Debug)]
pub struct NqBest {
    pub score: i64,
    pub column: usize,
    pub new_row: i64,
    /// position in the un-filtered candidate order (cols[index / n], index % n)
    pub index: usize,
}

#[derive(Clone, Debug)]
pub struct NqStep {
    pub n_scored: usize,
    pub best: Option<NqBest>,
    /// get_col_scores of the winner board (of the current board when nothing was scored):
    /// what `iter_local_moves` computes first on the next LS iteration (examples/nqueens/src/lib.rs:182)
    pub next_col_scores: Vec<i64>,
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum SchedMove {
    /// day <- employee index (examples/employee-scheduling/src/lib.rs:466-470)
    ChangeDay { day: usize, employee: usize },
    /// the two days exchange employees (:471-478)
    SwapDays { day1: usize, day2: usize },
}

impl SchedMove {
    fn raw(self) -> sys::ls_sched_move {
        match self {
            SchedMove::ChangeDay { day, employee } => sys::ls_sched_move { r#type: 0, a: day as i32, b: employee as i32 },
            SchedMove::SwapDays { day1, day2 } => sys::ls_sched_move { r#type: 1, a: day1 as i32, b: day2 as i32 },
        }
    }
}

#[derive(Clone, Copy, Debug)]
pub struct SchedBest {
    pub hard: i64,
    pub soft: i64,
    /// position in the move list
    pub index: usize,
}

#[derive(Clone, Copy, Debug)]
pub struct SchedNeighbourhood {
    pub n_scored: usize,
    pub best: Option<SchedBest>,
}

impl GpuContext {
    /// ls_ctx_create: one context (stream + persistent workspaces) on one GPU.  No CPU fallback exists.
    pub fn new(device: i32) -> Result<Self, LsError> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::ls_ctx_create(device, &mut raw) };
        if rc != sys::LS_OK {
            let msg = unsafe { CStr::from_ptr(sys::ls_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(LsError { code: rc, message: msg });
        }
        Ok(Self { raw })
    }

    fn check(&self, rc: i32) -> Result<(), LsError> {
        if rc == sys::LS_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::ls_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(LsError { code: rc, message: msg })
    }

    /// ls_ctx_set_nq_algo: LS_NQ_ALGO_AUTO (default), _TABLE or _SCAN.
    pub fn set_nq_algo(&mut self, algo: i32) -> Result<(), LsError> {
        let rc = unsafe { sys::ls_ctx_set_nq_algo(self.raw, algo) };
        self.check(rc)
    }

    // ------------------------------------------------------------------ N-queens

    /// ls_nqueens_col_scores: get_col_scores + sum (examples/nqueens/src/lib.rs:74-87,130-139).
    pub fn nqueens_col_scores(&mut self, rows: &[i64]) -> Result<(Vec<i64>, i64), LsError> {
        let packed: Vec<i32> = rows.iter().map(|&r| r as i32).collect();
        let mut cs = vec![0i64; rows.len()];
        let mut total = 0i64;
        let rc = unsafe {
            sys::ls_nqueens_col_scores(self.raw, 1, rows.len() as i32, packed.as_ptr(), cs.as_mut_ptr(), &mut total)
        };
        self.check(rc)?;
        Ok((cs, total))
    }

    /// ls_nqueens_eval_batch_next for one board: lines :315-325 of LocalSearch::execute --
    /// iter_local_moves(cols order).filter(!tabu).map(score).take(window).sort().first() -- in ONE GPU launch,
    /// together with the winner board's column scores.
    pub fn nqueens_step(&mut self, rows: &[i64], cur_score: i64, cols: &[usize], window: usize) -> Result<NqStep, LsError> {
        let n = rows.len();
        let packed: Vec<i32> = rows.iter().map(|&r| r as i32).collect();
        let cols32: Vec<i32> = cols.iter().map(|&c| c as i32).collect();
        let n_cols = cols32.len() as i32;
        let mut out = sys::ls_nq_result::default();
        let mut next = vec![0i64; n];
        let rc = unsafe {
            sys::ls_nqueens_eval_batch_next(self.raw, 1, n as i32, packed.as_ptr(), &cur_score, cols32.as_ptr(), &n_cols,
                                            n_cols, window as i64, &mut out, next.as_mut_ptr())
        };
        self.check(rc)?;
        let best = (out.best_index >= 0).then(|| NqBest {
            score: out.best_score,
            column: out.best_col as usize,
            new_row: out.best_value as i64,
            index: out.best_index as usize,
        });
        Ok(NqStep { n_scored: out.n_scored as usize, best, next_col_scores: next })
    }

    /// ls_nqueens_eval: the neighbourhood alone (no column scores of the winner).
    pub fn nqueens_best_of_neighbourhood(&mut self, rows: &[i64], cur_score: i64, cols: &[usize], window: usize)
        -> Result<(usize, Option<NqBest>), LsError> {
        let packed: Vec<i32> = rows.iter().map(|&r| r as i32).collect();
        let cols32: Vec<i32> = cols.iter().map(|&c| c as i32).collect();
        let mut out = sys::ls_nq_result::default();
        let rc = unsafe {
            sys::ls_nqueens_eval(self.raw, rows.len() as i32, packed.as_ptr(), cur_score, cols32.as_ptr(),
                                 cols32.len() as i32, window as i64, &mut out, std::ptr::null_mut())
        };
        self.check(rc)?;
        let best = (out.best_index >= 0).then(|| NqBest {
            score: out.best_score,
            column: out.best_col as usize,
            new_row: out.best_value as i64,
            index: out.best_index as usize,
        });
        Ok((out.n_scored as usize, best))
    }

    // --------------------------------------------------------- employee scheduling

    /// ls_sched_instance_upload: what ScheduleSolutionScoreCalculator::new and the solution's date range hold
    /// (examples/employee-scheduling/src/lib.rs:251-259,129-146).  `holiday_bits[e]` bit d <=> employee index e
    /// (position in the ascending `employees` vector) has a holiday on scored day d; start_weekday: Monday = 0.
    pub fn sched_instance_upload(&mut self, n_days_scored: usize, n_days_stored: usize, start_weekday: u32,
                                 holiday_bits: &[u64]) -> Result<(), LsError> {
        let rc = unsafe {
            sys::ls_sched_instance_upload(self.raw, n_days_scored as i32, n_days_stored as i32, start_weekday as i32,
                                          holiday_bits.len() as i32, holiday_bits.as_ptr())
        };
        self.check(rc)
    }

    /// ls_sched_score: get_scored_solution (lib.rs:261-375) -> (hard, soft).
    pub fn sched_score(&mut self, state: &[i32]) -> Result<(i64, i64), LsError> {
        let (mut hard, mut soft) = (0i64, 0i64);
        let rc = unsafe { sys::ls_sched_score(self.raw, 1, state.as_ptr(), &mut hard, &mut soft, std::ptr::null_mut()) };
        self.check(rc)?;
        Ok((hard, soft))
    }

    fn sched_out(out: sys::ls_sched_result) -> SchedNeighbourhood {
        let best = (out.best_index >= 0).then(|| SchedBest { hard: out.best_hard, soft: out.best_soft, index: out.best_index as usize });
        SchedNeighbourhood { n_scored: out.n_scored as usize, best }
    }

    /// ls_sched_eval: lines :315-325 over a caller-supplied move list (the proposer's stream, in order).
    pub fn sched_eval(&mut self, state: &[i32], moves: &[SchedMove], window: usize) -> Result<SchedNeighbourhood, LsError> {
        let raw: Vec<sys::ls_sched_move> = moves.iter().map(|m| m.raw()).collect();
        let mut out = sys::ls_sched_result::default();
        let rc = unsafe {
            sys::ls_sched_eval(self.raw, state.as_ptr(), raw.as_ptr(), raw.len() as i64, window as i64, &mut out,
                               std::ptr::null_mut(), std::ptr::null_mut())
        };
        self.check(rc)?;
        Ok(Self::sched_out(out))
    }

    /// ls_sched_moves_upload: keep the (constant) random move stream on the device (lib.rs:488: the iterator owns a
    /// clone of the generator, so every LS iteration replays the same stream).
    pub fn sched_moves_upload(&mut self, moves: &[SchedMove]) -> Result<(), LsError> {
        let raw: Vec<sys::ls_sched_move> = moves.iter().map(|m| m.raw()).collect();
        let rc = unsafe { sys::ls_sched_moves_upload(self.raw, raw.as_ptr(), raw.len() as i64) };
        self.check(rc)
    }

    /// ls_sched_eval_resident: like sched_eval over the first n_moves entries of the resident list.
    pub fn sched_eval_resident(&mut self, state: &[i32], n_moves: usize, window: usize) -> Result<SchedNeighbourhood, LsError> {
        let mut out = sys::ls_sched_result::default();
        let rc = unsafe { sys::ls_sched_eval_resident(self.raw, state.as_ptr(), n_moves as i64, window as i64, &mut out) };
        self.check(rc)?;
        Ok(Self::sched_out(out))
    }

    /// ls_sched_eval_exhaustive: the exhaustive ScheduleMoveProposer's neighbourhood (lib.rs:493-559), generated and
    /// scored on the device; best.index = day * n_employees + i with employee (state[day] + 1 + i) mod n_employees.
    pub fn sched_eval_exhaustive(&mut self, state: &[i32], window: usize) -> Result<SchedNeighbourhood, LsError> {
        let mut out = sys::ls_sched_result::default();
        let rc = unsafe { sys::ls_sched_eval_exhaustive(self.raw, 1, state.as_ptr(), window as i64, &mut out) };
        self.check(rc)?;
        Ok(Self::sched_out(out))
    }
}

/// Many independent scheduling restarts on one GPU (ls_multi_solver_create_sched): every restart is the unchanged get_ils
/// driver (examples/employee-scheduling/src/lib.rs:57-117) under its own seed (convention of examples/nqueens/src/main.rs:167);
/// the restarts are the slots of ONE resident launch.  `best(r)` = ((hard, soft), employee indices of the stored days).
pub struct MultiSchedSolver {
    raw: *mut sys::ls_multi_solver,
    n_stored: usize,
}

impl MultiSchedSolver {
    pub fn new(ctx: &mut GpuContext, n_days_scored: usize, start_weekday: u32, holiday_bits: &[u64], seeds: &[&str],
               params: &sys::ls_solver_params) -> Result<Self, LsError> {
        let owned: Vec<std::ffi::CString> = seeds.iter().map(|s| std::ffi::CString::new(*s).expect("seed without NUL")).collect();
        let ptrs: Vec<*const std::os::raw::c_char> = owned.iter().map(|s| s.as_ptr()).collect();
        let mut raw = std::ptr::null_mut();
        let rc = unsafe {
            sys::ls_multi_solver_create_sched(ctx.raw, n_days_scored as i32, start_weekday as i32, holiday_bits.len() as i32,
                                              holiday_bits.as_ptr(), ptrs.len() as i32, ptrs.as_ptr(), params, &mut raw)
        };
        ctx.check(rc)?;
        let mut m = MultiSchedSolver { raw, n_stored: n_days_scored + 1 };
        m.configure(ctx, 0)?; // the default slot count; makes what the slots share now rather than in the first run
        Ok(m)
    }

    /// Restarts in flight at once (0: the library's default, eight per usable core, at most what one resident launch holds).
    /// Also builds the slots' shared launch object and contexts (set-up, not part of a run).
    pub fn configure(&mut self, ctx: &mut GpuContext, n_slots: usize) -> Result<(), LsError> {
        let rc = unsafe { sys::ls_multi_solver_configure(self.raw, n_slots as i32, 1 /* LS_MULTI_FREE_RUNNING */) };
        ctx.check(rc)
    }

    pub fn run(&mut self, max_rounds_per_restart: u64, max_seconds: f64) -> Result<(), LsError> {
        let rc = unsafe { sys::ls_multi_solver_run(self.raw, max_rounds_per_restart, max_seconds) };
        if rc == 0 { Ok(()) } else { Err(LsError { code: rc, message: String::from("ls_multi_solver_run failed") }) }
    }

    pub fn best(&self, restart: usize) -> Option<((i64, i64), Vec<i64>)> {
        let (mut score, mut sol, mut len, mut have) = ([0i64; 2], vec![0i64; self.n_stored], 0i32, 0);
        let rc = unsafe {
            sys::ls_multi_solver_get_best(self.raw, restart as i32, score.as_mut_ptr(), sol.as_mut_ptr(), self.n_stored as i32, &mut len, &mut have)
        };
        if rc != 0 || have == 0 { return None; }
        sol.truncate(len as usize);
        Some(((score[0], score[1]), sol))
    }
}

impl Drop for MultiSchedSolver {
    fn drop(&mut self) {
        unsafe {
            sys::ls_multi_solver_free(self.raw);
        }
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe {
            sys::ls_ctx_destroy(self.raw);
        }
    }
}
