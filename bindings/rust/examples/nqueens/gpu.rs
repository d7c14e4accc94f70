//! examples/nqueens/src/gpu.rs — NEW FILE added to the reference's `nqueens` crate (it needs the private
//! `NQueensSolution::rows`).  GPU-backed plug-ins for the UNCHANGED `local-search` crate: nothing in
//! `local-search/` is edited, `LocalSearch::execute` (local-search/src/local_search.rs:301-342) runs as it is.
//!
//! How lines :315-325 end up on the GPU without touching them:
//!   `iter_local_moves` does the host half of the reference proposer (examples/nqueens/src/lib.rs:182-205: conflict
//!   columns, weighted sample, `gen_range`, `choose_multiple` — the same `rand` calls in the same order, so the RNG
//!   stream is the reference's), then ONE call into libls_amd.so scores the whole neighbourhood those columns span
//!   (:214-234 order, tabu filter :319, window :321, `(score, solution)` minimum :323-325) and the iterator yields
//!   ONLY the winner.  `execute` then filters it (not tabu: it differs from `current`, and the tabu set is
//!   `{current}` for every configuration the reference ships), scores it (a cache hit below), takes it, sorts one
//!   element and picks it.
//!
//! SOURCE ONLY — not compiled here (no Rust toolchain in this repository's build environment).
use std::sync::{Arc, Mutex};

use local_search::local_search::{MoveProposer, ScoredSolution, SolutionScoreCalculator};
use local_search_amd::GpuContext;
use rand::prelude::SliceRandom;
use rand::Rng;

use crate::{NQueensScore, NQueensSolution};

type Integer = i64;

/// One solver's GPU state, shared by its move proposer and its score calculators.
pub struct GpuNQueens {
    ctx: GpuContext,
    window_size: usize,
    /// board whose column scores and total the GPU has already delivered (rows, col scores, total)
    known: Option<(Vec<Integer>, Vec<Integer>, Integer)>,
}

impl GpuNQueens {
    /// `window_size` = the value also given to `LocalSearch::new` (main.rs:130): the proposer has to know where
    /// `take(window_size)` stops because it hands back the winner of exactly that prefix.
    pub fn new(device: i32, window_size: usize) -> Arc<Mutex<Self>> {
        let ctx = GpuContext::new(device).expect("ls_ctx_create failed (no CPU fallback exists)");
        Arc::new(Mutex::new(Self { ctx, window_size, known: None }))
    }

    /// get_col_scores + sum (lib.rs:74-87,130-139), from the previous launch when the board is its winner.
    fn col_scores(&mut self, rows: &[Integer]) -> (Vec<Integer>, Integer) {
        if let Some((known_rows, cs, total)) = &self.known {
            if known_rows.as_slice() == rows {
                return (cs.clone(), *total);
            }
        }
        let (cs, total) = self.ctx.nqueens_col_scores(rows).expect("ls_nqueens_col_scores");
        self.known = Some((rows.to_vec(), cs.clone(), total));
        (cs, total)
    }
}

pub struct GpuNQueensSolutionScoreCalculator {
    shared: Arc<Mutex<GpuNQueens>>,
}

impl GpuNQueensSolutionScoreCalculator {
    pub fn new(shared: Arc<Mutex<GpuNQueens>>) -> Self {
        Self { shared }
    }
}

impl SolutionScoreCalculator for GpuNQueensSolutionScoreCalculator {
    type _Solution = NQueensSolution;
    type _Score = NQueensScore;

    fn get_scored_solution(&self, solution: Self::_Solution) -> ScoredSolution<Self::_Solution, Self::_Score> {
        let (_cs, total) = self.shared.lock().unwrap().col_scores(&solution.rows);
        ScoredSolution { score: NQueensScore(total), solution }
    }
}

pub struct GpuNQueensMoveProposer {
    shared: Arc<Mutex<GpuNQueens>>,
}

impl GpuNQueensMoveProposer {
    pub fn new(shared: Arc<Mutex<GpuNQueens>>) -> Self {
        Self { shared }
    }
}

impl MoveProposer for GpuNQueensMoveProposer {
    type R = rand_chacha::ChaCha20Rng;
    type Solution = NQueensSolution;

    fn iter_local_moves(&self, start: &Self::Solution, rng: &mut Self::R) -> Box<dyn Iterator<Item = Self::Solution>> {
        let mut gpu = self.shared.lock().unwrap();
        let (col_scores, current_score) = gpu.col_scores(&start.rows);
        // host half of NQueensMoveProposer::iter_local_moves, lib.rs:182-205 (same draws, same order)
        let mut cols_with_conflicts: Vec<(usize, Integer)> =
            col_scores.into_iter().enumerate().filter(|(_col, score)| *score != 0).collect();
        cols_with_conflicts.sort();
        if cols_with_conflicts.is_empty() {
            return Box::new(std::iter::empty()); // `random_cols == None`: the iterator is empty (lib.rs:193-194, 236-238)
        }
        let amount = (start.rows.len() / 20).clamp(1, cols_with_conflicts.len());
        let cols: Vec<usize> = cols_with_conflicts
            .choose_multiple_weighted(rng, amount, |(_col, score)| *score as f64 + 1e-4_f64)
            .unwrap()
            .map(|(col, _score)| *col)
            .collect();
        let num_cols = rng.gen_range(1..=cols.len());
        let chosen: Vec<usize> = cols.choose_multiple(rng, num_cols).copied().collect();
        // device half: MoveIterator's candidates (lib.rs:214-234) -> filter -> score -> take -> sort -> first
        let window = gpu.window_size;
        let step = gpu.ctx.nqueens_step(&start.rows, current_score, &chosen, window).expect("ls_nqueens_eval_batch_next");
        match step.best {
            None => Box::new(std::iter::empty()),
            Some(best) => {
                let mut winner = start.clone();
                winner.rows[best.column] = best.new_row; // MoveIterator, lib.rs:226-228
                gpu.known = Some((winner.rows.clone(), step.next_col_scores, best.score));
                Box::new(std::iter::once(winner))
            }
        }
    }
}
