//! examples/employee-scheduling/src/gpu.rs — NEW FILE added to the reference's `employee-scheduling` crate.
//! GPU-backed plug-ins for the UNCHANGED `local-search` crate (see bindings/rust/examples/nqueens/gpu.rs for the
//! mechanism: the proposer yields only the neighbourhood's winner).
//!
//! ScheduleRandomMoveProposer (lib.rs:428-491) is an INFINITE stream of random moves drawn from `rng.clone()`
//! (:488): nothing ever advances LocalSearch's generator, so every LS iteration replays the same stream and only
//! the base solution (and with it the tabu filter) changes.  The adaptor therefore draws the stream once — with the
//! same `rand` calls in the same order as MoveIterator::next (:459-479) — keeps it on the device, and asks the GPU
//! for the first `window_size` non-tabu candidates' winner by (hard, soft, date_to_employee).
//!
//! SOURCE ONLY — not compiled here (no Rust toolchain in this repository's build environment).
use std::collections::{HashMap, HashSet};
use std::sync::{Arc, Mutex};

use chrono::Datelike;
use local_search::local_search::{MoveProposer, ScoredSolution, SolutionScoreCalculator};
use local_search_amd::{GpuContext, SchedMove};
use ordered_float::OrderedFloat;
use rand::prelude::SliceRandom;

use crate::ScheduleRandomMove::{ChangeDay, SwapDays};
use crate::{Employee, Holiday, ScheduleRandomMove, ScheduleScore, ScheduleSolution};

pub struct GpuSchedule {
    ctx: GpuContext,
    window_size: usize,
    employee_to_holidays: HashMap<Employee, HashSet<Holiday>>,
    employee_index: HashMap<Employee, i32>, // position in the ascending `employees` vector (lib.rs:83)
    instance_uploaded: bool,
    /// the proposer's stream as drawn so far, the generator it continues from and the state it was cloned from
    stream: Vec<SchedMove>,
    stream_rng: Option<(rand_chacha::ChaCha20Rng, rand_chacha::ChaCha20Rng)>, // (clone source, running generator)
    resident: usize,
    known: Option<(Vec<Employee>, (i64, i64))>,
}

impl GpuSchedule {
    pub fn new(device: i32, window_size: usize, employee_to_holidays: HashMap<Employee, HashSet<Holiday>>) -> Arc<Mutex<Self>> {
        let ctx = GpuContext::new(device).expect("ls_ctx_create failed (no CPU fallback exists)");
        Arc::new(Mutex::new(Self {
            ctx,
            window_size,
            employee_to_holidays,
            employee_index: HashMap::new(),
            instance_uploaded: false,
            stream: Vec::new(),
            stream_rng: None,
            resident: 0,
            known: None,
        }))
    }

    /// The instance is what the calculator and the solution's date range hold (lib.rs:251-259, 129-146); it is read
    /// off the first solution that comes by.
    fn ensure_instance(&mut self, solution: &ScheduleSolution) {
        if self.instance_uploaded {
            return;
        }
        let days = solution.get_days_to_employees(); // scored days (lib.rs:181-192); one fewer than stored (:407-411)
        let first = days[0].0;
        self.employee_index = solution.employees.iter().enumerate().map(|(i, e)| (*e, i as i32)).collect();
        let mut bits = vec![0u64; solution.employees.len()];
        for (employee, holidays) in &self.employee_to_holidays {
            for holiday in holidays {
                let d = holiday.0.signed_duration_since(first).num_days();
                // a holiday outside the range makes the reference panic (lib.rs:275 unwrap)
                assert!(d >= 0 && (d as usize) < days.len(), "holiday outside the schedule");
                bits[self.employee_index[employee] as usize] |= 1u64 << d;
            }
        }
        self.ctx
            .sched_instance_upload(days.len(), solution.date_to_employee.len(), first.weekday().num_days_from_monday(), &bits)
            .expect("ls_sched_instance_upload");
        self.instance_uploaded = true;
    }

    fn state(&self, solution: &ScheduleSolution) -> Vec<i32> {
        solution.date_to_employee.iter().map(|e| self.employee_index[e]).collect()
    }

    fn score(&mut self, solution: &ScheduleSolution) -> (i64, i64) {
        self.ensure_instance(solution);
        if let Some((known, score)) = &self.known {
            if *known == solution.date_to_employee {
                return *score;
            }
        }
        let state = self.state(solution);
        let score = self.ctx.sched_score(&state).expect("ls_sched_score");
        self.known = Some((solution.date_to_employee.clone(), score));
        score
    }

    /// MoveIterator::next (lib.rs:459-479) as a move description: same draws, same order.
    fn extend_stream(&mut self, want: usize, n_days: usize, n_employees: usize, move_types: &[(ScheduleRandomMove, u64)]) {
        let days: Vec<usize> = (0..n_days).collect();
        let employees: Vec<usize> = (0..n_employees).collect();
        let rng = &mut self.stream_rng.as_mut().unwrap().1;
        while self.stream.len() < want {
            let current_move = move_types.choose_weighted(rng, |s| s.1).unwrap().0;
            let mv = match current_move {
                ChangeDay => {
                    let day = *days.choose(rng).unwrap();
                    let employee = *employees.choose(rng).unwrap();
                    SchedMove::ChangeDay { day, employee }
                }
                SwapDays => {
                    let xs: Vec<&usize> = days.choose_multiple(rng, 2).collect();
                    SchedMove::SwapDays { day1: *xs[0], day2: *xs[1] }
                }
            };
            self.stream.push(mv);
        }
    }
}

pub struct GpuScheduleSolutionScoreCalculator {
    shared: Arc<Mutex<GpuSchedule>>,
}

impl GpuScheduleSolutionScoreCalculator {
    pub fn new(shared: Arc<Mutex<GpuSchedule>>) -> Self {
        Self { shared }
    }
}

impl SolutionScoreCalculator for GpuScheduleSolutionScoreCalculator {
    type _Solution = ScheduleSolution;
    type _Score = ScheduleScore;

    fn get_scored_solution(&self, solution: Self::_Solution) -> ScoredSolution<Self::_Solution, Self::_Score> {
        let (hard, soft) = self.shared.lock().unwrap().score(&solution);
        ScoredSolution {
            score: ScheduleScore { hard_score: OrderedFloat(hard as f64), soft_score: OrderedFloat(soft as f64) },
            solution,
        }
    }
}

pub struct GpuScheduleMoveProposer {
    shared: Arc<Mutex<GpuSchedule>>,
    random_move_types: Vec<(ScheduleRandomMove, u64)>,
}

impl GpuScheduleMoveProposer {
    pub fn new(shared: Arc<Mutex<GpuSchedule>>) -> Self {
        Self { shared, random_move_types: vec![(ChangeDay, 1), (SwapDays, 4)] } // ScheduleRandomMoveProposer::default, lib.rs:433-438
    }
}

impl MoveProposer for GpuScheduleMoveProposer {
    type R = rand_chacha::ChaCha20Rng;
    type Solution = ScheduleSolution;

    fn iter_local_moves(&self, start: &Self::Solution, rng: &mut Self::R) -> Box<dyn Iterator<Item = Self::Solution>> {
        let mut gpu = self.shared.lock().unwrap();
        gpu.ensure_instance(start);
        // the iterator owns rng.clone() (lib.rs:488); redraw the stream only if the caller's generator ever moved
        let same = matches!(&gpu.stream_rng, Some((source, _)) if source == &*rng);
        if !same {
            gpu.stream_rng = Some((rng.clone(), rng.clone()));
            gpu.stream.clear();
            gpu.resident = 0;
        }
        let n_days = start.get_days_to_employees().len();
        let n_employees = start.employees.len();
        let window = gpu.window_size;
        if window == 0 {
            return Box::new(std::iter::empty());
        }
        let state = gpu.state(start);
        let mut want = window + window / 4 + 16;
        loop {
            if gpu.resident < want {
                gpu.extend_stream(want + want / 2, n_days, n_employees, &self.random_move_types);
                let stream = gpu.stream.clone();
                gpu.ctx.sched_moves_upload(&stream).expect("ls_sched_moves_upload");
                gpu.resident = stream.len();
            }
            let out = gpu.ctx.sched_eval_resident(&state, want, window).expect("ls_sched_eval_resident");
            if out.n_scored == window {
                let best = out.best.expect("a scored neighbourhood has a winner");
                let mut winner = start.clone();
                match gpu.stream[best.index] {
                    SchedMove::ChangeDay { day, employee } => winner.date_to_employee[day] = start.employees[employee],
                    SchedMove::SwapDays { day1, day2 } => {
                        winner.date_to_employee[day1] = start.date_to_employee[day2];
                        winner.date_to_employee[day2] = start.date_to_employee[day1];
                    }
                }
                gpu.known = Some((winner.date_to_employee.clone(), (best.hard, best.soft)));
                return Box::new(std::iter::once(winner));
            }
            // fewer than `window` non-tabu moves in the prefix: look further down the (infinite) stream
            assert!(want <= 64 * window + (1 << 16), "the random move stream cannot fill the window (the reference would not terminate)");
            want *= 2;
        }
    }
}
