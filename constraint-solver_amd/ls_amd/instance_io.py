"""Scheduling instances and results in the reference's only wire format: the JSON the wasm bindings
exchange with the web front-end (web/employee-scheduling-wasm-bindgen/src/lib.rs:86-110).

  EmployeeSchedulingInput  {"startDate": "YYYY-MM-DD", "endDate": "YYYY-MM-DD",
                            "employees": [{"id": 3}, ...], "employeeHolidays": [["YYYY-MM-DD", ...], ...]}
                           (holiday lists are zipped with `employees` in input order, lib.rs:21-29)
  ScoredSolutionWrapper    {"score": {"hard_score": h, "soft_score": s},
                            "days_to_employees": [["Mon 2022-05-09", {"id": 3}], ...]}   (scored days only)
"""
from __future__ import annotations

import datetime as dt
import json
from dataclasses import dataclass

import numpy as np

# create_solver's hard-coded hyper-parameters (wasm lib.rs:30-37)
WASM_PARAMS = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64,
                   all_solutions_capacity=100_000, all_solution_iteration_expiry=1_000,
                   iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)
WASM_SEED = "42"


@dataclass
class SchedulingInput:
    start: dt.date
    end: dt.date
    employee_ids: list          # ascending ids; position = the employee INDEX the C ABI uses
    holiday_bits: np.ndarray    # uint64[n_employees], bit d = holiday on day index d

    @property
    def n_days_scored(self) -> int:
        return (self.end - self.start).days + 1

    @property
    def start_weekday(self) -> int:       # chrono Weekday::num_days_from_monday
        return self.start.weekday()


def load_input(text: str) -> SchedulingInput:
    d = json.loads(text)
    start = dt.date.fromisoformat(d["startDate"])
    end = dt.date.fromisoformat(d["endDate"])
    if end < start:
        raise ValueError("endDate before startDate")
    n_days = (end - start).days + 1
    if n_days + 1 > 64:
        raise ValueError(f"{n_days} scored days (+1 stored) exceeds LS_SCHED_MAX_DAYS = 64")
    ids_in_order = [int(e["id"]) for e in d["employees"]]
    ids = sorted(set(ids_in_order))                      # BTreeSet<Employee> (lib.rs:41)
    index = {e: i for i, e in enumerate(ids)}
    bits = np.zeros(len(ids), dtype=np.uint64)
    for emp, holidays in zip(ids_in_order, d.get("employeeHolidays", [])):
        for h in holidays:
            day = (dt.date.fromisoformat(h) - start).days
            if not 0 <= day < n_days:
                raise ValueError(f"holiday {h} outside the schedule (the reference panics: employee lib.rs:275)")
            bits[index[emp]] |= np.uint64(1) << np.uint64(day)
    return SchedulingInput(start, end, ids, bits)


def create_solver(ctx, inp: SchedulingInput, seed: str = WASM_SEED, params: dict | None = None):
    """create_solver (wasm lib.rs:18-53) on the GPU path."""
    from . import Solver
    return Solver.scheduling(ctx, inp.n_days_scored, inp.start_weekday, len(inp.employee_ids), seed,
                             params or WASM_PARAMS, holiday_bits=inp.holiday_bits)


def solution_wrapper(inp: SchedulingInput, score, solution_indices) -> dict:
    """get_best_solution (wasm lib.rs:71-84): scored days only, '%a %Y-%m-%d' labels, employees by id."""
    hard, soft = score
    days = []
    for i in range(inp.n_days_scored):
        day = inp.start + dt.timedelta(days=i)
        days.append([day.strftime("%a %Y-%m-%d"), {"id": inp.employee_ids[int(solution_indices[i])]}])
    return {"score": {"hard_score": float(hard), "soft_score": float(soft)}, "days_to_employees": days}


def dumps_solution(inp: SchedulingInput, score, solution_indices) -> str:
    return json.dumps(solution_wrapper(inp, score, solution_indices))
