#!/usr/bin/env python3
"""Employee-scheduling example on the GPU path — the counterpart of the reference's
`examples/employee-scheduling` binary (src/main.rs:8-63: 7 employees, 2022-05-09 + 30 days, no
holidays, seed "42") and of the wasm worker protocol: an `EmployeeSchedulingInput` JSON file may be
given instead, and the result is printed as `ScoredSolutionWrapper` JSON
(web/employee-scheduling-wasm-bindgen/src/lib.rs:86-110).

    python examples/employee_scheduling.py [input.json] [--seed 42]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd  # noqa: E402
from ls_amd import instance_io as io  # noqa: E402

DEFAULT = {"startDate": "2022-05-09", "endDate": "2022-06-08",
           "employees": [{"id": i} for i in range(7)], "employeeHolidays": [[] for _ in range(7)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("input", nargs="?", help="EmployeeSchedulingInput JSON (default: the reference's example instance)")
    ap.add_argument("--seed", default="42")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    text = open(args.input).read() if args.input else json.dumps(DEFAULT)
    inp = io.load_input(text)
    print("employee scheduling local search example", file=sys.stderr)
    with ls_amd.Context(args.device) as ctx:
        s = io.create_solver(ctx, inp, seed=args.seed)
        t0 = time.perf_counter()
        while not s.is_finished():
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == (0, 0):
                break
        dt = time.perf_counter() - t0
        score, sol = s.best()
        st = s.stats()
        s.close()
    print(io.dumps_solution(inp, score, sol))
    print(f"[{dt:.3f} s, {st['ls_iterations']} local-search iterations, {st['neighbours_scored']} candidates scored on the GPU]",
          file=sys.stderr)


if __name__ == "__main__":
    main()
