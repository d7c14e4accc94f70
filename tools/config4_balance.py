#!/usr/bin/env python3
"""Diagnostic: LS iterations of each of config 4's 64 restarts (n = 4096, seeds "42".."105"), run one at a time, and
what they imply for the 16-slot end-to-end run: the longest restart alone (critical path) against the sum over 16 slots."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd

n = 4096
params = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32, all_solutions_capacity=100_000,
              all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
its, secs = [], []
with ls_amd.Context(0) as ctx:
    for i in range(64):
        s = ls_amd.Solver.nqueens(ctx, n, str(42 + i), params)
        t0 = time.perf_counter()
        while not s.is_finished():
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        secs.append(time.perf_counter() - t0)
        its.append(s.stats()["ls_iterations"])
        s.close()
us = [1e6 * a / b for a, b in zip(secs, its)]
print(json.dumps(dict(iterations=its, total=sum(its), longest=max(its), mean=sum(its) / 64,
                      us_per_iteration_alone=sum(secs) / sum(its) * 1e6,
                      seconds_longest_alone=max(secs), seconds_sum_over_16=sum(secs) / 16)))
