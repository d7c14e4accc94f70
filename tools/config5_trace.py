#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5 per GPU end to end (10 000 employees x 31 days, holidays U{0..3}, window 65 536, seed "42",
250 rounds) on the resident scheduling grid (default) and with one launch pair per LS iteration (LS_SCHED_SERVER=0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import ls_amd

sched = dict(local_search_max_iterations=1_000, window_size=65_536, best_solutions_capacity=64, all_solutions_capacity=100_000,
             all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)
rng = np.random.default_rng(5)
bits = np.zeros(10_000, dtype=np.uint64)
for e in range(10_000):
    for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
        bits[e] |= np.uint64(1) << np.uint64(d)
res = {}
for mode in ("1", "0", "1"):
    os.environ["LS_SCHED_SERVER"] = mode
    with ls_amd.Context(0) as c:
        s = ls_amd.Solver.scheduling(c, 31, 0, 10_000, "42", sched, holiday_bits=bits)
        t0 = time.perf_counter()
        while not s.is_finished():
            s.execute_round()
        dt = time.perf_counter() - t0
        st = s.stats()
        print("resident grid" if mode == "1" else "launch per call", "config 5:", s.best()[0], st["ls_iterations"], "LS iterations", round(dt * 1e3, 1), "ms",
              round(dt / st["ls_iterations"] * 1e6, 2), "us per LS iteration", round(st["neighbours_scored"] / dt / 1e9, 2), "e9 candidates/s", c.counters(), flush=True)
        res[mode] = s.best()
        s.close()
print("same result:", res["1"] == res["0"])
