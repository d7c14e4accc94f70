#!/usr/bin/env python3
"""Diagnostic: one config-4 restart alone (n = 4096, seed given) with LS_AMD_TRACE=1: where an LS iteration's time goes
(host proposer / API call: host before, answer wait, workgroup service time, host after / rest of the driver)."""
import os, sys, time
os.environ["LS_AMD_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd

n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
seed = sys.argv[1] if len(sys.argv) > 1 else "99"
params = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32, all_solutions_capacity=100_000,
              all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
with ls_amd.Context(0) as ctx:
    s = ls_amd.Solver.nqueens(ctx, n, seed, params)
    t0 = time.perf_counter()
    while not s.is_finished():
        s.execute_round()
        b = s.best()
        if b is not None and b[0] == 0:
            break
    dt = time.perf_counter() - t0
    it = s.stats()["ls_iterations"]
    print(f"seed {seed} n {n}: {it} LS iterations in {dt:.3f} s = {dt / it * 1e6:.2f} us per iteration", flush=True)
    s.close()
