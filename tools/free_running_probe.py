#!/usr/bin/env python3
"""Diagnostic: ls_multi_solver in free-running mode with many slots, repeated; reports the first inconsistency."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd

n, rounds = 48, 6
seeds = [str(s) for s in range(42, 42 + 13)]
p = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32, all_solutions_capacity=100_000,
         all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
with ls_amd.Context(0) as ctx:
    singles = []
    for seed in seeds:
        s = ls_amd.Solver.nqueens(ctx, n, seed, p)
        for _ in range(rounds):
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        singles.append(s.best())
        s.close()
    for mode, slots in (("free", 16), ("free", 13), ("free", 8), ("lockstep", 13), ("free", 16), ("free", 16)):
        for rep in range(5):
            m = ls_amd.MultiSolver(ctx, n, seeds, p, n_slots=slots, mode=mode)
            try:
                m.run(max_rounds_per_restart=rounds)
                got = [m.best(r) for r in range(len(seeds))]
                bad = [r for r in range(len(seeds)) if got[r] != singles[r]]
                print(mode, slots, rep, "mismatching restarts:", bad, [got[r][0] for r in bad], flush=True)
            except ls_amd.LsError as e:
                print(mode, slots, rep, "ERROR", e, flush=True)
            m.close()
