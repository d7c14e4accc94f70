#!/usr/bin/env python3
"""Many scheduling restarts on one GPU (ls_multi_solver_create_sched), timed: config 3 x R restarts (slots of the resident LS-step
launch) and config 5 x R restarts (sub-grids of the resident grid launch).  LS_AMD_TRACE=1 adds the library's own phase times
and the workgroups' timelines (stamps build).  Usage: python tools/multi_sched_trace.py [--c3 64] [--c5 8] [--repeat 3]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd  # noqa: E402

SCHED = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
             all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3", type=int, default=64)
    ap.add_argument("--c5", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--slots", type=int, default=0)
    args = ap.parse_args()
    E = 10_000
    rng = np.random.default_rng(5)
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    with ls_amd.Context(0) as ctx:
        for name, R, a, p, hb in (("config 3", args.c3, (31, 0, 7), SCHED, None), ("config 5", args.c5, (31, 0, E), dict(SCHED, window_size=65_536), bits)):
            if R <= 0:
                continue
            seeds = [str(42 + r) for r in range(R)]
            s = ls_amd.Solver.scheduling(ctx, *a, "42", p, holiday_bits=hb)
            t0 = time.perf_counter()
            while not s.is_finished():
                s.execute_round()
            one = time.perf_counter() - t0
            its = s.stats()["ls_iterations"]
            s.close()
            print(f"{name}: one restart alone {one * 1e3:.1f} ms, {one / its * 1e6:.2f} us per LS iteration", flush=True)
            for rep in range(args.repeat):
                m = ls_amd.MultiSolver.scheduling(ctx, *a, seeds, p, holiday_bits=hb, n_slots=args.slots or (None if name == "config 3" else R))
                t0 = time.perf_counter()
                m.run(max_rounds_per_restart=250)
                dt = time.perf_counter() - t0
                st = m.stats()
                m.close()
                print(f"{name} x {R}: {dt * 1e3:.1f} ms = {dt / one:.2f} x one restart, {st['neighbours_scored'] / dt:.3e} neighbours/s, "
                      f"{st['ls_iterations'] / dt:.3e} LS iterations/s, {dt / st['ls_iterations'] * R * 1e6:.2f} us per LS iteration per restart", flush=True)


if __name__ == "__main__":
    main()
