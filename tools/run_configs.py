#!/usr/bin/env python3
"""End-to-end runs of BASELINE.json's single-GPU configurations through the product solver
(host driver in C++, scores + neighbourhoods on the GPU), plus kernel-throughput lines for the
scheduling path.  Prints one JSON object.  Usage: python tools/run_configs.py [--quick]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np  # noqa: E402

import ls_amd  # noqa: E402

NQ = dict(local_search_max_iterations=10_000, best_solutions_capacity=32, all_solutions_capacity=100_000,
          all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000,
          max_allow_no_improvement_for=5)
SCHED = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64,
             all_solutions_capacity=100_000, all_solution_iteration_expiry=1_000,
             iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)


def run_solver(s, max_rounds, max_seconds):
    t0 = time.perf_counter()
    rounds = 0
    while not s.is_finished() and rounds < max_rounds and time.perf_counter() - t0 < max_seconds:
        s.execute_round()
        rounds += 1
        b = s.best()
        if b is not None and (b[0] == 0 or b[0] == (0, 0)):
            break
    dt = time.perf_counter() - t0
    st = s.stats()
    b = s.best()
    return dict(rounds=rounds, seconds=dt, best_score=None if b is None else b[0], ls_iterations=st["ls_iterations"],
                neighbours_scored=st["neighbours_scored"], neighbours_per_s=st["neighbours_scored"] / dt,
                ls_iterations_per_s=st["ls_iterations"] / dt,
                us_per_ls_iteration=dt / max(st["ls_iterations"], 1) * 1e6)


def main():
    quick = "--quick" in sys.argv
    out = {}
    with ls_amd.Context(0) as ctx:
        out["device"] = ctx.device_info()["name"]
        for algo in ("scan", "table"):
            ctx.set_algo(algo)
            # config 1 shape on the GPU path: n = 8, seed "42"
            s = ls_amd.Solver.nqueens(ctx, 8, "42", dict(NQ, window_size=40))
            out[f"nq8_seed42_{algo}"] = run_solver(s, 10_000, 30)
            s.close()
            # config 2: n = 1024, single restart, window 5n
            s = ls_amd.Solver.nqueens(ctx, 1024, "42", dict(NQ, window_size=5 * 1024))
            out[f"nq1024_seed42_{algo}"] = run_solver(s, 10_000, 10 if quick else 60)
            s.close()
        ctx.set_algo("scan")
        # config 3: the employee-scheduling example instance
        s = ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", SCHED)
        out["sched_example_seed42"] = run_solver(s, 250, 20 if quick else 120)
        s.close()

        # config 5, end to end on one GPU (one restart per GPU): 10 000 employees, window 65 536, holidays U{0..3}
        rng5 = np.random.default_rng(5)
        bits5 = np.zeros(10_000, dtype=np.uint64)
        for e in range(10_000):
            for d in rng5.choice(31, size=int(rng5.integers(0, 4)), replace=False):
                bits5[e] |= np.uint64(1) << np.uint64(d)
        s = ls_amd.Solver.scheduling(ctx, 31, 0, 10_000, "42", dict(SCHED, window_size=65_536), holiday_bits=bits5)
        out["sched10k_window65536_seed42"] = run_solver(s, 250, 10 if quick else 60)
        s.close()

        # config 5 shape, kernel throughput: 10 000 employees x 31 days, holidays h ~ U{0..3} per employee
        rng = np.random.default_rng(5)
        E, D, S = 10_000, 31, 32
        bits = np.zeros(E, dtype=np.uint64)
        for e in range(E):
            for d in rng.choice(D, size=int(rng.integers(0, 4)), replace=False):
                bits[e] |= np.uint64(1) << np.uint64(d)
        ctx.sched_instance_upload(D, S, 0, E, bits)
        state = rng.integers(0, E, size=S)
        # (a) exhaustive ScheduleMoveProposer order: 32 x 10 000 candidates (32 of them no-ops)
        moves = ctx.sched_exhaustive_moves(state)[0]                # generated on the device (lib.rs:493-559 order)
        for name, mv, window in (("exhaustive_320k", moves, 10 ** 9),):
            ctx.sched_eval(state, mv, window)                       # warm-up (allocations)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                res, _, _ = ctx.sched_eval(state, mv, window)
            dt = (time.perf_counter() - t0) / reps
            out[f"sched10k_{name}_host_api"] = dict(n_scored=res["n_scored"], ms_per_call=dt * 1e3,
                                                   neighbours_per_s=res["n_scored"] / dt,
                                                   note="host-pointer entry: includes the 3.8 MB move upload per call")
        # (b) the same neighbourhood generated on the device: only the state and the result cross PCIe
        ref = ctx.sched_eval(state, moves, 10 ** 9)[0]
        got = ctx.sched_eval_exhaustive(state, 10 ** 9)[0]
        assert got == ref, (got, ref)
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            ctx.sched_eval_exhaustive(state, 10 ** 9)
        dt = (time.perf_counter() - t0) / reps
        out["sched10k_exhaustive_320k_generated_on_device"] = dict(n_scored=got["n_scored"], us_per_call=dt * 1e6,
                                                                   neighbours_per_s=got["n_scored"] / dt)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
