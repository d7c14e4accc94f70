#!/usr/bin/env python3
"""Diagnostic: ls_sched_eval_exhaustive for one state of the config-5 instance (10 000 employees x 32 stored days = 320 000
candidates) on the resident grid (sched_grid_server_kernel<GEN>) against the launched pipeline (LS_SCHED_SERVER=0): per-call time
on the bare C entry, and with LS_AMD_TRACE=1 (+ LS_AMD_LIB=tools/_build/libls_amd_stamps.so) the finisher's timeline.
    LS_AMD_TRACE=1 python tools/exhaustive_grid_trace.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np  # noqa: E402


def run(server):
    os.environ["LS_SCHED_SERVER"] = server
    import ls_amd
    E, D, S = 10_000, 31, 32
    rng = np.random.default_rng(5)
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(D, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    out = []
    with ls_amd.Context(0) as c:
        c.sched_instance_upload(D, S, 0, E, bits)
        state = np.ascontiguousarray(rng.integers(0, E, size=S), dtype=np.int32)
        res = ls_amd.SchedResult()
        fn, h = c._l.ls_sched_eval_exhaustive, c._h
        args = (h, 1, state.ctypes.data_as(C.c_void_p), 10 ** 9, C.byref(res))
        for window in (10 ** 9, 200_000):
            args = (h, 1, state.ctypes.data_as(C.c_void_p), window, C.byref(res))
            for _ in range(300):
                assert fn(*args) == 0
            t0 = time.perf_counter()
            for _ in range(3000):
                fn(*args)
            us = (time.perf_counter() - t0) / 3000 * 1e6
            out.append((window, res.n_scored, res.best_index, res.best_hard, res.best_soft))
            print(f"LS_SCHED_SERVER={server} window {window}: {us:.2f} us per call (bare C entry), {res.n_scored / us / 1e3:.2f} e9 candidates/s", flush=True)
    return out


a = run("1")
b = run("0")
print("same result:", a == b)
