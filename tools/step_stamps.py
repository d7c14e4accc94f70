#!/usr/bin/env python3
"""Diagnostic: phase timeline of the fused LS-step kernel (n = 4096, window 5n, one restart, board resident) from in-kernel
wall-clock stamps of the -DLS_NQ_STAMPS build (python tools/table_stamps.py build).
    LS_AMD_LIB=tools/_build/libls_amd_stamps.so python tools/step_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import ls_amd
from ls_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(3)
rows = rng.integers(0, n, size=(1, n)).astype(np.int32)
with ls_amd.Context(0) as ctx:
    _, cur = ctx.nqueens_col_scores(rows)
    lib = _lib.load()
    lib.ls_debug_nq_stamps.argtypes = [C.c_void_p, C.c_int]
    acc = []
    for it in range(200):
        cols = rng.permutation(n)[: max(1, n // 20)].astype(np.int32).reshape(1, -1)
        out, nxt = ctx.nqueens_eval_batch_next(rows, cur, cols, [cols.shape[1]], 5 * n)
        if out[0]["best_index"] >= 0:
            rows[0, out[0]["best_col"]] = out[0]["best_value"]          # continue from the winner: the board stays resident
            cur = np.array([out[0]["best_score"]])
        st = np.zeros(8, dtype=np.uint64)
        assert lib.ls_debug_nq_stamps(st.ctypes.data_as(C.c_void_p), 8) == 0
        if it >= 20:
            acc.append((st.astype(np.int64) - int(st[0])) / 100.0)
    acc = np.median(np.array(acc), axis=0)
    names = ["entry, board loads issued", "inputs in registers / LDS", "tables built", "slot records", "candidates scored",
             "move applied, result written", "column scores written", "completion word published"]
    for nm, t in zip(names, acc):
        print(f"{nm:32s} {t:6.2f} us")
