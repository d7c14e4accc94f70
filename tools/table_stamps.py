#!/usr/bin/env python3
"""Diagnostic: phase timeline of the default TABLE kernel on the bench workload, from in-kernel wall-clock stamps of a
separate -DLS_NQ_STAMPS build of the library (tools/_build/libls_amd_stamps.so; the product build carries no stamps).
    python tools/table_stamps.py build      # here (hipcc cross-compiles)
    LS_AMD_LIB=tools/_build/libls_amd_stamps.so python tools/table_stamps.py   # on the GPU box"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libls_amd_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = [os.path.join(ROOT, "constraint-solver_amd", "csrc", f) for f in ("ls_api.hip", "nq_kernels.hip", "nq_server.hip", "sched_wave.hip", "sched_delta.hip", "sched_server.hip", "sched_grid.hip", "ls_solver.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLS_NQ_STAMPS",
                           "-Wno-unused-function", "-o", SO] + src)
    print(SO); sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import ls_amd
from ls_amd import _lib

n, R = 4096, 8
boards = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)]).astype(np.int32)
with ls_amd.Context(0, algo="table") as ctx:
    _, cur = ctx.nqueens_col_scores(boards)
    cols = np.tile(np.arange(n, dtype=np.int32), (R, 1))
    for _ in range(3):
        ctx.nqueens_eval_batch(boards, cur, cols, np.full(R, n, np.int32), n * (n - 1))
    lib = _lib.load()
    out = np.zeros(1024 * 8, dtype=np.uint64)
    lib.ls_debug_nq_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.ls_debug_nq_stamps(out.ctypes.data_as(C.c_void_p), len(out)) == 0
    st = out.reshape(1024, 8)[:512].astype(np.int64)
    t0 = st[:, 0].min()
    names = ["start", "loads issued", "LDS zeroed", "tables built", "records ready", "units done", "block reduced"]
    for p, nm in enumerate(names):
        rel = (st[:, p] - t0) / 100.0   # us
        print(f"{nm:14s}: min {rel.min():7.2f}  median {np.median(rel):7.2f}  max {rel.max():7.2f} us after the first block's start")
