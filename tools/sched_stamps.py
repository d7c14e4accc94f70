#!/usr/bin/env python3
"""Diagnostic: phase timelines of the scheduling delta kernels from in-kernel wall-clock stamps of the -DLS_NQ_STAMPS build
(python tools/table_stamps.py build):
    LS_AMD_LIB=tools/_build/libls_amd_stamps.so python tools/sched_stamps.py
(a) the fused one-block launch of the reference example (512 resident moves, window 100);
(b) sched_eval_all_kernel on config 5's window-65 536 random stream and on the 320 000-candidate exhaustive neighbourhood."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import ls_amd
from ls_amd import _lib


def stamps(lib, n_blocks):
    out = np.zeros(1024 * 16, dtype=np.uint64)
    assert lib.ls_debug_sched_stamps(out.ctypes.data_as(C.c_void_p), len(out)) == 0
    return out.reshape(1024, 16)[:n_blocks].astype(np.int64)


def random_moves(rng, n, D, E):
    mv = np.zeros((n, 3), dtype=np.int32)
    for i in range(n):
        if rng.integers(0, 5) == 0:
            mv[i] = (0, rng.integers(0, D), rng.integers(0, E))
        else:
            a, b = rng.choice(D, size=2, replace=False)
            mv[i] = (1, a, b)
    return mv


rng = np.random.default_rng(7)
with ls_amd.Context(0) as ctx:
    lib = _lib.load()
    lib.ls_debug_sched_stamps.argtypes = [C.c_void_p, C.c_int]
    # (a) fused launch
    D, S, E = 31, 32, 7
    ctx.sched_instance_upload(D, S, 0, E)
    ctx.sched_moves_upload(random_moves(rng, 512, D, E))
    acc = []
    for it in range(300):
        state = rng.integers(0, E, size=S)
        ctx.sched_eval_resident(state, 512, 100)
        if it >= 50:
            st = stamps(lib, 1)[0]
            acc.append((st - st[0]) / 100.0)
    acc = np.median(np.array(acc), axis=0)
    names = ["entry", "state + move loaded", "edit, ballot, holiday gathers issued", "base table built (wavefront 0)",
             "block synchronised", "candidates scored", "block minimum", "result + completion word published"]
    print("fused one-block launch, 512 moves, window 100 (us since entry, thread 0):")
    for nm, t in zip(names, acc):
        print(f"  {nm:44s} {t:6.2f}")
    print("  inside the table build (stamps 8, 11):", " ".join(f"{acc[i]:.2f}" for i in (8, 11)))
    # (b) eval_all
    D, S, E = 31, 32, 10_000
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(D, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(d)
    ctx.sched_instance_upload(D, S, 0, E, bits)
    state = rng.integers(0, E, size=S)
    names = ["entry", "state + move loaded, edit known", "holiday gathers issued, block synchronised", "base table built (wavefront 0)",
             "block synchronised", "candidates scored", "chunk record written"]
    for label, nb, call in (("window 65 536 random stream (129 blocks)", 129, "resident"), ("exhaustive 320 000 (625 blocks)", 625, "gen")):
        if call == "resident":
            ctx.sched_moves_upload(random_moves(rng, 66_000, D, E))
        for it in range(5):
            if call == "resident":
                ctx.sched_eval_resident(state, 66_000, 65_536)
            else:
                ctx.sched_eval_exhaustive(state, 10 ** 9)
        st = stamps(lib, nb)
        t0 = st[:, 0].min()
        rel = (st - st[:, :1]) / 100.0
        print(f"sched_eval_all_kernel, {label}: block start spread {(st[:, 0].max() - t0) / 100.0:.2f} us, "
              f"last block end {(st[:, 6].max() - t0) / 100.0:.2f} us; median per-block phase times (us since block entry):")
        for i, nm in enumerate(names):
            print(f"  {nm:44s} {np.median(rel[:, i]):6.2f}")
        print("  inside the table build (stamps 8, 11):", " ".join(f"{np.median(rel[:, i]):.2f}" for i in (8, 11)))
        print("  inside the record phase (stamps 12..14: wave count, block minimum, block count):", " ".join(f"{np.median(rel[:, i]):.2f}" for i in (12, 13, 14)))
