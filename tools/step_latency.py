#!/usr/bin/env python3
"""Round-trip latency of the solver's per-iteration call (ls_nqueens_eval_batch_next, host pointers):
fused LS-step kernel (LS_NQ_STEP=2 zero-copy / 1 staged) against the multi-launch path (LS_NQ_STEP=0).
    python tools/step_latency.py [--board-size 4096]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board-size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=2000)
    args = ap.parse_args()
    import ls_amd
    n = args.board_size
    rng = np.random.default_rng(1)
    mc = max(1, n // 20)
    out = {}
    for mode in ("2", "1", "0"):
        os.environ["LS_NQ_STEP"] = mode
        with ls_amd.Context(0) as ctx:
            for R in (1, 8, 1, 8):
                rows = np.stack([rng.integers(0, n, size=n) for _ in range(R)]).astype(np.int32)
                cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
                ncols = np.full(R, mc, dtype=np.int32)
                cur = np.zeros(R, dtype=np.int64)
                for _ in range(50):
                    ctx.nqueens_eval_batch_next(rows, cur, cols, ncols, 5 * n)
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    ctx.nqueens_eval_batch_next(rows, cur, cols, ncols, 5 * n)
                us = (time.perf_counter() - t0) / args.reps * 1e6
                out.setdefault(f"step_mode_{mode}", []).append({"restarts": R, "us_per_call": round(us, 2)})
    print(json.dumps({"n": n, "window": 5 * n, "proposed_columns": mc, **out}, indent=1))


if __name__ == "__main__":
    main()
