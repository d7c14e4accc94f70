#!/usr/bin/env python3
"""Times the TABLE family on the bench workload (8 restarts x full 4096 x 4096 neighbourhood) and on the reference-shaped
launch for the current LS_NQ_TABLE_VARIANT; results are checked against the scan family's results of the same launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import torch
import ls_amd

n, R = 4096, 8
dev = torch.device("cuda:0")
boards = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)]).astype(np.int32)
with ls_amd.Context(0, algo="table") as ctx:
    _, cur = ctx.nqueens_col_scores(boards)
    rows_d, cur_d = torch.from_numpy(boards).to(dev), torch.from_numpy(cur).to(dev)
    res_d = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(12345)
    shapes = {"full": (torch.arange(n, dtype=torch.int32, device=dev).repeat(R, 1).contiguous(), n, n * (n - 1)),
              "ref_5n": (torch.from_numpy(np.stack([rng.permutation(n)[:204] for _ in range(R)]).astype(np.int32)).to(dev), 204, 5 * n),
              "cols512": (torch.from_numpy(np.stack([rng.permutation(n)[:512] for _ in range(R)]).astype(np.int32)).to(dev), 512, 10 ** 12)}
    for name, (cols_d, mc, window) in shapes.items():
        ncols_d = torch.full((R,), mc, dtype=torch.int32, device=dev)
        def step():
            ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_d.data_ptr(), ncols_d.data_ptr(), mc, window, res_d.data_ptr())
        with torch.cuda.stream(stream):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            iters = 100
            for _ in range(iters):
                step()
            b.record(stream)
            torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        got = res_d.cpu().numpy().copy()
        ctx.set_algo("scan")
        with torch.cuda.stream(stream):
            step()
            torch.cuda.synchronize()
        ref = res_d.cpu().numpy().copy()
        ctx.set_algo("table")
        scored = int(got[:, 0].sum())
        print(f"variant {os.environ.get('LS_NQ_TABLE_VARIANT', 'default')} {name}: {us:8.2f} us per launch, {scored / us * 1e6:.3e} neighbours/s, matches scan: {bool((got == ref).all())}", flush=True)
    ctx.set_stream(None)
