#!/usr/bin/env python3
"""Full-neighbourhood scan/table throughput for arbitrary n (device-resident inputs).
Usage: python tools/nq_throughput.py n [restarts] [max_cols]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ls_amd  # noqa: E402

n = int(sys.argv[1])
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mc = int(sys.argv[3]) if len(sys.argv) > 3 else min(n, 512)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = ls_amd.Context(0)
ctx.set_stream(stream.cuda_stream)
boards = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)]).astype(np.int32)
ctx.set_stream(None)
_, cur = ctx.nqueens_col_scores(boards)
ctx.set_stream(stream.cuda_stream)
rows, curd = torch.from_numpy(boards).to(dev), torch.from_numpy(cur).to(dev)
cols = torch.arange(mc, dtype=torch.int32, device=dev).repeat(R, 1).contiguous()
ncols = torch.full((R,), mc, dtype=torch.int32, device=dev)
res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
out = {"n": n, "restarts": R, "cols": mc}
for algo in ("scan", "table"):
    ctx.set_algo(algo)
    fn = lambda: ctx.nqueens_eval_batch_dev(R, n, rows.data_ptr(), curd.data_ptr(), cols.data_ptr(), ncols.data_ptr(), mc,
                                            mc * (n - 1), res.data_ptr())
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 3 if algo == "scan" else 20
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    s = a.elapsed_time(b) / iters / 1e3
    cand = R * mc * (n - 1)
    out[algo] = {"neighbours_per_s": cand / s, "ms_per_launch": s * 1e3,
                 "pair_checks_per_s": cand / s * (n - 1)}
print(json.dumps(out))
