#!/usr/bin/env python3
"""Scheduling kernels alone (device-resident inputs), for profiling: config-5 shaped launches
(10 000 employees, 31/32 days, holidays U{0..3}): exhaustive 320k-candidate neighbourhood and the
random stream with window 65 536.  Usage: python tools/sched_throughput.py [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import ls_amd  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = ls_amd.Context(0)
ctx.set_stream(stream.cuda_stream)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(min(n, iters)):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / min(n, iters) / 1e3


print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} for k, v in bench.sched_measurements(ctx, torch, dev, timed, lambda *a: None).items()}))
ctx.set_stream(None)
ctx.close()
