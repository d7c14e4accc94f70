#!/usr/bin/env python3
"""Diagnostic: the default TABLE kernel (nq_table_pairs_kernel) on the bench workload shape (n = 4096, full n x n
neighbourhood per restart) with 8 / 16 / 32 / 64 restarts per launch, inputs resident in HBM, HIP events around back-to-back
launches.  At 8 restarts the launch is granularity-bound (512 workgroups start over ~20 us, each rebuilds its restart's
tables); more restarts per launch give every workgroup more columns per table build."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import numpy as np
import torch
import ls_amd

n = 4096
dev = torch.device("cuda", 0)
kc = json.load(open(os.path.join(ROOT, "profiles", "kernel_counters.json"))).get("nq_table_pairs_kernel", {})
lds_per_cand = kc.get("lds_wave_instr_per_candidate")
with ls_amd.Context(0, algo="table") as ctx:
    info = ctx.device_info()
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    for R in (8, 16, 32, 64):
        boards = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)]).astype(np.int32)
        _, cur = ctx.nqueens_col_scores(boards)
        rows_d, cur_d = torch.from_numpy(boards).to(dev), torch.from_numpy(cur).to(dev)
        cols_d = torch.arange(n, dtype=torch.int32, device=dev).repeat(R, 1).contiguous()
        nc_d = torch.full((R,), n, dtype=torch.int32, device=dev)
        res_d = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        step = lambda: ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_d.data_ptr(), nc_d.data_ptr(), n, n * (n - 1), res_d.data_ptr())
        with torch.cuda.stream(stream):
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(50):
                step()
            b.record(stream)
            torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        cand = R * n * (n - 1)
        out = dict(restarts=R, us_per_launch_incl_finalize=us, neighbours_per_s=cand / (us * 1e-6))
        if lds_per_cand:
            # LDS-array bound at the spec clock: one wave-instruction per 2 cycles... (tools/summarise_counters.py convention: 64 lanes x 4 B per cycle per CU)
            bound_us = kc["lds_array_bound_us"] * R / 8
            out.update(lds_array_bound_us=bound_us, frac=bound_us / us)
        print(json.dumps(out), flush=True)
    ctx.set_stream(None)
