import os, sys, json
sys.path.insert(0, "/root/repo/constraint-solver_amd"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch, ls_amd
n = 4096
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
out = {}
with ls_amd.Context(0, algo="scan") as ctx:
    st = torch.cuda.Stream(device=dev)
    ctx.set_stream(st.cuda_stream)
    for R in (1, 8):
        rows = torch.from_numpy(np.stack([rng.integers(0, n, size=n) for _ in range(R)]).astype(np.int32)).to(dev)
        cols = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(R)]).astype(np.int32)).to(dev)
        ncols = torch.full((R,), n, dtype=torch.int32, device=dev)
        cur = torch.zeros(R, dtype=torch.int64, device=dev)
        res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        for mult in (1, 5, 20, 80, 320):
            w = mult * n
            f = lambda: ctx.nqueens_eval_batch_dev(R, n, rows.data_ptr(), cur.data_ptr(), cols.data_ptr(), ncols.data_ptr(), n, w, res.data_ptr())
            with torch.cuda.stream(st):
                for _ in range(5): f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                for _ in range(100): f()
                b.record(st)
            st.synchronize()
            us = a.elapsed_time(b) * 10
            out[f"R{R}_window_{mult}n"] = round(us, 1)
    ctx.set_stream(None)
print(json.dumps(out, indent=1))
