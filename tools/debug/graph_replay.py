import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import ls_amd, oracle as orc
rng = np.random.default_rng(21)
n, R, mc = 512, 4, 6
dev = torch.device("cuda:0")
rows_np = np.stack([rng.integers(0, n, size=n) for _ in range(R)]).astype(np.int32)
cols_np = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
cur_np = np.array([orc.nq_score(rows_np[r]) for r in range(R)], dtype=np.int64)
rows, cols = torch.from_numpy(rows_np).to(dev), torch.from_numpy(cols_np).to(dev)
cur, ncols = torch.from_numpy(cur_np).to(dev), torch.full((R,), mc, dtype=torch.int32, device=dev)
res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
ctx = ls_amd.Context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
def launch():
    ctx.nqueens_eval_batch_dev(R, n, rows.data_ptr(), cur.data_ptr(), cols.data_ptr(), ncols.data_ptr(), mc, 5 * n, res.data_ptr())
with torch.cuda.stream(stream):
    launch()
torch.cuda.synchronize()
print("direct", res.cpu().numpy().tolist())
exp = [orc.nq_neighbourhood(rows_np[r], cols_np[r], 5 * n, "table", want_scores=False) for r in range(R)]
print("expect", [(e["n_scored"], e["best_index"], e["best_score"]) for e in exp])
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=stream):
    launch()
torch.cuda.synchronize()
print("after capture (not executed)", res.cpu().numpy().tolist())
for t in range(2):
    res.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("replay", t, res.cpu().numpy().tolist())
with torch.cuda.stream(stream):
    res.zero_(); launch()
torch.cuda.synchronize()
print("direct again", res.cpu().numpy().tolist())
