#!/usr/bin/env python3
"""BASELINE.json config 4, end to end: n-queens, 8 independent restarts per GPU in lockstep (seeds
"42"+global restart id, the reference's consecutive-seed convention, examples/nqueens/src/main.rs:167),
restart r on rank r mod N, host-side gather of the per-restart bests and a global minimum in
ScoredSolution order.  One process per GPU:

    python tools/run_config4.py [--board-size 4096] [--restarts-per-gpu 8] [--seconds 60]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/run_config4.py ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board-size", "--n", dest="n", type=int, default=4096)
    ap.add_argument("--restarts-per-gpu", type=int, default=8, help="seeds per GPU (64 on one GPU = the whole of config 4)")
    ap.add_argument("--slots", type=int, default=0, help="restarts in flight at once (0: library default, four per usable core); further seeds queue behind them")
    ap.add_argument("--mode", default="free", choices=["lockstep", "free"])
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--rounds", type=int, default=10_000)
    ap.add_argument("--algo", default="auto", choices=["auto", "table", "scan"])
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on real multi-GPU nodes; gloo to rehearse ranks on one GPU")
    ap.add_argument("--all-on-device", type=int, default=None, help="rehearsal only: every rank uses this device")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    import ls_amd
    from ls_amd import sharding

    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    ids = sharding.shard_restarts(args.restarts_per_gpu * world, world, rank)
    seeds = [str(42 + r) for r in ids]
    params = dict(local_search_max_iterations=10_000, window_size=5 * args.n, best_solutions_capacity=32,
                  all_solutions_capacity=100_000, all_solution_iteration_expiry=10_000,
                  iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
    with ls_amd.Context(local_rank, algo=args.algo) as ctx:
        m = ls_amd.MultiSolver(ctx, args.n, seeds, params, n_slots=args.slots, mode=args.mode)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        m.run(max_rounds_per_restart=args.rounds, max_seconds=args.seconds)
        dt = time.perf_counter() - t0
        st = m.stats()
        counters = m.counters()
        recs = [sharding.record(rid, m.best(i)[0], m.best(i)[1]) for i, rid in enumerate(ids) if m.best(i) is not None]
        m.close()
    best, allr = sharding.gather_best(recs)
    tot = torch.tensor([st["neighbours_scored"], st["ls_iterations"], st["gpu_launches"]], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps(dict(n=args.n, n_gpus=world, restarts=len(allr), seconds=float(tmax.item()),
                              neighbours_scored=int(tot[0].item()), neighbours_per_s=tot[0].item() / tmax.item(),
                              ls_iterations=int(tot[1].item()), gpu_launches=int(tot[2].item()),
                              scores=sorted(r[0][0] for r in allr), global_best_score=best[0][0],
                              global_best_restart=best[2], algo=args.algo, slots=args.slots, mode=args.mode,
                              ls_iterations_per_s=tot[1].item() / tmax.item(), rank0_counters=counters)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
