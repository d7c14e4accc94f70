#!/usr/bin/env python3
"""BASELINE.json config 5, end to end: synthetic employee scheduling, 10 000 employees x 31 scored days (32 stored),
holidays h ~ U{0..3} per employee, independent restarts (seed "42"+global restart id; ONE per GPU by default), random move
stream with window 65 536, reference hyper-parameters of examples/employee-scheduling/src/main.rs:22-31.  Restart r on rank
r mod N, no data-path collective: a host-side gather of one (score, solution) record per restart and a minimum in
ScoredSolution order.  Several restarts on one GPU (--restarts-per-gpu 8: the whole of config 5 on ONE GPU) run side by
side as the sub-grids of one resident launch (ls_multi_solver_create_sched).  One process per GPU:

    python tools/run_config5.py [--employees 10000] [--window 65536] [--rounds 250] [--seconds 60] [--restarts-per-gpu 1]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/run_config5.py ...
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
    ap.add_argument("--employees", type=int, default=10_000)
    ap.add_argument("--window", type=int, default=65_536)
    ap.add_argument("--rounds", type=int, default=250)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--restarts-per-gpu", type=int, default=1)
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
    E, D = args.employees, 31
    rng = np.random.default_rng(5)                    # the instance is the same on every rank
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(D, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    params = dict(local_search_max_iterations=1_000, window_size=args.window, best_solutions_capacity=64,
                  all_solutions_capacity=100_000, all_solution_iteration_expiry=1_000,
                  iterated_local_search_max_iterations=args.rounds, max_allow_no_improvement_for=20)
    ids = sharding.shard_restarts(args.restarts_per_gpu * world, world, rank)
    recs, tot = [], np.zeros(3)
    with ls_amd.Context(local_rank) as ctx:
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if len(ids) == 1:                             # one restart per GPU: the plain solver (its own one-sub-grid launch)
            rid = ids[0]
            s = ls_amd.Solver.scheduling(ctx, D, 0, E, str(42 + rid), params, holiday_bits=bits)
            rounds = 0
            while not s.is_finished() and rounds < args.rounds and time.perf_counter() - t0 < args.seconds:
                s.execute_round()
                rounds += 1
            st, b = s.stats(), s.best()
            tot += (st["neighbours_scored"], st["ls_iterations"], rounds)
            if b is not None:
                recs.append(sharding.record(rid, b[0], b[1]))
            s.close()
        else:                                         # several: side by side, the sub-grids of ONE resident launch
            m = ls_amd.MultiSolver.scheduling(ctx, D, 0, E, [str(42 + rid) for rid in ids], params, holiday_bits=bits, n_slots=len(ids))
            m.run(max_rounds_per_restart=args.rounds, max_seconds=args.seconds)
            st = m.stats()
            tot += (st["neighbours_scored"], st["ls_iterations"], st["rounds"])
            for i, rid in enumerate(ids):
                b = m.best(i)
                if b is not None:
                    recs.append(sharding.record(rid, b[0], b[1]))
            m.close()
        dt = time.perf_counter() - t0
    best, allr = sharding.gather_best(recs)
    t_tot = torch.tensor(tot, dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps(dict(employees=E, window=args.window, n_gpus=world, restarts=len(allr), seconds=float(tmax.item()),
                              neighbours_scored=int(t_tot[0].item()), neighbours_per_s=t_tot[0].item() / tmax.item(),
                              ls_iterations=int(t_tot[1].item()), rounds=int(t_tot[2].item()),
                              scores=sorted(list(r[0]) for r in allr), global_best_score=list(best[0]) if best else None,
                              global_best_restart=best[2] if best else None)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
