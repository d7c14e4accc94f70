#!/usr/bin/env python3
"""One launch set per kernel family, for `rocprofv3 --pmc ...` (tools/profile_counters.sh): the bench workload of
each N-queens family, the solver's per-iteration step, and the scheduling launches of configs 3 and 5.  Prints the
number of candidates each launch scores, which tools/summarise_counters.py divides the counters by.  Host-pointer
entries only (no torch): the program runs directly under the profiler."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))

import numpy as np  # noqa: E402

# the LAUNCHED kernels are what per-dispatch counters can describe: one-restart host calls take the launch-per-call route here
# (the resident servers are one long dispatch each; their service timelines come from tools/resident_timelines.py)
os.environ.setdefault("LS_NQ_STEP", "3")
os.environ.setdefault("LS_SCHED_SERVER", "0")

import ls_amd  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n, R = 4096, 8
    boards = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)]).astype(np.int32)
    meta = {}
    with ls_amd.Context(0) as ctx:
        _, cur = ctx.nqueens_col_scores(boards)
        cols = np.tile(np.arange(n, dtype=np.int32), (R, 1))
        ncols = np.full(R, n, dtype=np.int32)
        for algo, kernel in (("scan", "nq_scan_regs_kernel"), ("table", "nq_table_quads_kernel")):
            ctx.set_algo(algo)
            for _ in range(reps):
                out, _ = ctx.nqueens_eval_batch(boards, cur, cols, ncols, n * (n - 1))
            meta[kernel] = dict(candidates_per_launch=int(sum(o["n_scored"] for o in out)), restarts=R, n=n,
                                workload="bench workload: 8 restarts x full n x n neighbourhood")
        # the same TABLE kernel with 64 restarts per launch: the same 512 workgroups, eight times the columns per table build
        R64 = 64
        b64 = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R64)]).astype(np.int32)
        _, cur64 = ctx.nqueens_col_scores(b64)
        ctx.set_algo("table")
        for _ in range(reps):
            out, _ = ctx.nqueens_eval_batch(b64, cur64, np.tile(np.arange(n, dtype=np.int32), (R64, 1)), np.full(R64, n, np.int32), n * (n - 1))
        meta["nq_table_quads_kernel_r64"] = dict(candidates_per_launch=int(sum(o["n_scored"] for o in out)), restarts=R64, n=n,
                                                 workload="64 restarts x full n x n neighbourhood (same grid as the 8-restart launch: the second group of dispatches)",
                                                 match="nq_table_quads_kernel", dispatch_slice=[reps, 2 * reps], shares_group_with="nq_table_quads_kernel")
        meta["nq_table_quads_kernel"]["dispatch_slice"] = [0, reps]
        # n = 1024 scan: 32 restarts x 1024 columns
        n2, R2 = 1024, 32
        b2 = np.stack([np.random.default_rng(100 + r).integers(0, n2, size=n2) for r in range(R2)]).astype(np.int32)
        _, cur2 = ctx.nqueens_col_scores(b2)
        ctx.set_algo("scan")
        for _ in range(reps):
            out, _ = ctx.nqueens_eval_batch(b2, cur2, np.tile(np.arange(n2, dtype=np.int32), (R2, 1)), np.full(R2, n2, np.int32),
                                            n2 * (n2 - 1))
        meta["nq_scan_regs_kernel_n1024"] = dict(candidates_per_launch=int(sum(o["n_scored"] for o in out)), restarts=R2, n=n2,
                                                  workload="32 restarts x full 1024 x 1024 neighbourhood",
                                                  match="BoardF32<16>")
        # the solver's per-iteration call: 204 proposed columns, window 5n -> fused step kernel (auto)
        ctx.set_algo("auto")
        rng = np.random.default_rng(12345)
        mc = n // 20
        cols_ref = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
        for RR in (1, R):
            # the solver's steady state: every call continues from the previous winner (board, tables and column
            # scores stay resident); 25 calls so that the first one (fresh upload) does not weigh on the means
            b, c = boards[:RR].copy(), np.array(cur[:RR]).copy()
            for _ in range(25):
                cols_it = np.stack([rng.permutation(n)[:mc] for _ in range(RR)]).astype(np.int32)
                out, _ = ctx.nqueens_eval_batch_next(b, c, cols_it, np.full(RR, mc, np.int32), 5 * n)
                for rr in range(RR):
                    if out[rr]["best_index"] >= 0:
                        b[rr, out[rr]["best_col"]] = out[rr]["best_value"]
                        c[rr] = out[rr]["best_score"]
            meta[f"nq_step_kernel_r{RR}"] = dict(candidates_per_launch=int(sum(o["n_scored"] for o in out)), restarts=RR, n=n,
                                                 workload="LS step: window 5n, 204 proposed columns, continuing from the previous winner (resident board)", match="nq_step_kernel",
                                                 grid_blocks=RR)
        # scheduling, config 5 shape
        rng = np.random.default_rng(2022)
        D, S, E = 31, 32, 10_000
        bits = np.zeros(E, dtype=np.uint64)
        for e in range(E):
            for d in rng.choice(D, size=int(rng.integers(0, 4)), replace=False):
                bits[e] |= np.uint64(1) << np.uint64(d)
        ctx.sched_instance_upload(D, S, 0, E, bits)
        state = rng.integers(0, E, size=(1, S)).astype(np.int32)
        for _ in range(reps):
            r0 = ctx.sched_eval_exhaustive(state, 10 ** 9)[0]
        # the launcher takes 1 / 2 / 3 batches of 64 per wavefront: the first that leaves at most two blocks per CU (sched_launch_delta_pipeline)
        bpw_ex = next((b for b in (1, 2) if (S * E + 512 * b - 1) // (512 * b) <= 2 * 256), 3)
        blocks_ex = (S * E + 512 * bpw_ex - 1) // (512 * bpw_ex)
        meta["sched_eval_all_kernel_exhaustive"] = dict(candidates_per_launch=int(r0["n_scored"]),
                                                        workload="10 000 employees, exhaustive 32 x 10 000, candidates generated in the kernel",
                                                        match="sched_eval_all_kernel<unsigned int, 8, true>", grid_blocks=blocks_ex)
        ex = ctx.sched_exhaustive_moves(state)[0]
        for _ in range(reps):
            r2 = ctx.sched_eval(state[0], ex, 10 ** 9)[0]
        assert r2 == r0, (r2, r0)
        meta["sched_eval_all_kernel_exhaustive_list"] = dict(candidates_per_launch=int(r2["n_scored"]),
                                                             workload="10 000 employees, exhaustive 32 x 10 000 as a move list in HBM",
                                                             match="sched_eval_all_kernel<unsigned int, 8, false>", grid_blocks=blocks_ex)
        mv = np.zeros((70_000, 3), dtype=np.int32)
        for i in range(len(mv)):
            if rng.integers(0, 5) == 0:
                mv[i] = (0, rng.integers(0, D), rng.integers(0, E))
            else:
                a, b = rng.choice(D, size=2, replace=False)
                mv[i] = (1, a, b)
        ctx.sched_moves_upload(mv)
        for _ in range(reps):
            r1 = ctx.sched_eval_resident(state[0], len(mv), 65_536)
        # the scoring launch covers the whole resident list (where the window ends is found afterwards)
        meta["sched_eval_all_kernel_random65536"] = dict(candidates_per_launch=int(r1["n_scored"]), workload="10 000 employees, random stream of 70 000 moves, window 65 536",
                                                         match="sched_eval_all_kernel<unsigned int, 8, false>", grid_blocks=(len(mv) + 511) // 512)
        meta["sched_finish_kernel_random65536"] = dict(candidates_per_launch=int(r1["n_scored"]), workload="finish step of the launch above",
                                                       match="sched_finish_kernel<8>", grid_blocks=1)
        # config 3: the reference example's LS iteration -- 512 resident moves, window 100, one fused block
        D3, S3, E3 = 31, 32, 7
        ctx.sched_instance_upload(D3, S3, 0, E3)
        mv3 = np.zeros((512, 3), dtype=np.int32)
        for i in range(len(mv3)):
            if rng.integers(0, 5) == 0:
                mv3[i] = (0, rng.integers(0, D3), rng.integers(0, E3))
            else:
                a, b = rng.choice(D3, size=2, replace=False)
                mv3[i] = (1, a, b)
        ctx.sched_moves_upload(mv3)
        st3 = rng.integers(0, E3, size=S3).astype(np.int32)
        for _ in range(reps):
            r3 = ctx.sched_eval_resident(st3, 512, 100)
        meta["sched_eval_small_kernel_window100"] = dict(candidates_per_launch=int(r3["n_scored"]),
                                                               workload="example instance, 512 resident moves, window 100: one fused block",
                                                               match="sched_eval_small_kernel<unsigned int, 8>", grid_blocks=1)
    print("PMC_WORKLOADS " + json.dumps(meta))


if __name__ == "__main__":
    main()
