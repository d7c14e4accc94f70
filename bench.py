#!/usr/bin/env python3
"""bench.py — neighbour evaluations / second of the N-queens neighbourhood
evaluator on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (kernel-throughput mode): per GPU 8
independent restarts, each a fixed pseudo-random n = 4096 board (uniform rows,
seed = global restart id); one STEP = one launch that scores the FULL n x n
neighbourhood of every restart (all columns in order, window = n(n-1) non-tabu
candidates) and selects each restart's (score, board)-minimum.  Inputs are
resident in HBM before the timed region.  Weak scaling: restarts per GPU fixed.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_neighbour(n: int) -> int:
    """one scan of the packed int32 state (4n) + 8 B move descriptor + 8 B result."""
    return 4 * n + 16


def make_boards(n: int, restart_ids) -> np.ndarray:
    return np.stack([np.random.default_rng(int(r)).integers(0, n, size=n) for r in restart_ids]).astype(np.int32)


def cpu_baseline(n: int, board: np.ndarray, target_seconds: float):
    """The oracle's literal restatement of the reference CPU path (clone + tabu test + full
    O(n^2) rescoring + running (score, board) minimum per candidate), single thread, on a
    bounded sample of the same workload: the first `sample` candidates of restart 0."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    orc.build()
    cols = np.arange(n)
    t0 = time.perf_counter()
    ns, _ = orc.nq_baseline_run(board, cols, 32)
    dt = max(time.perf_counter() - t0, 1e-6)
    sample = int(max(64, min(n * (n - 1), target_seconds * ns / dt)))
    t0 = time.perf_counter()
    ns, best = orc.nq_baseline_run(board, cols, sample)
    dt = time.perf_counter() - t0
    out = dict(value=ns / dt, unit="neighbour evals/s", cores=1, kind="port",
               sample=f"first {ns} non-tabu candidates of restart 0 (n={n}, full O(n^2) rescoring each), "
                      f"{dt:.1f} s on one host core; reference is single-threaded Rust, unbuildable here")
    out["other_lines"] = cpu_other_lines(n, board, ns / dt)
    return out, ns, best


def host_core_share() -> int:
    """Cores this process may really use: the cgroup CPU quota when there is one, else the affinity mask,
    never more than 16 (the per-GPU CPU share of the benchmark boxes)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 16))


def cpu_other_lines(n, board, single_rate):
    """(2) one independent restart per host core, same per-candidate work as the
    reference; (5) the oracle's O(n) delta and O(1) count-table scorers on one core — NOT the reference
    algorithm, shown to separate algorithmic gain from hardware gain.  A few seconds in total.
    Threads, not processes: the oracle is called through ctypes, which releases the GIL."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle as orc
    lines = {}
    cores = host_core_share()
    sample = max(64, int(single_rate * 4.0))            # about 4 s per core
    cols = np.arange(n)

    def one(rid):
        b = np.random.default_rng(1000 + rid).integers(0, n, size=n)
        t0 = time.perf_counter()
        ns, _ = orc.nq_baseline_run(b, cols, sample)
        return ns, time.perf_counter() - t0

    try:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            res = list(ex.map(one, range(cores)))
        wall = time.perf_counter() - t0
        lines["reference_algorithm_all_host_cores"] = {
            "value": sum(r[0] for r in res) / wall, "unit": "neighbour evals/s", "cores": cores,
            "sample": f"{cores} independent restarts, {sample} candidates each, one thread per core, {wall:.1f} s"}
    except Exception as e:  # a baseline line must never break the benchmark
        lines["reference_algorithm_all_host_cores"] = {"error": repr(e)}
    few = np.arange(min(n, 64))
    for form in ("delta", "table"):
        t0 = time.perf_counter()
        r = orc.nq_neighbourhood(board, few, len(few) * (n - 1), form, want_scores=False)
        dt = max(time.perf_counter() - t0, 1e-9)
        lines[f"not_reference_algorithm_{form}_one_core"] = {
            "value": r["n_scored"] / dt, "unit": "neighbour evals/s", "cores": 1,
            "sample": f"{r['n_scored']} candidates ({len(few)} columns), oracle's {form} formulation"}
    return lines


def ensure_built():
    """The shared library is a git-ignored build artefact: build it (hipcc, gfx950) when a clean checkout
    has none.  Only local rank 0 builds; the other ranks wait for the file."""
    so = os.path.join(ROOT, "constraint-solver_amd", "lib", "libls_amd.so")
    if os.path.exists(so):
        return
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import importlib.util
        spec = importlib.util.spec_from_file_location("ls_amd_build", os.path.join(ROOT, "constraint-solver_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    else:
        t0 = time.time()
        while not os.path.exists(so) and time.time() - t0 < 600:
            time.sleep(1.0)
        time.sleep(2.0)  # let the linker finish writing


VALU_CLOCK_HZ = 2.4e9          # MI355X max engine clock (MI355X_MICROARCH.md: "Max clock 2400 MHz")
VALU_CYCLES_PER_WAVE_INSTR = 2  # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (same guide, "SIMD / EU")


def valu_issue_peak_wave_instr_per_s(n_cus: int) -> float:
    """Spec VALU issue peak: 4 SIMDs per CU, one wave64 instruction per 2 cycles per SIMD at 2.4 GHz
    (= 78.6 T lane-operations/s on 256 CUs, half the 157.3 TFLOP/s FMA figure: one operation per lane per instruction)."""
    return 4 * n_cus * VALU_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INSTR


class _ThreadSync:
    """Barrier + max-over-ranks + gather for ranks that are host threads of ONE process (one host thread,
    one context and one stream per GPU; the path has no collective, so no torch.distributed is needed)."""

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world

    def barrier(self):
        self._bar.wait()

    def all_gather(self, rank, obj):
        self._slots[rank] = obj
        self._bar.wait()
        out = list(self._slots)
        self._bar.wait()
        return out


class _DistSync:
    """Ranks are processes (torch.distributed.run): the barrier runs on the job's backend (RCCL under `nccl`); the
    host-side gather of the result records goes through a gloo group — it is CPU data and needs no GPU transport."""

    def __init__(self, dist, world, object_group=None):
        self.dist, self.world, self.object_group = dist, world, object_group

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def all_gather(self, rank, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.object_group)
        return out


def run_rank(args, rank, world, device, sync, launcher):
    """One rank = one GPU: `device` is the HIP device index; `sync` provides barrier / all_gather over the ranks."""
    import torch

    import ls_amd
    from ls_amd import sharding

    dev = torch.device("cuda", device)
    n, R = args.n, args.restarts_per_gpu
    if args.total_restarts is not None:
        if args.total_restarts % world:
            raise SystemExit(f"--total-restarts {args.total_restarts} is not divisible by the {world} ranks")
        R = args.total_restarts // world
    with torch.cuda.device(dev):
        # restart r -> rank r mod world; R restarts per GPU
        my_ids = sharding.shard_restarts(R * world, world, rank)
        boards = make_boards(n, my_ids)
        ctx = ls_amd.Context(device, algo=args.algo)
        info = ctx.device_info()
        # a dedicated (non-default) torch stream: HIP events recorded on it bracket the kernels
        # (ls_ctx_set_stream(NULL) would select the context's own stream, and 0 is torch's default stream)
        stream = torch.cuda.Stream(device=dev)
        assert stream.cuda_stream != 0
        ctx.set_stream(stream.cuda_stream)

        # current scores from the GPU col-score kernel (bit-identical to get_col_scores; tested)
        _, cur_np = ctx.nqueens_col_scores(boards)
        rows_d = torch.from_numpy(boards).to(dev)
        cur_d = torch.from_numpy(cur_np).to(dev)
        cols_d = torch.arange(n, dtype=torch.int32, device=dev).repeat(R, 1).contiguous()
        ncols_d = torch.full((R,), n, dtype=torch.int32, device=dev)
        res_d = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        window = n * (n - 1)
        neighbours_per_step = R * n * (n - 1)

        def step():
            ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_d.data_ptr(), ncols_d.data_ptr(),
                                       n, window, res_d.data_ptr())

        with torch.cuda.stream(stream):
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize(dev)
            sync.barrier()
            torch.cuda.synchronize(dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            t0 = time.perf_counter()
            for i in range(args.steps):
                ev[i][0].record(stream)
                step()
                ev[i][1].record(stream)
            torch.cuda.synchronize(dev)
            sync.barrier()
            torch.cuda.synchronize(dev)
            elapsed = time.perf_counter() - t0
        launch_ms = [a.elapsed_time(b) for a, b in ev]

        # results: host-side gather of one (score, board) record per restart, global minimum (no collective on the data path)
        res = res_d.cpu().numpy()
        recs = []
        for i, rid in enumerate(my_ids):
            b = boards[i].copy()
            col, val = int(res[i, 3] & 0xFFFFFFFF), int(res[i, 3] >> 32)
            b[col] = val
            recs.append(sharding.record(rid, int(res[i, 2]), b))
            assert res[i, 0] == n * (n - 1), "every non-tabu candidate of the full neighbourhood must be scored"
        gathered = sync.all_gather(rank, dict(elapsed=elapsed, recs=recs, launch_ms=launch_ms))
        elapsed = max(g["elapsed"] for g in gathered)                    # MAX over ranks
        all_recs = [r for g in gathered for r in g["recs"]]
        best = sharding.best_of(all_recs)

        # the scheduling line (config 5 shape, one restart per rank): every rank takes part, also at N > 1
        sched_line = None
        if not args.no_sched:
            sched_line = sched_rank_line(args, ctx, torch, dev, stream, rank, world, sync)
        out = None
        if world > 1:
            print(f"[bench rank {rank}] device {device}, host cores this rank may use: {host_core_share()} "
                  f"(affinity {len(os.sched_getaffinity(0))})", file=sys.stderr, flush=True)
        if rank == 0:
            value = neighbours_per_step * world * args.steps / elapsed
            avg_launch_s = float(np.mean(launch_ms)) / 1e3
            out = build_report(args, world, R, n, info, value, elapsed, avg_launch_s, neighbours_per_step, window, best, launcher)
            roof = out["roofline"]
            out["verified"] = verify_results(n, all_recs)           # every restart's record vs the oracle (after the timed region)
            if not out["verified"]["ok"]:
                print(json.dumps(out), flush=True)
                raise SystemExit("bench.py: the timed launch's results differ from the oracle")
            if sched_line is not None:
                out["sched"] = sched_line
                roof["sched_gpu_per_s"] = sched_line["value"]
                roof["sched_gpu_us_per_step"] = sched_line["us_per_step"]
                roof["sched_verified"] = bool(sched_line["verified"]["ok"])
                if not sched_line["verified"]["ok"]:
                    print(json.dumps(out), flush=True)
                    raise SystemExit("bench.py: the timed scheduling launch's results differ from the oracle")
            if not args.no_cpu_baseline and world == 1:
                cb, ns, cb_best = cpu_baseline(n, boards[0], args.cpu_seconds)
                out["cpu_baseline"] = cb
                out["cpu_baseline"]["gpu_over_cpu"] = value / cb["value"]
            if not args.no_extra and world == 1:
                with torch.cuda.stream(stream):
                    out["extra"] = extra_measurements(ctx, torch, dev, n, R, boards, cur_np, info, e2e=not args.no_e2e,
                                                      sched_cpu_seconds=0.0 if args.no_cpu_baseline else 0.4)
                ex = out["extra"]
                # what LS_NQ_ALGO_AUTO (the library default) gives a caller, next to the headline scan kernel: kept inside
                # `roofline` so that it survives consumers that drop `extra`
                tb, e2e = ex.get("table_full_neighbourhood", {}), ex.get("config4_end_to_end_64_restarts_one_gpu", {})
                tb64 = ex.get("table_full_neighbourhood_64_restarts", {})
                st1 = ex.get("fused_ls_step_host_round_trip_1_restarts", {})  # (bare C entry where measured)
                eq = (e2e.get("timed_run_vs_single_solver_runs_on_the_launch_per_call_route") or {}).get("equal")
                roof["default_path"] = {
                    "table_neighbours_per_s": tb.get("neighbours_per_s"), "table_frac": (tb.get("roofline") or {}).get("frac"),
                    "table_bound": (tb.get("roofline") or {}).get("bound"),
                    "table_64_restarts_neighbours_per_s": tb64.get("neighbours_per_s"),
                    "table_64_restarts_frac": (tb64.get("roofline") or {}).get("frac"),
                    "ls_step_round_trip_us_one_restart": st1.get("us_per_call_bare_c_entry") or st1.get("us_per_call"),
                    "ls_step_round_trip_us_one_restart_through_the_python_wrapper": st1.get("us_per_call"),
                    "ls_step_round_trip_us_8_restarts_per_call": ex.get("fused_ls_step_host_round_trip_8_restarts", {}).get("us_per_call_bare_c_entry"),
                    "config4_end_to_end_s": e2e.get("seconds"), "config4_end_to_end_neighbours_per_s": e2e.get("neighbours_per_s"),
                    "config4_end_to_end_equals_single_solver_runs": eq}
                roof["int_formulation"] = ex.get("scan_int_formulation_packed_u16")
                # the same facts as FLAT scalars: consumers of the line keep scalar keys of `roofline` only
                hs = roof.get("hbm_effective_secondary") or {}
                flat = {
                    "table_frac": (tb.get("roofline") or {}).get("frac"), "table_per_s": tb.get("neighbours_per_s"),
                    "table_r64_frac": (tb64.get("roofline") or {}).get("frac"), "table_r64_per_s": tb64.get("neighbours_per_s"),
                    "int_formulation_per_s": (ex.get("scan_int_formulation_packed_u16") or {}).get("neighbours_per_s"),
                    "scan_n1024_x32_per_s": (ex.get("scan_full_neighbourhood_n1024_x32") or {}).get("neighbours_per_s"),
                    "scan_n8192_x32_per_s": (ex.get("scan_full_neighbourhood_n8192_x32") or {}).get("neighbours_per_s"),
                    "scan_other_sizes_verified": bool((ex.get("scan_full_neighbourhood_n1024_x32") or {}).get("verified_restart_0")
                                                      and (ex.get("scan_full_neighbourhood_n8192_x32") or {}).get("verified_restart_0")),
                    "hbm_measured_bytes": hs.get("measured_hbm_bytes_per_launch"), "hbm_effective_over_peak": hs.get("effective_over_hbm_peak"),
                    "ls_step_us": st1.get("us_per_call_bare_c_entry") or st1.get("us_per_call"),
                    "c4_e2e_s": e2e.get("seconds"), "c4_e2e_per_s": e2e.get("neighbours_per_s"), "c4_e2e_equal": eq,
                    "c2_us_per_ls_iteration": (ex.get("config2_end_to_end_nq1024_seed42") or {}).get("us_per_ls_iteration"),
                    "c3_us_per_ls_iteration": (ex.get("config3_end_to_end_sched_example_seed42") or {}).get("us_per_ls_iteration"),
                    "c5_us_per_ls_iteration": (ex.get("config5_end_to_end_sched10k_window65536_seed42") or {}).get("us_per_ls_iteration"),
                    "sched_legs_verified": (ex.get("sched_verified") or {}).get("ok"),
                }
                legs = (ex.get("sched_cpu_baseline") or {}).get("legs") or {}
                c5 = legs.get("sched10k_random_stream_window_65536") or {}
                flat.update({"sched_cpu_per_s": c5.get("one_core_per_s"), "sched_cpu_all_cores_per_s": c5.get("all_cores_per_s"),
                             "sched_cpu_cores": c5.get("cores"),
                             "sched_example_cpu_per_s": (legs.get("sched_example_window100") or {}).get("one_core_per_s"),
                             "sched_exhaustive_cpu_per_s": (legs.get("sched10k_exhaustive_320k") or {}).get("one_core_per_s"),
                             "sched_exhaustive_gpu_per_s": (ex.get("sched10k_exhaustive_320k") or {}).get("neighbours_per_s"),
                             "sched_example_x64_gpu_per_s": (ex.get("sched_example_window100_x64_restarts") or {}).get("neighbours_per_s")})
                for tagk, key in (("nq_srv", "fused_ls_step_host_round_trip_1_restarts"), ("c3_srv", "config3_end_to_end_sched_example_seed42"),
                                  ("c5_grid", "config5_end_to_end_sched10k_window65536_seed42"), ("xgrid", "sched10k_exhaustive_320k_host_round_trip_generated_on_device")):
                    rr = (ex.get(key) or {}).get("resident") or {}
                    flat.update({f"{tagk}_service_us": rr.get("service_us"), f"{tagk}_frac": rr.get("frac"), f"{tagk}_mailbox_us": rr.get("mailbox_us")})
                ms = ex.get("config3_multi_restart_end_to_end") or {}
                flat.update({"c3_x64_e2e_per_s": ms.get("neighbours_per_s"), "c3_x64_e2e_equal": ms.get("equal_single_solver_runs")})
                m5 = ex.get("config5_multi_restart_end_to_end") or {}
                flat.update({"c5_x8_e2e_s": m5.get("seconds"), "c5_x8_e2e_per_s": m5.get("neighbours_per_s"), "c5_x8_over_single": m5.get("wall_over_single_restart"),
                             "c5_x8_e2e_equal": m5.get("equal_single_solver_runs")})
                roof.update({k: v for k, v in flat.items() if v is not None})
                if "cpu_baseline" in out and c5.get("one_core_per_s"):
                    out["cpu_baseline"]["sched_value"] = c5["one_core_per_s"]
                    out["cpu_baseline"]["sched_all_cores_value"] = c5.get("all_cores_per_s")
                    out["cpu_baseline"]["sched_sample"] = (f"first {c5.get('one_core_sample')} candidates of the window-65536 move list at 10 000 employees "
                                                           f"(full get_scored_solution each), {c5.get('one_core_seconds', 0):.2f} s on one core; "
                                                           f"{c5.get('cores')} cores: {c5.get('all_cores_sample')} candidates in {c5.get('all_cores_seconds', 0):.2f} s")
                failed = [k for k in ("config4_end_to_end_64_restarts_one_gpu", "config3_multi_restart_end_to_end", "config5_multi_restart_end_to_end")
                          if (ex.get(k) or {}).get("error")]
                if failed or not (ex.get("sched_verified") or {"ok": True})["ok"]:
                    print(json.dumps(out), flush=True)
                    raise SystemExit("bench.py: " + (str(ex[failed[0]]["error"]) if failed else "a timed scheduling launch differs from the oracle"))
            print(json.dumps(out), flush=True)
        sync.barrier()
        ctx.set_stream(None)
        ctx.close()
    return out


def verify_results(n, all_recs):
    """Parity gate of the timed region: the (score, winner board) record of EVERY restart, as produced by the timed
    launches, against the oracle's count-table formulation of local_search.rs:315-325 on the same board (boards are
    regenerated from the restart id).  The oracle is the checker here, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    orc.build()
    t0 = time.perf_counter()
    bad = []
    cols = np.arange(n)
    for (score, board, rid) in all_recs:
        base = make_boards(n, [rid])[0]
        exp = orc.nq_neighbourhood(base, cols, n * (n - 1), "table", want_scores=False)
        want = base.copy()
        want[exp["best_index"] // n] = exp["best_index"] % n
        if score != (exp["best_score"],) or tuple(int(x) for x in want) != board or exp["n_scored"] != n * (n - 1):
            bad.append(rid)
    return {"ok": not bad, "restarts_checked": len(all_recs), "mismatching_restarts": bad,
            "against": "oracle/nq_oracle.c table formulation, (score, winner board) per restart",
            "seconds": time.perf_counter() - t0}


def build_report(args, world, R, n, info, value, elapsed, avg_launch_s, neighbours_per_step, window, best, launcher):
    bpn = algorithmic_bytes_per_neighbour(n)
    eff_gbps = neighbours_per_step * bpn / avg_launch_s / 1e9
    counters = load_counters()
    traffic = counters.get("hbm_traffic", {}).get(f"nq_{args.algo}_n{n}_r{R}", {}).get("bytes_per_launch")
    kernel = ("nq_scan_regs_kernel<BoardF32<64>, false>" if (args.algo == "scan" and 2048 < n <= 4096)
              else ("nq_table_quads_kernel" if args.algo == "table" else "nq_scan_*"))
    # VALU-issue roofline of the scan kernel: per neighbour every lane runs full-rate fp32 instructions
    # for each of its ceil64(n)/64 columns -- 4 (sub, sub, mul+clamp, add) in the one-column walk; 3.5 in the two-column
    # walk of 512 < n <= 4096, where two list columns share the first sub (whole columns only: this workload's case).
    # achieved counts ONLY those algorithmic instructions (measured SQ_INSTS_VALU per neighbour is in counters.*), peak
    # is the spec issue rate.
    simds = 4 * info["n_cus"]
    two_column_walk = args.algo == "scan" and 512 < n <= 4096
    algo_wave_instr = (3.5 if two_column_walk else 4.0) * ((n + 63) // 64)
    peak_wi = valu_issue_peak_wave_instr_per_s(info["n_cus"])
    ach_wi = neighbours_per_step * algo_wave_instr / avg_launch_s
    roof = {
        "bound": "valu_issue",
        "achieved": ach_wi * 64 / 1e12,
        "peak": peak_wi * 64 / 1e12,
        "unit": "TFLOP/s",
        "frac": ach_wi / peak_wi,
        "traffic": traffic,
        "kernel": kernel,
        "avg_launch_ms": avg_launch_s * 1e3,
        "algorithmic_wave_instr_per_neighbour": algo_wave_instr,
        "peak_source": "MI355X_MICROARCH.md: wave64 VALU instruction = 2 cycles per SIMD-32, 4 SIMDs x n_cus, 2.4 GHz max clock; "
                       "1 fp32 operation per lane per instruction (sub / mul / add, no fused pairs), i.e. half the 157.3 TFLOP/s FMA peak",
        "note": "no MFMA and not HBM-bound: the 16 KiB state of a restart is read once per wavefront and held in registers; "
                "every candidate then costs 3.5 (two list columns per pass sharing rows[j] - v; 4 in the one-column walk) fp32 instructions per column (exact integer-valued arithmetic), so the limiter is VALU issue",
        # SURVEY 8(d)'s fixed byte accounting, kept as a labelled secondary: NOT a usable roofline fraction (> 1), because
        # those bytes are served from registers, not from HBM; measured HBM traffic is `traffic` (rocprofv3 counters)
        "hbm_effective_secondary": {
            "algorithmic_bytes_per_neighbour": bpn, "effective_GBps": eff_gbps, "hbm_peak_GBps": HBM_PEAK_GBPS,
            "effective_over_hbm_peak": eff_gbps / HBM_PEAK_GBPS,
            "measured_hbm_bytes_per_launch": traffic,
            "measured_hbm_GBps": (traffic / avg_launch_s / 1e9) if traffic else None,
            "measured_hbm_frac_of_peak": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            "note": "SURVEY 8(d): 4n+16 bytes per neighbour; exceeds the HBM peak because the state is shared by all candidates of a "
                    "restart and lives in VGPRs. Counter traffic: profiles/hbm_traffic.json"},
    }
    c = counters.get("kernels", {}).get("nq_scan_regs_kernel")
    if c:   # committed rocprofv3 counters of this kernel on this workload (profiles/): what the kernel really issues
        roof["counters"] = {k: c.get(k) for k in ("valu_wave_instr_per_candidate", "lds_wave_instr_per_candidate",
                                                    "salu_wave_instr_per_candidate", "duration_us", "valu_issue_bound_us", "source")}
        roof["counters"]["measured_issue_frac"] = (c["valu_issue_bound_us"] / 1e3 / (avg_launch_s * 1e3)) if c.get("valu_issue_bound_us") else None
    return {
        "metric": "neighbor evals/sec (N=4096 nqueens)" if n == 4096 else f"neighbor evals/sec (N={n} nqueens)",
        "value": value,
        "unit": "neighbour evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if args.total_restarts is None else "strong",
        "vs_baseline": None,
        # scores are i64 at the boundary (NQueensScore, examples/nqueens/src/lib.rs:13,63); the scan kernel decides "does column j
        # attack (c, v)" on integers < 2^15 held exactly in fp32 (d * (d - e) == 0: no rounding can create or destroy a zero); the
        # all-integer formulation of the same kernel is timed beside it (roofline.int_formulation)
        "dtype": "i64 scores / exact-integer fp32 compares" if args.algo == "scan" and n <= 4096 else "i64 scores / int32 compares",
        "data": "synthetic",
        "config": {
            "workload": f"nqueens n={n}, {R} independent restarts per GPU ({R * world} total, restart r on rank r mod {world}), "
                        f"full n x n neighbourhood per restart per launch (window = n(n-1) = {window}), "
                        f"uniform-random boards seeded by restart id, algo={args.algo}",
            "n": n, "restarts_per_gpu": R, "neighbours_per_step_per_gpu": neighbours_per_step,
            "algo": args.algo, "device": info["name"], "n_cus": info["n_cus"],
            "ranks": launcher,
            "algo_note": "scan = the reference's O(n) pair checks per candidate, the kernel the metric and SURVEY 8(d)'s byte "
                         "accounting are defined on; the library default (auto) resolves to the count-table family, "
                         "reported under extra.table_*",
            "arithmetic": "state int32, scores int64; the scan kernel tests attacks with exact integer-valued fp32 "
                          "arithmetic (|values| < 2^24), see DESIGN.md 3.1",
        },
        "roofline": roof,
        # SURVEY 8(d) secondary metric.  Each neighbour tests its new placement against the n - 1 other
        # columns; the old placement's n - 1 tests are done once per column and shared by its n - 1
        # neighbours, so the kernel performs (n - 1) + 1 pair checks per neighbour (the survey's 2(n - 1)
        # figure counts the shared half once per neighbour).
        "secondary": {"pair_constraint_checks_per_s": value * n,
                      "pair_constraint_checks_per_s_survey_convention_2x": value * 2 * (n - 1)},
        "global_best": {"score": best[0][0], "restart": best[2]},
    }


def load_counters():
    """Committed rocprofv3 summaries (profiles/): HBM traffic per launch and per-kernel SQ instruction counts."""
    out = {}
    for key, name in (("hbm_traffic", "hbm_traffic.json"), ("kernels", "kernel_counters.json"), ("resident", "resident_timelines.json")):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                out[key] = json.load(f)
    return out


def resident_route(route, live_us=None):
    """{service_us, bound_us, frac, mailbox_us} of a RESIDENT route (no launch per request) from the committed stamps-build
    timelines (profiles/resident_timelines.json, tools/resident_timelines.py): service = request seen -> answer written inside
    the workgroup, bound = the VALU-issue bound of the same work from the launched twin kernel's counters, mailbox = the answer
    wait minus the service (two PCIe crossings + poll delays).  live_us: what this run measured for the whole call / iteration."""
    r = (load_counters().get("resident") or {}).get("routes", {}).get(route)
    if not r or "service_us" not in r:
        return None
    out = {k: r.get(k) for k in ("service_us", "bound_us", "frac", "mailbox_us", "bound", "answer_wait_us", "bound_from")}
    out["route"] = route
    out["timelines"] = "profiles/resident_timelines.json (in-kernel stamps, -DLS_NQ_STAMPS build)"
    if live_us:
        out["service_share_of_live_call"] = r["service_us"] / live_us
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher, process flavour (--spawn): N fresh child processes, one per GPU,
    started BEFORE this process touches the GPU, rendezvous on 127.0.0.1.  The parent relays rank 0's JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus))
        argv = [a for a in sys.argv[1:] if a != "--spawn"]
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, abs(p.wait()))
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--board-size", "--n", dest="n", type=int, default=4096,
                    help="(under torch.distributed.run use --board-size: its parser treats --n as ambiguous)")
    ap.add_argument("--restarts-per-gpu", type=int, default=8)
    ap.add_argument("--total-restarts", type=int, default=None,
                    help="strong-scaling mode (SURVEY 8(e)): a fixed total, e.g. 64, split evenly over the ranks")
    ap.add_argument("--algo", choices=["scan", "table"], default="scan")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-sched", action="store_true", help="skip the scheduling line (config 5 shape, one restart per rank)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs of `extra` (config 2-5 solver runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse ranks on one GPU)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this one device")
    ap.add_argument("--spawn", action="store_true",
                    help="without a launcher and --gpus N > 1: N child processes (torch.distributed) instead of N host threads")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={env_world} but --gpus {args.gpus}; launch N ranks for --gpus N")
    ensure_built()
    if env_world is None and args.gpus > 1 and args.spawn:
        spawn_ranks(args)           # never returns

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the evaluator has no CPU fallback")
    if args.all_on_device is None and args.gpus > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible "
                         "(--all-on-device D rehearses the ranks on one device)")

    if env_world is None and args.gpus > 1:
        # No launcher: ONE process, one host thread + one context + one stream per GPU (SURVEY 8(e) / north star).
        # The path has no collective, so the ranks only meet at the barriers and at the host-side gather.
        import threading

        world = args.gpus
        sync = _ThreadSync(world)
        errors = []

        def target(rank):
            try:
                device = args.all_on_device if args.all_on_device is not None else rank
                run_rank(args, rank, world, device, sync, "host threads of one process, one context + stream per GPU, no collective")
            except BaseException as e:  # noqa: BLE001 - a failed rank must not leave the others at a barrier
                errors.append((rank, repr(e)))
                sync._bar.abort()

        threads = [threading.Thread(target=target, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise SystemExit(f"bench.py: rank failure(s): {errors}")
        return

    # One process per GPU under torch.distributed.run (the driver's launch for N > 1), or the plain 1-GPU run.
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    device = args.all_on_device if args.all_on_device is not None else local_rank
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and args.all_on_device is None:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")     # several ranks on one device: RCCL refuses duplicate GPUs
    object_group = None
    if world > 1 and dist.get_backend() != "gloo":
        object_group = dist.new_group(backend="gloo")
    sync = _DistSync(dist, world, object_group)
    run_rank(args, rank, world, device, sync,
             "torch.distributed.run, one process per GPU; RCCL for the barrier only, host-side gather" if world > 1 else "single process")
    if world > 1:
        dist.destroy_process_group()


def extra_measurements(ctx, torch, dev, n, R, boards, cur_np, info, e2e=True, sched_cpu_seconds=0.4):
    """Secondary lines (not `value`): the other kernel family on the same workload and the
    reference-shaped small launch (<= 204 proposed columns, window 5n), SURVEY 8(d) config 4."""
    out = {}
    rows_d = torch.from_numpy(boards).to(dev)
    cur_d = torch.from_numpy(cur_np).to(dev)
    res_d = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    kc = load_counters().get("kernels", {})

    def issue_roofline(workload, live_s):
        """{bound, frac} of a secondary kernel: the larger of its VALU-issue and LDS-array bounds (instruction counts from
        the committed rocprofv3 counters of the same workload, spec clock, every SIMD / LDS array busy) over the live time."""
        c = kc.get(workload)
        if not c or "valu_issue_bound_us" not in c:
            return None
        bounds = {"valu_issue": c["valu_issue_bound_us"]}
        if "lds_array_bound_us" in c:
            bounds["lds_array"] = c["lds_array_bound_us"]
        b = max(bounds, key=bounds.get)
        return {"bound": b, "frac": bounds[b] / (live_s * 1e6), "bound_us": bounds[b],
                "valu_wave_instr_per_candidate": c.get("valu_wave_instr_per_candidate"),
                "lds_wave_instr_per_candidate": c.get("lds_wave_instr_per_candidate"), "counters": c.get("source")}

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters / 1e3

    cols_full = torch.arange(n, dtype=torch.int32, device=dev).repeat(R, 1).contiguous()
    ncols_full = torch.full((R,), n, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(12345)
    mc = max(1, min(n // 20, n))
    cols_ref = torch.from_numpy(np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)).to(dev)
    ncols_ref = torch.full((R,), mc, dtype=torch.int32, device=dev)
    for algo in ("scan", "table"):
        ctx.set_algo(algo)
        s = timed(lambda: ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_full.data_ptr(),
                                                     ncols_full.data_ptr(), n, n * (n - 1), res_d.data_ptr()),
                  5 if algo == "scan" else 50)
        out[f"{algo}_full_neighbourhood"] = {"neighbours_per_s": R * n * (n - 1) / s, "ms_per_launch": s * 1e3,
                                             "roofline": issue_roofline("nq_scan_regs_kernel" if algo == "scan" else
                                                                        ("nq_table_quads_kernel" if "nq_table_quads_kernel" in kc else "nq_table_pairs_kernel"), s)}
        w = 5 * n
        s = timed(lambda: ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_ref.data_ptr(),
                                                     ncols_ref.data_ptr(), mc, w, res_d.data_ptr()), 200)
        out[f"{algo}_reference_shaped_window_5n"] = {"neighbours_per_s": R * min(w, mc * (n - 1)) / s,
                                                     "us_per_launch": s * 1e6, "candidates_per_launch": R * min(w, mc * (n - 1))}
    # the default TABLE kernel with 64 restarts per launch (config 4's restart count): the same 512 workgroups, eight times the
    # columns per table build -- the 8-restart launch above is bounded by the workgroup dispatch ramp, not by issue
    try:
        R64 = 64
        b64 = make_boards(n, range(R64))
        _, cur64 = ctx.nqueens_col_scores(b64)
        rows64, cur64_d = torch.from_numpy(b64).to(dev), torch.from_numpy(cur64).to(dev)
        cols64 = torch.arange(n, dtype=torch.int32, device=dev).repeat(R64, 1).contiguous()
        nc64 = torch.full((R64,), n, dtype=torch.int32, device=dev)
        res64 = torch.zeros((R64, 4), dtype=torch.int64, device=dev)
        ctx.set_algo("table")
        s = timed(lambda: ctx.nqueens_eval_batch_dev(R64, n, rows64.data_ptr(), cur64_d.data_ptr(), cols64.data_ptr(), nc64.data_ptr(), n,
                                                     n * (n - 1), res64.data_ptr()), 30)
        out["table_full_neighbourhood_64_restarts"] = {"neighbours_per_s": R64 * n * (n - 1) / s, "us_per_launch": s * 1e6,
                                                       "roofline": issue_roofline("nq_table_quads_kernel_r64" if "nq_table_quads_kernel_r64" in kc else "nq_table_pairs_kernel_r64", s)}
        del rows64, cur64_d, cols64, nc64, res64
    except Exception as e:  # noqa: BLE001 - a secondary line must not take the bench down
        out["table_full_neighbourhood_64_restarts"] = {"error": repr(e)}
    # the scan kernel's other tiers on full neighbourhoods: n = 1024 (16-register tier, config 2's size) and n = 8192 (the LDS
    # kernel, 2048 of its columns); restart 0's record of each timed launch is re-checked against the oracle afterwards
    if n == 4096:
        for n2, R2, mc2 in ((1024, 32, 1024), (8192, 32, 2048)):
            key = f"scan_full_neighbourhood_n{n2}_x{R2}"
            try:
                b2 = make_boards(n2, range(R2))
                _, cur2 = ctx.nqueens_col_scores(b2)
                rows2, cur2_d = torch.from_numpy(b2).to(dev), torch.from_numpy(cur2).to(dev)
                cols2 = torch.arange(mc2, dtype=torch.int32, device=dev).repeat(R2, 1).contiguous()
                nc2 = torch.full((R2,), mc2, dtype=torch.int32, device=dev)
                res2 = torch.zeros((R2, 4), dtype=torch.int64, device=dev)
                ctx.set_algo("scan")
                s = timed(lambda: ctx.nqueens_eval_batch_dev(R2, n2, rows2.data_ptr(), cur2_d.data_ptr(), cols2.data_ptr(), nc2.data_ptr(),
                                                             mc2, mc2 * (n2 - 1), res2.data_ptr()), 3)
                torch.cuda.synchronize()
                got = res2.cpu().numpy()[0]
                if os.path.join(ROOT, "oracle") not in sys.path:
                    sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as orc                                   # the checker, after the timed region
                exp = orc.nq_neighbourhood(b2[0], np.arange(mc2), mc2 * (n2 - 1), "table", want_scores=False)
                ok = (int(got[0]), int(got[1]), int(got[2])) == (exp["n_scored"], exp["best_index"], exp["best_score"])
                out[key] = {"neighbours_per_s": R2 * mc2 * (n2 - 1) / s, "ms_per_launch": s * 1e3,
                            "pair_checks_per_s": R2 * mc2 * (n2 - 1) / s * (n2 - 1), "columns": mc2, "verified_restart_0": bool(ok)}
                del rows2, cur2_d, cols2, nc2, res2
            except Exception as e:  # noqa: BLE001 - a secondary line must not take the bench down
                out[key] = {"error": repr(e)}
    # the all-integer formulation of the scan kernel (BoardPk16: packed u16 xor / min / add, LS_NQ_SCAN_VARIANT=1) on the same
    # launch: the same results (parity-tested), half-rate instructions -- the fp32 product test is a speed choice, not a precision one
    try:
        import ls_amd
        old = os.environ.get("LS_NQ_SCAN_VARIANT")
        os.environ["LS_NQ_SCAN_VARIANT"] = "1"
        try:
            with ls_amd.Context(ctx.device, algo="scan") as ci:
                ci.set_stream(torch.cuda.current_stream().cuda_stream)
                res_i = torch.zeros((R, 4), dtype=torch.int64, device=dev)
                s = timed(lambda: ci.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_full.data_ptr(),
                                                            ncols_full.data_ptr(), n, n * (n - 1), res_i.data_ptr()), 3)
                torch.cuda.synchronize()
                ctx.set_algo("scan")
                ctx.nqueens_eval_batch_dev(R, n, rows_d.data_ptr(), cur_d.data_ptr(), cols_full.data_ptr(), ncols_full.data_ptr(), n,
                                           n * (n - 1), res_d.data_ptr())
                torch.cuda.synchronize()
                out["scan_int_formulation_packed_u16"] = {"neighbours_per_s": R * n * (n - 1) / s, "ms_per_launch": s * 1e3, "dtype": "u16 (packed pairs)",
                                                          "same_result_records_as_fp32_formulation": bool(torch.equal(res_i, res_d))}
                ci.set_stream(None)
        finally:
            if old is None:
                os.environ.pop("LS_NQ_SCAN_VARIANT", None)
            else:
                os.environ["LS_NQ_SCAN_VARIANT"] = old
    except Exception as e:  # noqa: BLE001 - a secondary line must not take the bench down
        out["scan_int_formulation_packed_u16"] = {"error": repr(e)}
    # the same full-neighbourhood workload through the HOST-pointer entry (inputs cross PCIe every call,
    # one pinned H2D + one D2H, stream sync): never `value`, recorded for DESIGN.md section 6
    ctx.set_algo("scan")
    cols_np = np.tile(np.arange(n, dtype=np.int32), (R, 1))
    ncols_np = np.full(R, n, dtype=np.int32)
    ctx.nqueens_eval_batch(boards, cur_np, cols_np, ncols_np, n * (n - 1))
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ctx.nqueens_eval_batch(boards, cur_np, cols_np, ncols_np, n * (n - 1))
    s = (time.perf_counter() - t0) / reps
    out["scan_full_neighbourhood_host_pointer_api_pcie_inclusive"] = {
        "neighbours_per_s": R * n * (n - 1) / s, "ms_per_call": s * 1e3,
        "host_bytes_per_call": int(boards.nbytes + cur_np.nbytes + cols_np.nbytes + ncols_np.nbytes + 32 * R)}
    # the solver's per-iteration call (config 4 shape): fused LS step -- neighbourhood argmin + move + winner's column
    # scores -- through the HOST-pointer entry, i.e. a complete round trip incl. PCIe and the stream sync
    ctx.set_algo("auto")
    cols_h, ncols_h = cols_ref.cpu().numpy(), ncols_ref.cpu().numpy()
    for RR in (1, R):
        # the solver's steady state: every call continues from the previous winner (board, count tables and column
        # scores stay resident on the device / host); the board update below is part of the timed loop
        b_h, c_h = boards[:RR].copy(), np.array(cur_np[:RR]).copy()

        def step():
            out_h, _ = ctx.nqueens_eval_batch_next(b_h, c_h, cols_h[:RR], ncols_h[:RR], 5 * n)
            for rr in range(RR):
                if out_h[rr]["best_index"] >= 0:
                    b_h[rr, out_h[rr]["best_col"]] = out_h[rr]["best_value"]
                    c_h[rr] = out_h[rr]["best_score"]

        for _ in range(200):     # also lets the clocks settle after the heavy launches above
            step()
        reps = 1000
        requests_before = ctx.counters()["server_requests"]
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        s = (time.perf_counter() - t0) / reps
        on_server = ctx.counters()["server_requests"] - requests_before >= reps * RR
        c = kc.get(f"nq_step_kernel_r{RR}") or {}
        raw_us = None
        if True:
            # the same loop on the bare C entry (arguments marshalled once): what a C / Rust caller pays per call
            import ctypes as C
            import ls_amd as _la
            rows1, cur1 = np.ascontiguousarray(b_h[:RR], np.int32), np.ascontiguousarray(c_h[:RR], np.int64)
            cols1, nc1 = np.ascontiguousarray(cols_h[:RR], np.int32), np.ascontiguousarray(ncols_h[:RR], np.int32)
            res1, nxt1 = (_la.NqResult * RR)(), np.empty((RR, n), dtype=np.int64)
            vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            cargs = (ctx._h, RR, n, vp(rows1), vp(cur1), vp(cols1), vp(nc1), int(cols1.shape[1]), 5 * n, C.cast(res1, C.c_void_p), vp(nxt1))
            fn = ctx._l.ls_nqueens_eval_batch_next

            def raw_step():
                if fn(*cargs) != 0:
                    raise RuntimeError(ctx._l.ls_last_error(ctx._h).decode())
                for rr in range(RR):
                    r = res1[rr]
                    if r.best_index >= 0:
                        rows1[rr, r.best_col] = r.best_value
                        cur1[rr] = r.best_score

            for _ in range(200):
                raw_step()
            t0 = time.perf_counter()
            for _ in range(5000 // RR):
                raw_step()
            raw_us = (time.perf_counter() - t0) / (5000 // RR) * 1e6
        out[f"fused_ls_step_host_round_trip_{RR}_restarts"] = {
            "resident": resident_route("nq_server_kernel", raw_us) if on_server else None,
            "us_per_call": s * 1e6, "candidates_per_call": RR * min(5 * n, mc * (n - 1)),
            "us_per_call_bare_c_entry": raw_us,
            "note": "us_per_call includes ctypes/numpy marshalling of the Python wrapper (about 8 us); us_per_call_bare_c_entry calls the C entry "
                    "with arguments marshalled once (still a Python loop: about 1 us of interpreter per call and restart)",
            "route": ("resident LS-step server (no launch per call" + ("" if RR == 1 else f": one slot per restart, {RR} requests in flight") + ")") if on_server
                     else "one fused LS-step launch per call",
            # a latency-bound launch: one 1024-thread block per restart; its issue bounds on the CUs it can occupy are a small
            # fraction of the kernel time, which is PCIe-free set-up, LDS round trips and the completion word
            ("launched_kernel_of_the_launch_per_call_route_LS_NQ_STEP_3" if on_server else "kernel"):
                {k: c.get(k) for k in ("duration_us", "valu_issue_bound_us", "lds_array_bound_us", "cus_usable", "roofline", "source")} if c else None}
    if e2e:
        out["config4_end_to_end_64_restarts_one_gpu"] = config4_end_to_end(ctx, n)
        out.update(single_restart_end_to_end(ctx))
        out.update(multi_restart_sched_end_to_end(ctx, out))
        for key, route in (("config4_end_to_end_64_restarts_one_gpu", "nq_server_kernel"), ("config2_end_to_end_nq1024_seed42", "nq_server_kernel"),
                           ("config3_end_to_end_sched_example_seed42", "sched_server_kernel"), ("config3_multi_restart_end_to_end", "sched_server_kernel"),
                           ("config5_end_to_end_sched10k_window65536_seed42", "sched_grid_server_kernel"),
                           ("config5_multi_restart_end_to_end", "sched_grid_server_kernel")):
            if isinstance(out.get(key), dict) and "error" not in out[key]:
                out[key]["resident"] = resident_route(route, out[key].get("us_per_ls_iteration"))
                if key == "config2_end_to_end_nq1024_seed42" and out[key]["resident"]:
                    out[key]["resident"]["note"] = "timeline measured at n = 4096 (config 4's board size); the same kernel serves n = 1024"
                if key == "config5_multi_restart_end_to_end" and out[key]["resident"]:
                    out[key]["resident"]["note"] = "timeline of the one-sub-grid launch (one chunk per workgroup); with 8 sub-grids a workgroup scores six chunks per request"
    ctx.set_algo("scan")
    out.update(sched_measurements(ctx, torch, dev, timed, issue_roofline, cpu_seconds=max(sched_cpu_seconds, 0.05)))
    # the HBM figure the rooflines divide by is the vendor's 8 TB/s; this is what a plain device-to-device
    # copy of 2 GiB reaches on this box (torch's copy kernel; bytes read + bytes written per second)
    try:
        src = torch.empty(1 << 29, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        s = timed(lambda: dst.copy_(src), 10)
        out["hbm_copy_bandwidth"] = {"GBps_read_plus_write": 2 * src.numel() * 4 / s / 1e9, "bytes_copied": int(src.numel() * 4)}
        del src, dst
    except Exception as e:  # noqa: BLE001 - a measurement line must not take the bench down
        out["hbm_copy_bandwidth"] = {"error": repr(e)}
    return out


def config4_end_to_end(ctx, n):
    """BASELINE config 4 end to end on ONE GPU: the 64 seeds "42".."105" (examples/nqueens/src/main.rs:167 convention)
    stream through ls_multi_solver's slots; every restart is the reference's complete ILS driver on the host (proposer,
    acceptance, perturbation; as fibres over the host cores), every score and neighbourhood on the GPU -- since round 3
    through the resident LS-step server (one never-relaunched kernel, one workgroup per slot, no launch per iteration).
    Host- and latency-inclusive: what a user of the solver sees, not the kernel-throughput `value`.
    Parity of the TIMED run (after the timed region): four sampled restarts -- the longest one among them -- are run again
    alone through ls_solver on the launch-per-call route (LS_NQ_STEP=3, a different code path: nq_step_kernel<true>) and
    must give the same best solution and the same number of LS iterations ("same seed => same result",
    examples/nqueens/src/main.rs:156-200); the first eight best boards are re-scored by the oracle's literal O(n^2) loop."""
    import ls_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    params = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32,
                  all_solutions_capacity=100_000, all_solution_iteration_expiry=10_000,
                  iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)   # examples/nqueens/src/main.rs:129-135
    seeds = [str(42 + r) for r in range(64)]
    ctx.set_algo("auto")
    try:
        m = ls_amd.MultiSolver(ctx, n, seeds, params)
        t0 = time.perf_counter()
        m.run(max_rounds_per_restart=10_000, max_seconds=120.0)
        dt = time.perf_counter() - t0
        st, cnt = m.stats(), m.counters()
        bests = [m.best(r) for r in range(len(seeds))]
        per = [m.restart_stats(r) for r in range(len(seeds))]
        m.close()
        solved = sum(1 for b in bests if b is not None and b[0] == 0)
        checked = all(b is not None and orc.nq_score(np.array(b[1])) == b[0] for b in bests[:8])   # literal O(n^2) rescoring
        longest = int(np.argmax([p["ls_iterations"] for p in per]))
        sample = sorted({longest, 0, 21, 42})
        compared, same = [], True
        old = os.environ.get("LS_NQ_STEP")
        os.environ["LS_NQ_STEP"] = "3"
        try:
            with ls_amd.Context(ctx.device, algo="auto") as c3:
                for r in sample:
                    s1 = ls_amd.Solver.nqueens(c3, n, seeds[r], params)
                    rounds = 0
                    while not s1.is_finished():
                        s1.execute_round()
                        rounds += 1
                        b = s1.best()
                        if b is not None and b[0] == 0:
                            break
                    ok = s1.best() == bests[r] and s1.stats()["ls_iterations"] == per[r]["ls_iterations"] and rounds == per[r]["rounds"]
                    compared.append({"seed": seeds[r], "ls_iterations": per[r]["ls_iterations"], "rounds": per[r]["rounds"], "equal": bool(ok)})
                    same = same and ok
                    s1.close()
        finally:
            if old is None:
                os.environ.pop("LS_NQ_STEP", None)
            else:
                os.environ["LS_NQ_STEP"] = old
        out = {"seconds": dt, "restarts": len(seeds), "solved": solved, "neighbours_scored": st["neighbours_scored"],
               "neighbours_per_s": st["neighbours_scored"] / dt, "ls_iterations": st["ls_iterations"],
               "ls_iterations_per_s": st["ls_iterations"] / dt, "gpu_requests": st["gpu_launches"],
               "host_cores": host_core_share(), "counters": cnt,
               "best_scores_rechecked_by_oracle": checked,
               "timed_run_vs_single_solver_runs_on_the_launch_per_call_route": {"equal": bool(same), "restarts_compared": compared},
               "mode": "free-running slots = workgroups of ONE resident LS-step server launch (board + count tables in LDS, a 64-byte "
                       "request and a 32-byte answer per LS iteration), slots beyond the host cores as fibres"
                       if cnt["server_requests"] else "free-running slots, one fused LS-step launch per iteration"}
        if not same or not checked:
            out["error"] = "the timed end-to-end run differs from the single-solver runs"
        return out
    except Exception as e:  # noqa: BLE001 - a secondary line must not take the bench down
        return {"error": repr(e)}


def single_restart_end_to_end(ctx):
    """BASELINE configs 2, 3 and 5 end to end on one GPU (one restart each, seed "42", the reference's hyper-parameters:
    examples/nqueens/src/main.rs:129-135, examples/employee-scheduling/src/main.rs:11-31): the reference's ILS driver on
    the host, every score and neighbourhood on the GPU.  Latency-bound: one launch per LS iteration."""
    import ls_amd

    def run(s, max_rounds):
        t0 = time.perf_counter()
        rounds = 0
        while not s.is_finished() and rounds < max_rounds:
            s.execute_round()
            rounds += 1
            b = s.best()
            if b is not None and (b[0] == 0 or b[0] == (0, 0)):
                break
        dt = time.perf_counter() - t0
        st, b = s.stats(), s.best()
        s.close()
        return {"seconds": dt, "rounds": rounds, "best_score": None if b is None else b[0], "ls_iterations": st["ls_iterations"],
                "neighbours_scored": st["neighbours_scored"], "neighbours_per_s": st["neighbours_scored"] / dt,
                "us_per_ls_iteration": dt / max(1, st["ls_iterations"]) * 1e6}

    out = {}
    try:
        ctx.set_algo("auto")
        nq = dict(local_search_max_iterations=10_000, best_solutions_capacity=32, all_solutions_capacity=100_000,
                  all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
        sched = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
                     all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)
        out["config2_end_to_end_nq1024_seed42"] = run(ls_amd.Solver.nqueens(ctx, 1024, "42", dict(nq, window_size=5 * 1024)), 10_000)
        out["config3_end_to_end_sched_example_seed42"] = run(ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", sched), 250)
        rng = np.random.default_rng(5)
        bits = np.zeros(10_000, dtype=np.uint64)
        for e in range(10_000):
            for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
                bits[e] |= np.uint64(1) << np.uint64(d)
        out["config5_end_to_end_sched10k_window65536_seed42"] = run(
            ls_amd.Solver.scheduling(ctx, 31, 0, 10_000, "42", dict(sched, window_size=65_536), holiday_bits=bits), 250)
    except Exception as e:  # noqa: BLE001 - a secondary line must not take the bench down
        out["single_restart_end_to_end_error"] = repr(e)
    return out


def multi_restart_sched_end_to_end(ctx, so_far):
    """Many scheduling restarts on ONE GPU through ls_multi_solver_create_sched (driver per restart:
    examples/employee-scheduling/src/lib.rs:57-117, seeds by examples/nqueens/src/main.rs:167): config 3 x 64 restarts as the
    workgroups of one resident LS-step launch, config 5 x 8 restarts as the sub-grids of one resident grid launch.  After the
    timed runs, sampled restarts are run again alone through ls_solver and must give the same best solution and LS-iteration
    count; config 5's wall is compared with one restart alone (from the single-restart leg)."""
    import ls_amd
    sched = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
                 all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)
    out = {}

    def single(args, kw, seed):
        s = ls_amd.Solver.scheduling(ctx, *args, seed, kw.get("params"), holiday_bits=kw.get("bits"))
        rounds = 0
        while not s.is_finished() and rounds < 250:
            s.execute_round()
            rounds += 1
        r = (s.best(), s.stats()["ls_iterations"])
        s.close()
        return r

    def leg(name, args, params, bits, seeds, sample, n_slots=None):
        try:
            m = ls_amd.MultiSolver.scheduling(ctx, *args, seeds, params, holiday_bits=bits, n_slots=n_slots)
            t0 = time.perf_counter()
            m.run(max_rounds_per_restart=250, max_seconds=120.0)
            dt = time.perf_counter() - t0
            st, cnt = m.stats(), m.counters()
            got = {r: (m.best(r), m.restart_stats(r)["ls_iterations"]) for r in sample}
            scores = sorted(m.best(r)[0] for r in range(len(seeds)) if m.best(r) is not None)
            m.close()
            same = all(got[r] == single(args, dict(params=params, bits=bits), seeds[r]) for r in sample)
            out[name] = {"seconds": dt, "restarts": len(seeds), "neighbours_scored": st["neighbours_scored"],
                         "neighbours_per_s": st["neighbours_scored"] / dt, "ls_iterations": st["ls_iterations"],
                         "ls_iterations_per_s": st["ls_iterations"] / dt, "us_per_ls_iteration_per_restart": dt / max(1, st["ls_iterations"]) * len(seeds) * 1e6,
                         "best_scores": [list(x) for x in scores[:4]], "host_cores": host_core_share(), "counters": cnt,
                         "equal_single_solver_runs": bool(same), "restarts_compared": [seeds[r] for r in sample]}
            if not same:
                out[name]["error"] = "a restart of the timed multi-restart run differs from its single-solver run"
        except Exception as e:  # noqa: BLE001 - reported through the line; the caller fails the bench on it
            out[name] = {"error": repr(e)}

    leg("config3_multi_restart_end_to_end", (31, 0, 7), sched, None, [str(42 + r) for r in range(64)], (0, 21, 63))
    E = 10_000
    rng = np.random.default_rng(5)
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(d)
    leg("config5_multi_restart_end_to_end", (31, 0, E), dict(sched, window_size=65_536), bits, [str(42 + r) for r in range(8)], (0, 5), n_slots=8)
    one = (so_far.get("config5_end_to_end_sched10k_window65536_seed42") or {}).get("seconds")
    if one and "seconds" in out["config5_multi_restart_end_to_end"]:
        out["config5_multi_restart_end_to_end"]["single_restart_seconds"] = one
        out["config5_multi_restart_end_to_end"]["wall_over_single_restart"] = out["config5_multi_restart_end_to_end"]["seconds"] / one
    return out


# ------------------------------------------------------------------------------------------------ scheduling workloads
SCHED_D, SCHED_S = 31, 32          # examples/employee-scheduling/src/main.rs:11-12: 31 scored days, 32 stored (lib.rs:407-411)


def sched_random_moves(rng, n, D, E):
    """A list shaped like ScheduleRandomMoveProposer's stream (lib.rs:444-490): ChangeDay : SwapDays = 1 : 4, a swap's two
    days distinct.  (Synthetic: the real stream comes from the host driver's generator; the kernels do not care.)"""
    mv = np.zeros((n, 3), dtype=np.int32)
    change = rng.integers(0, 5, size=n) == 0
    a = rng.integers(0, D, size=n)
    mv[:, 0] = np.where(change, 0, 1)
    mv[:, 1] = a
    mv[:, 2] = np.where(change, rng.integers(0, E, size=n), (a + 1 + rng.integers(0, D - 1, size=n)) % D)
    return mv


def sched_holidays_10k(E=10_000, D=SCHED_D, seed=2022):
    """every employee gets h ~ U{0..3} distinct holidays, uniform over the scored days."""
    rng = np.random.default_rng(seed)
    h = rng.integers(0, 4, size=E)
    days = np.argsort(rng.random((E, D)), axis=1)[:, :3]      # three distinct days per employee, the first h[e] are used
    bits = np.zeros(E, dtype=np.uint64)
    for k in range(3):
        bits |= np.where(h > k, np.uint64(1) << days[:, k].astype(np.uint64), np.uint64(0))
    return bits


def sched_oracle_instance(bits=None):
    """The oracle's view of an instance: the example's dates (2022-05-09 .. 2022-06-08) + holidays from the bit masks."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    hol = {}
    if bits is not None:
        for e in np.nonzero(bits)[0]:
            hol[int(e)] = [d for d in range(SCHED_D) if (int(bits[e]) >> d) & 1]
    return orc.Sched((2022, 5, 9), (2022, 6, 8), hol)


def _sched_apply(state, m):
    c = [int(x) for x in state]
    if m[0] == 0:
        c[m[1]] = int(m[2])
    else:
        c[m[1]], c[m[2]] = int(state[m[2]]), int(state[m[1]])
    return tuple(c)


def sched_oracle_record(inst, state, moves, window, threads=1):
    """local_search.rs:315-325 by the oracle's literal pipeline (clone + tabu test + full get_scored_solution per candidate,
    minimum by (hard, soft, vector), first of equals) -> (n_scored, best_index, hard, soft).  The list prefix take(window)
    reaches is cut into `threads` slices scored in parallel (ctypes releases the GIL); the slices' minima are combined in
    the same order.  Checker only: never the thing measured."""
    from concurrent.futures import ThreadPoolExecutor

    st = np.asarray(state, dtype=np.int64)
    mv = np.ascontiguousarray(np.asarray(moves, dtype=np.int64).reshape(-1, 3))
    if len(mv) == 0 or window <= 0:
        return (0, -1, 0, 0)
    real = np.where(mv[:, 0] == 0, st[mv[:, 1]] != mv[:, 2], st[mv[:, 1]] != st[np.minimum(mv[:, 2], len(st) - 1)])
    cum = np.cumsum(real)
    n_scored = int(min(cum[-1], window))
    if n_scored == 0:
        return (0, -1, 0, 0)
    k_end = int(np.searchsorted(cum, n_scored)) + 1
    bounds = np.linspace(0, k_end, max(1, threads) + 1).astype(int)

    def one(t):
        lo, hi = bounds[t], bounds[t + 1]
        if hi <= lo:
            return None
        r = inst.neighbourhood(st, mv[lo:hi], 1 << 62, want_scores=False)
        if r["best_index"] < 0:
            return None
        k = lo + r["best_index"]
        return (r["best_hard"], r["best_soft"], _sched_apply(st, mv[k]), k)

    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        parts = [p for p in ex.map(one, range(len(bounds) - 1)) if p is not None]
    h, so, _, k = min(parts)
    return (n_scored, int(k), int(h), int(so))


def sched_cpu_rate(inst, state, moves, window, seconds, threads):
    """Candidates / second of the oracle's literal scheduling pipeline (the reference's work per candidate,
    examples/employee-scheduling/src/lib.rs:265-374) on a bounded sample of one workload: the first candidates of the SAME
    move list the GPU leg times, on one core, and -- one disjoint slice of the list per thread -- on `threads` cores."""
    from concurrent.futures import ThreadPoolExecutor

    st = np.asarray(state, dtype=np.int64)
    mv = np.ascontiguousarray(np.asarray(moves, dtype=np.int64).reshape(-1, 3))
    t0 = time.perf_counter()
    r = inst.neighbourhood(st, mv[:256], 1 << 62, want_scores=False)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate0 = max(r["n_scored"], 1) / dt
    sample = int(max(64, min(min(window, len(mv)), rate0 * seconds)))
    t0 = time.perf_counter()
    r = inst.neighbourhood(st, mv[:sample + sample // 8], sample, want_scores=False)
    dt1 = time.perf_counter() - t0
    one = r["n_scored"] / max(dt1, 1e-9)
    per = max(64, min(len(mv) // max(1, threads), sample))

    def leg(t):
        return inst.neighbourhood(st, mv[t * per:(t + 1) * per], 1 << 62, want_scores=False)["n_scored"]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        tot = sum(ex.map(leg, range(threads)))
    dtn = time.perf_counter() - t0
    return {"one_core_per_s": one, "one_core_sample": int(r["n_scored"]), "one_core_seconds": dt1,
            "all_cores_per_s": tot / max(dtn, 1e-9), "cores": threads, "all_cores_sample": int(tot), "all_cores_seconds": dtn}


def sched_rank_line(args, ctx, torch, dev, stream, rank, world, sync):
    """The scheduling line every run carries, also at --gpus N > 1 (10 000 employees x 31 days, ONE
    restart per GPU, the random proposer's stream with window 65 536): kernel throughput with state and move list resident
    in HBM, barrier + synchronise on both sides, MAX over ranks; each rank's timed result record is then re-checked against
    the oracle's literal pipeline.  Called by ALL ranks (it has barriers)."""
    E, n_moves, window, steps, warm = 10_000, 70_000, 65_536, 50, 5
    bits = sched_holidays_10k(E)
    rid = rank                                     # global restart id: one restart per rank
    rng = np.random.default_rng(7000 + rid)
    state_np = rng.integers(0, E, size=(1, SCHED_S)).astype(np.int32)
    moves_np = sched_random_moves(rng, n_moves, SCHED_D, E)
    ctx.sched_instance_upload(SCHED_D, SCHED_S, 0, E, bits)
    with torch.cuda.stream(stream):
        states = torch.from_numpy(state_np).to(dev)
        mv = torch.from_numpy(moves_np[None]).to(dev)
        nm = torch.full((1,), n_moves, dtype=torch.int64, device=dev)
        res = torch.zeros((1, 4), dtype=torch.int64, device=dev)

        def step():
            ctx.sched_eval_batch_dev(1, states.data_ptr(), mv.data_ptr(), nm.data_ptr(), n_moves, window, res.data_ptr())

        for _ in range(warm):
            step()
        torch.cuda.synchronize(dev)
        sync.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        sync.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        rec = tuple(int(x) for x in res.cpu().numpy()[0])
    t0 = time.perf_counter()
    exp = sched_oracle_record(sched_oracle_instance(bits), state_np[0], moves_np, window, threads=host_core_share())
    t_check = time.perf_counter() - t0
    gathered = sync.all_gather(rank, dict(elapsed=elapsed, n_scored=rec[0], ok=(rec == exp), rid=rid, got=rec, want=exp))
    if rank != 0:
        return None
    elapsed = max(g["elapsed"] for g in gathered)
    total = sum(g["n_scored"] for g in gathered)
    bad = [g["rid"] for g in gathered if not g["ok"]]
    return {"metric": "scheduling neighbour evals/sec (10 000 employees x 31 days, one restart per GPU, random stream, window 65 536)",
            "value": total * steps / elapsed, "unit": "neighbour evals/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "us_per_step": elapsed / steps * 1e6, "scaling": "weak", "restarts": world, "candidates_per_step_per_gpu": gathered[0]["n_scored"],
            "effective_GBps_at_144B": total * steps * 144 / elapsed / 1e9,
            "verified": {"ok": not bad, "restarts_checked": len(gathered), "mismatching_restarts": bad, "seconds": t_check,
                         "against": "oracle/sched_oracle.c literal pipeline, (n_scored, best_index, hard, soft) per restart",
                         "first_mismatch": None if not bad else next(dict(got=g["got"], want=g["want"]) for g in gathered if not g["ok"])}}


def sched_measurements(ctx, torch, dev, timed, issue_roofline, cpu_seconds=0.4):
    """Scheduling path: kernel throughput with moves and states resident in HBM; every timed
    launch's result records re-checked against the oracle AFTER the timed region; the oracle's literal pipeline timed on the
    same move lists beside it (one core, all host cores: a reported baseline, not a target)."""
    out = {}
    rng = np.random.default_rng(2022)
    cores = host_core_share()
    verified, cpu = {}, {}

    # config 3 shape: example instance (7 employees, 31/32 days), window 100, 64 restarts per launch
    D, S, E, R, W = SCHED_D, SCHED_S, 7, 64, 100
    ctx.sched_instance_upload(D, S, 0, E)
    inst7 = sched_oracle_instance(None)
    st7 = rng.integers(0, E, size=(R, S)).astype(np.int32)
    mv7 = np.stack([sched_random_moves(rng, 160, D, E) for _ in range(R)])
    states = torch.from_numpy(st7).to(dev)
    mv = torch.from_numpy(mv7).to(dev)
    nm = torch.full((R,), 160, dtype=torch.int64, device=dev)
    res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    s = timed(lambda: ctx.sched_eval_batch_dev(R, states.data_ptr(), mv.data_ptr(), nm.data_ptr(), 160, W, res.data_ptr()), 200)
    torch.cuda.synchronize()
    got = res.cpu().numpy()
    bad = [r for r in range(R) if tuple(int(x) for x in got[r]) != sched_oracle_record(inst7, st7[r], mv7[r], W)]
    verified["sched_example_window100_x64_restarts"] = {"ok": not bad, "records_checked": R, "mismatching": bad}
    cpu["sched_example_window100"] = sched_cpu_rate(inst7, st7[0], np.concatenate(list(mv7)), 10 ** 9, cpu_seconds, cores)
    out["sched_example_window100_x64_restarts"] = {"neighbours_per_s": R * W / s, "us_per_launch": s * 1e6,
                                                   "candidates_per_launch": R * W,
                                                   "cpu_one_core_per_s": cpu["sched_example_window100"]["one_core_per_s"]}
    # config 5 shape: 10 000 employees, holidays U{0..3} per employee, one restart per GPU
    E = 10_000
    bits = sched_holidays_10k(E)
    inst10k = sched_oracle_instance(bits)
    ctx.sched_instance_upload(D, S, 0, E, bits)
    state_np = rng.integers(0, E, size=(1, S)).astype(np.int32)
    states = torch.from_numpy(state_np).to(dev)
    ex = ctx.sched_exhaustive_moves(state_np)[0]       # exhaustive ScheduleMoveProposer order (lib.rs:493-559), built on the device
    res = torch.zeros((1, 4), dtype=torch.int64, device=dev)
    for name, moves, window in (("sched10k_exhaustive_320k", ex, 10 ** 9),
                                ("sched10k_random_stream_window_65536", sched_random_moves(rng, 70_000, D, E), 65_536)):
        mv = torch.from_numpy(moves[None]).to(dev)
        nm = torch.full((1,), len(moves), dtype=torch.int64, device=dev)
        s = timed(lambda: ctx.sched_eval_batch_dev(1, states.data_ptr(), mv.data_ptr(), nm.data_ptr(), len(moves), window,
                                                   res.data_ptr()), 50)
        torch.cuda.synchronize()
        rec = tuple(int(x) for x in res.cpu().numpy()[0])
        t0 = time.perf_counter()
        exp = sched_oracle_record(inst10k, state_np[0], moves, window, threads=cores)
        verified[name] = {"ok": rec == exp, "records_checked": 1, "oracle_seconds": time.perf_counter() - t0,
                          **({} if rec == exp else {"got": rec, "want": exp})}
        cpu[name] = sched_cpu_rate(inst10k, state_np[0], moves, window, cpu_seconds, cores)
        n_scored = rec[0]
        out[name] = {"neighbours_per_s": n_scored / s, "us_per_launch": s * 1e6, "candidates_per_launch": n_scored,
                     "effective_GBps_at_144B": n_scored * 144 / s / 1e9,
                     "cpu_one_core_per_s": cpu[name]["one_core_per_s"], "cpu_all_cores_per_s": cpu[name]["all_cores_per_s"],
                     "roofline_eval_kernel_bound_over_whole_chain_time": issue_roofline("sched_eval_all_kernel_exhaustive_list" if "exhaustive" in name
                                                                 else "sched_eval_all_kernel_random65536", s)}
    # the same exhaustive neighbourhood through the host entry that generates the move list on the device:
    # only the 128-byte state and the 32-byte result cross PCIe (round trip incl. sync and the Python wrapper)
    before = ctx.counters()["server_requests"]
    for _ in range(20):
        ctx.sched_eval_exhaustive(state_np, 10 ** 9)
    t0 = time.perf_counter()
    reps = 500
    for _ in range(reps):
        r0 = ctx.sched_eval_exhaustive(state_np, 10 ** 9)[0]
    s = (time.perf_counter() - t0) / reps
    on_grid = ctx.counters()["server_requests"] - before >= reps
    want = sched_oracle_record(inst10k, state_np[0], ex, 10 ** 9, threads=cores)
    got_rt = (r0["n_scored"], r0["best_index"], r0["best_hard"], r0["best_soft"])
    verified["sched10k_exhaustive_320k_host_round_trip"] = {"ok": got_rt == want, "records_checked": 1}
    out["sched10k_exhaustive_320k_host_round_trip_generated_on_device"] = {
        "resident": resident_route("sched_grid_server_kernel_gen", s * 1e6) if on_grid else None,
        "neighbours_per_s": r0["n_scored"] / s, "us_per_call": s * 1e6, "candidates_per_call": r0["n_scored"],
        "route": "resident grid (no launch per call; candidates generated in the kernel, one workgroup per three 512-candidate chunks)" if on_grid
                 else "two launches per call (sched_eval_all_kernel + sched_finish_kernel)",
        "note": "includes about 4 us of ctypes / numpy marshalling in the Python wrapper (tools/exhaustive_grid_trace.py times the bare C entry)"}
    out["sched_verified"] = {"ok": all(v["ok"] for v in verified.values()), "legs": verified,
                             "against": "oracle/sched_oracle.c: literal clone + tabu test + get_scored_solution per candidate, after the timed regions"}
    out["sched_cpu_baseline"] = {"kind": "port", "unit": "neighbour evals/s", "legs": cpu,
                                 "note": "the oracle's literal scheduling pipeline (examples/employee-scheduling/src/lib.rs:265-374 per candidate) on "
                                         "bounded samples of the same move lists; the reference is single-threaded Rust, unbuildable here"}
    return out


if __name__ == "__main__":
    main()
