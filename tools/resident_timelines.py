#!/usr/bin/env python3
"""Timelines of the RESIDENT routes (no launch per LS iteration) from a -DLS_NQ_STAMPS build of the library: what a request costs
between "the workgroup sees it" and "the answer line is written", phase by phase (in-kernel wall-clock stamps, 100 MHz, returned in
spare dwords of the answer line), next to what the host sees (answer wait, whole call).  One entry per route:

    nq_server_kernel              config 4's LS step, one restart alone (n = 4096, window 5n)
    sched_server_kernel           config 3's LS step (example instance, window 100)
    sched_grid_server_kernel      config 5's LS step on the stream grid (10 000 employees, window 65 536)
    sched_grid_server_kernel_gen  the exhaustive proposer's 32 x 10 000 candidates on the generating grid

Each entry: {answer_wait_us, call_us, service_us (request seen -> answer written), phases_us, mailbox_us = answer wait - service,
bound_us = the VALU-issue bound of the same work measured on the LAUNCHED twin kernel (profiles/kernel_counters.json: instruction
counters per dispatch cannot be taken on a kernel that never ends), frac = bound_us / service_us}.

    python tools/resident_timelines.py build                       # here (hipcc cross-compiles): tools/_build/libls_amd_stamps.so
    python tools/resident_timelines.py [out.json]                  # on the GPU box -> profiles/<tag>_resident_timelines.json
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libls_amd_stamps.so")
SOURCES = ["ls_api.hip", "nq_kernels.hip", "nq_server.hip", "sched_wave.hip", "sched_delta.hip", "sched_server.hip", "sched_grid.hip", "ls_solver.hip"]

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%(root)r, "constraint-solver_amd"))
import numpy as np
import ls_amd
route = sys.argv[1]
sched = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
             all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)
def run(s, solved=lambda b: False):
    t0 = time.perf_counter()
    while not s.is_finished():
        s.execute_round()
        b = s.best()
        if b is not None and solved(b):
            break
    dt = time.perf_counter() - t0
    it = s.stats()["ls_iterations"]
    s.close()
    print("RESULT", it, dt, flush=True)
def big_instance():
    rng = np.random.default_rng(5)
    bits = np.zeros(10_000, dtype=np.uint64)
    for e in range(10_000):
        for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(d)
    return bits
with ls_amd.Context(0) as ctx:
    if route == "nq_server_kernel":
        n = 4096
        p = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32, all_solutions_capacity=100_000,
                 all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
        run(ls_amd.Solver.nqueens(ctx, n, "42", p), lambda b: b[0] == 0)
    elif route == "sched_server_kernel":
        run(ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", sched))
    elif route == "sched_grid_server_kernel":
        run(ls_amd.Solver.scheduling(ctx, 31, 0, 10_000, "42", dict(sched, window_size=65_536), holiday_bits=big_instance()))
    else:
        ctx.sched_instance_upload(31, 32, 0, 10_000, big_instance())
        st = np.random.default_rng(2022).integers(0, 10_000, size=(1, 32)).astype(np.int32)
        for _ in range(50):
            ctx.sched_eval_exhaustive(st, 10 ** 9)
        t0 = time.perf_counter()
        for _ in range(2000):
            ctx.sched_eval_exhaustive(st, 10 ** 9)
        print("RESULT", 2000, time.perf_counter() - t0, flush=True)
'''

TWINS = {  # route -> (launched twin in profiles/kernel_counters.json, what it is)
    "nq_server_kernel": ("nq_step_kernel_r1", "nq_step_kernel<true>, one restart, window 5n at n = 4096"),
    "sched_server_kernel": ("sched_eval_small_kernel_window100", "sched_eval_small_kernel, 512 resident moves, window 100"),
    "sched_grid_server_kernel": ("sched_eval_all_kernel_random65536", "sched_eval_all_kernel, stream of 70 000 moves, window 65 536"),
    "sched_grid_server_kernel_gen": ("sched_eval_all_kernel_exhaustive", "sched_eval_all_kernel<GEN>, 32 x 10 000 candidates"),
}


def floats_after(label_line):
    return [float(x) for x in re.findall(r"(?<![\w.])(\d+\.\d+)(?![\w.])", label_line)]


def parse(route, err):
    out = {}
    lines = err.splitlines()
    for i, ln in enumerate(lines):
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if route == "nq_server_kernel" and ln.startswith("ls step server"):
            m = re.search(r"host before ([\d.]+) us, answer wait ([\d.]+) us, host after ([\d.]+) us", ln)
            out.update(host_before_us=float(m.group(1)), answer_wait_us=float(m.group(2)), host_after_us=float(m.group(3)))
            out["call_us"] = out["host_before_us"] + out["answer_wait_us"] + out["host_after_us"]
            if "workgroup, after the request was seen" in nxt:
                f = floats_after(nxt)
                names = ["request published to the block", "slot records", "candidates scored", "block minimum", "answer written"]
                out["phases_us"] = dict(zip(names, f[:5]))
                out["service_us"] = f[4]
                m2 = re.search(r"; (\d+) shader cycles \(([\d.]+) GHz\)", nxt)
                if m2:
                    out["service_shader_cycles"], out["shader_clock_GHz"] = int(m2.group(1)), float(m2.group(2))
        if route == "sched_server_kernel" and ln.startswith("scheduling step server"):
            m = re.search(r"answer wait ([\d.]+) us per step \(whole call ([\d.]+) us\)", ln)
            if not m:
                continue  # (the line of the single-state scores)
            out.update(answer_wait_us=float(m.group(1)), call_us=float(m.group(2)))
            if "workgroup, after the request was seen" in nxt:
                f = floats_after(nxt)
                names = ["state + stream in", "table built", "candidates scored", "block minimum", "answer written"]
                out["phases_us"] = dict(zip(names, f[:5]))
                out["service_us"] = f[4]
        if route.startswith("sched_grid") and ln.startswith("scheduling grid"):
            m = re.search(r"(\d+) workgroups x (\d+) chunks\): (\d+) requests, answer wait ([\d.]+) us per request \(whole call ([\d.]+) us\)", ln)
            if not m:
                continue
            out.update(workgroups=int(m.group(1)), chunks_per_workgroup=int(m.group(2)), requests=int(m.group(3)),
                       answer_wait_us=float(m.group(4)), call_us=float(m.group(5)))
            if "the finisher, after it saw the request" in nxt:
                f = floats_after(nxt)
                names = ["edit + holiday bits in", "table built", "chunks scored", "own record out", "all records in", "window placed", "answer"]
                out["phases_us"] = dict(zip(names, f[:7]))
                out["service_us"] = f[6]
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        src = [os.path.join(ROOT, "constraint-solver_amd", "csrc", f) for f in SOURCES]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLS_NQ_STAMPS",
                               "-Wno-unused-function", "-o", SO] + src)
        print(SO)
        return
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "resident_timelines.json")
    counters = {}
    kc = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if os.path.exists(kc):
        with open(kc) as f:
            counters = json.load(f)
    result = {"source": "tools/resident_timelines.py on a -DLS_NQ_STAMPS build (in-kernel wall_clock64 stamps, 100 MHz, averaged over the run's requests); "
                        "the product build carries no stamps, its host-side times are in the plain entries",
              "routes": {}}
    for route in TWINS:
        ent = {}
        for build, so in (("stamps", SO), ("product", None)):
            env = dict(os.environ, LS_AMD_TRACE="1")
            if so:
                env["LS_AMD_LIB"] = so
            r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), route], capture_output=True, text=True, env=env, timeout=600)
            if r.returncode != 0:
                ent[build] = {"error": r.stderr[-1500:]}
                continue
            p = parse(route, r.stderr)
            m = re.search(r"RESULT (\d+) ([\d.eE+-]+)", r.stdout)
            if m:
                p["requests_timed"] = int(m.group(1))
                p["us_per_ls_iteration_or_call"] = float(m.group(2)) / int(m.group(1)) * 1e6
            ent[build] = p
        st = ent.get("stamps", {})
        twin, what = TWINS[route]
        c = counters.get(twin, {})
        if "service_us" in st:
            ent["service_us"] = st["service_us"]
            ent["phases_us"] = st.get("phases_us")
            pw = ent.get("product", {}).get("answer_wait_us", st.get("answer_wait_us"))
            ent["answer_wait_us"] = pw
            ent["call_us"] = ent.get("product", {}).get("call_us", st.get("call_us"))
            ent["mailbox_us"] = st["answer_wait_us"] - st["service_us"]  # same build: both crossings + the poll's detection delay
            if c.get("valu_issue_bound_us"):
                # the twin's bound is for its whole grid on the CUs it can occupy; a resident workgroup does one CU's share of it
                ent["bound_us"] = c["valu_issue_bound_us"]
                ent["bound"] = "valu_issue"
                ent["frac"] = c["valu_issue_bound_us"] / st["service_us"]
                ent["bound_from"] = f"profiles/kernel_counters.json[{twin}] ({what}; {c.get('source', 'rocprofv3 --pmc')})"
        result["routes"][route] = ent
        print(route, json.dumps({k: v for k, v in ent.items() if k not in ("stamps", "product")}), flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
