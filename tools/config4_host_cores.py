#!/usr/bin/env python3
"""Multi-GPU readiness that can be measured on ONE GPU: BASELINE config 4 end to end (n = 4096, the
64 seeds "42".."105", examples/nqueens/src/main.rs:167) with the process restricted to 2 / 4 / 8 / 16 host cores
(os.sched_setaffinity in the child BEFORE anything touches the GPU) -- the host-core share a rank has when 8 ranks share
one node's CPUs.  Prints one JSON line per core count: the slot count the library chose (four per usable core), wall time,
neighbours/s, LS iterations/s, counters.  No 1 -> 8 GPU curve is measured by this: restarts are independent, a rank's
throughput depends on its own host cores and its own GPU only.

    python tools/config4_host_cores.py [--restarts 64] [--cores 2,4,8,16] [--step-mode 4]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--restarts", type=int, default=64)
    ap.add_argument("--cores", default="2,4,8,16")
    ap.add_argument("--step-mode", default=None, help="LS_NQ_STEP for the child (4 = resident server, 3 = launch per call)")
    args = ap.parse_args()
    avail = sorted(os.sched_getaffinity(0))
    for k in [int(x) for x in args.cores.split(",")]:
        if k > len(avail):
            print(json.dumps({"host_cores": k, "skipped": f"only {len(avail)} cores available"}), flush=True)
            continue
        cores = set(avail[:k])
        env = dict(os.environ)
        if args.step_mode is not None:
            env["LS_NQ_STEP"] = args.step_mode
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_config4.py"), "--restarts-per-gpu", str(args.restarts)],
                           capture_output=True, text=True, timeout=150, env=env, preexec_fn=lambda: os.sched_setaffinity(0, cores))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(json.dumps({"host_cores": k, "error": r.stderr[-500:]}), flush=True)
            continue
        d = json.loads(line[-1])
        print(json.dumps({"host_cores": k, "slots_chosen": min(4 * k, args.restarts), "seconds": d["seconds"],
                          "neighbours_per_s": d["neighbours_per_s"], "ls_iterations_per_s": d["ls_iterations_per_s"],
                          "solved": sum(1 for s in d["scores"] if s == 0), "restarts": d["restarts"],
                          "step_mode": args.step_mode or "default (4)", "counters": d.get("rank0_counters")}), flush=True)


if __name__ == "__main__":
    main()
