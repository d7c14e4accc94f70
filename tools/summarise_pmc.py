#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into the JSON files under
profiles/ (per-kernel mean / min / max KB per dispatch) and refresh profiles/hbm_traffic.json, which bench.py reads
for roofline.traffic.  Counter values are KB per dispatch as rocprofv3 reports them (x 1024 = bytes).
    python tools/summarise_pmc.py gpurun_out/prof_<tag> <tag> [n] [restarts]
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(prof_dir):
    out = {}
    for path in glob.glob(os.path.join(prof_dir, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = re.sub(r"\(.*$", "", row["Kernel_Name"])          # drop the argument list
                if not name.startswith(("lsamd::", "void lsamd::")):
                    continue
                out.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: {c: dict(dispatches=len(v), mean_KB=sum(v) / len(v), min_KB=min(v), max_KB=max(v)) for c, v in cs.items()}
            for k, cs in out.items()}


def main():
    prof_dir, tag = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    summary = collect(prof_dir)
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_counters_bench_scan_n{n}.json")
    with open(dst, "w") as f:
        json.dump(summary, f, indent=1)
    scan = next((v for k, v in summary.items() if "nq_scan_regs_kernel" in k or "nq_scan_lds_kernel" in k), None)
    if scan and "FETCH_SIZE" in scan and "WRITE_SIZE" in scan:
        fetch, write = scan["FETCH_SIZE"]["mean_KB"], scan["WRITE_SIZE"]["mean_KB"]
        traffic = {f"nq_scan_n{n}_r{R}": dict(
            bytes_per_launch=int(round((fetch + write) * 1024)), fetch_KB=fetch, write_KB=write,
            source=f"profiles/{os.path.basename(dst)} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; raw counter "
                   "KB x 1024; dword-per-lane loads are uncalibrated on gfx950 (<= 2x under-count on the read side per "
                   "MI355X_MICROARCH.md HBM section))")}
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
        print(json.dumps(traffic, indent=1))
    print("wrote", dst)


if __name__ == "__main__":
    main()
