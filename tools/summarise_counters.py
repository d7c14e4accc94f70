#!/usr/bin/env python3
"""Summarise tools/profile_counters.sh's rocprofv3 passes into profiles/<tag>_kernel_counters.json (every kernel of the
run, grouped by name and grid size: mean counter values per dispatch, mean duration from the un-instrumented trace pass)
and profiles/kernel_counters.json (one entry per workload of tools/pmc_workloads.py with per-candidate instruction counts
and the issue-rate bounds bench.py and DESIGN.md quote).
    python tools/summarise_counters.py gpurun_out/pmc_<tag> <tag>
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLOCK = 1024, 2.4e9          # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)
VALU_CYC, LDS_B128_CYC = 2, 4       # cycles per wave64 instruction per SIMD / per CU (LDS: 64 dwords per clock per CU)


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("lsamd::", "")


def groups_from_counters(prof_dir):
    g = {}
    for sub in ("insts", "cycles", "lds"):
        for path in glob.glob(os.path.join(prof_dir, sub, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if "lsamd::" not in row["Kernel_Name"]:
                        continue
                    key = (short(row["Kernel_Name"]), int(row["Grid_Size"]))
                    e = g.setdefault(key, {"counters": {}, "first_dispatch": {}, "vgpr": int(row["VGPR_Count"]),
                                           "lds_block_size": int(row["LDS_Block_Size"]), "workgroup_size": int(row["Workgroup_Size"])})
                    e["counters"].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    d = int(row["Dispatch_Id"])
                    e["first_dispatch"][sub] = min(d, e["first_dispatch"].get(sub, d))
    return g


def durations(prof_dir):
    d = {}
    for path in glob.glob(os.path.join(prof_dir, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "lsamd::" not in row["Kernel_Name"]:
                    continue
                grid = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])
                d.setdefault((short(row["Kernel_Name"]), grid), []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return d


def main():
    prof_dir, tag = sys.argv[1], sys.argv[2]
    g = groups_from_counters(prof_dir)
    dur = durations(prof_dir)
    raw = {}
    for (name, grid), e in sorted(g.items(), key=lambda kv: min(kv[1]["first_dispatch"].values())):
        ent = {"kernel": name, "grid_threads": grid, "workgroup_size": e["workgroup_size"], "vgprs": e["vgpr"],
               "lds_bytes": e["lds_block_size"], "order": min(e["first_dispatch"].values()),
               "counters_mean_per_dispatch": {c: sum(v) / len(v) for c, v in e["counters"].items()},
               "counters_per_dispatch_in_order": e["counters"],
               "dispatches_per_counter": {c: len(v) for c, v in e["counters"].items()}}
        ds = dur.get((name, grid))
        if ds:
            ent["durations_ns_in_order"] = list(ds)
            ds = sorted(ds)
            ent["duration_ns"] = {"mean": sum(ds) / len(ds), "min": ds[0], "calls": len(ds)}
        raw[f"{name} | grid {grid}"] = ent
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_counters.json"), "w") as f:
        json.dump({k: {kk: vv for kk, vv in v.items() if kk not in ("counters_per_dispatch_in_order", "durations_ns_in_order")} for k, v in raw.items()}, f, indent=1)
    with open(os.path.join(prof_dir, "workloads.json")) as f:
        meta = json.loads(f.read().split("PMC_WORKLOADS ", 1)[1])
    used = set()
    out = {}
    for wname, m in meta.items():
        match = m.get("match", wname)
        cands = [k for k, e in raw.items() if match in e["kernel"] and (k not in used or m.get("shares_group_with"))
                 and ("grid_blocks" not in m or e["grid_threads"] == m["grid_blocks"] * e["workgroup_size"])]
        if not cands:
            continue
        k = min(cands, key=lambda k: raw[k]["order"])
        used.add(k)
        e = dict(raw[k])
        c = e["counters_mean_per_dispatch"]
        if "dispatch_slice" in m:   # two workloads with the same kernel and grid: this one's dispatches are values[a:b] of every counter
            a, b = m["dispatch_slice"]
            c = {name: (sum(v[a:b]) / len(v[a:b])) for name, v in e["counters_per_dispatch_in_order"].items() if len(v[a:b])}
            ds = e.get("durations_ns_in_order", [])[a:b]
            if ds:
                e["duration_ns"] = {"mean": sum(ds) / len(ds), "min": min(ds), "calls": len(ds)}
        cand = m["candidates_per_launch"]
        ent = {"kernel": e["kernel"], "grid_threads": e["grid_threads"], "vgprs": e["vgprs"], "lds_bytes": e["lds_bytes"],
               "workload": m["workload"], "candidates_per_launch": cand, "counters_per_launch": c,
               "source": f"profiles/{tag}_kernel_counters.json (rocprofv3 --pmc, tools/profile_counters.sh)"}
        if "duration_ns" in e:
            ent["duration_us"] = e["duration_ns"]["mean"] / 1e3
        # a grid smaller than the chip (LS step: one 1024-thread block per restart) can only use its own CUs
        blocks = max(1, e["grid_threads"] // max(1, e.get("workgroup_size", 256)))
        cus = min(256, blocks)
        ent["cus_usable"] = cus
        if "SQ_INSTS_VALU" in c:
            ent["valu_wave_instr_per_candidate"] = c["SQ_INSTS_VALU"] / cand
            ent["valu_issue_bound_us"] = c["SQ_INSTS_VALU"] * VALU_CYC / (4 * cus * CLOCK) * 1e6
        if "SQ_INSTS_LDS" in c:
            ent["lds_wave_instr_per_candidate"] = c["SQ_INSTS_LDS"] / cand
        if "SQ_INSTS_SALU" in c:
            ent["salu_wave_instr_per_candidate"] = c["SQ_INSTS_SALU"] / cand
        if "SQ_LDS_IDX_ACTIVE" in c:
            ent["lds_array_bound_us"] = c["SQ_LDS_IDX_ACTIVE"] / (cus * CLOCK) * 1e6   # LDS-array cycles summed over CUs
        if "duration_us" in ent and "valu_issue_bound_us" in ent:
            bounds = {"valu_issue": ent["valu_issue_bound_us"]}
            if "lds_array_bound_us" in ent:
                bounds["lds_array"] = ent["lds_array_bound_us"]
            b = max(bounds, key=bounds.get)
            ent["roofline"] = {"bound": b, "frac": bounds[b] / ent["duration_us"], "bound_us": bounds[b],
                               "note": "issue bound at the spec clock with every SIMD / LDS array of the CUs the grid can occupy busy"}
        out[wname] = ent
    with open(os.path.join(ROOT, "profiles", "kernel_counters.json"), "w") as f:
        json.dump({"kernels": out} if False else out, f, indent=1)
    print(json.dumps({k: {kk: v.get(kk) for kk in ("duration_us", "valu_wave_instr_per_candidate", "lds_wave_instr_per_candidate",
                                                    "valu_issue_bound_us", "lds_array_bound_us", "roofline")} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
