#!/bin/bash
# rocprofv3 counter evidence for every kernel family (scan, table, LS step, scheduling): instruction counters in one pass,
# busy / wave counters in a second, each with --kernel-trace only (no other trace domain with --pmc).  The program runs
# DIRECTLY after `--` (python3, no wrapper).  Usage on the GPU box, from the repo root:
#   bash tools/profile_counters.sh <tag>
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d "$OUT/insts" -- python3 "$ROOT/tools/pmc_workloads.py" 3 > "$OUT/insts.log" 2>&1 || { tail -20 "$OUT/insts.log"; exit 1; }
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/cycles" -- python3 "$ROOT/tools/pmc_workloads.py" 3 > "$OUT/cycles.log" 2>&1 || { tail -20 "$OUT/cycles.log"; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/lds" -- python3 "$ROOT/tools/pmc_workloads.py" 3 > "$OUT/lds.log" 2>&1 || { tail -20 "$OUT/lds.log"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/pmc_workloads.py" 3 > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
grep -h PMC_WORKLOADS "$OUT/insts.log" > "$OUT/workloads.json"
python3 "$ROOT/tools/summarise_counters.py" "$OUT" "$TAG" > "$OUT/summary.log" 2>&1 || { tail -20 "$OUT/summary.log"; exit 1; }
tail -5 "$OUT/summary.log"
