#!/bin/bash
# final validation of the round's build: GPU suite, bench, kernel-trace stats of the config runs, parity soak
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
timeout -k 10 700 python -m pytest tests/ -m gpu -x -q > gpurun_out/r4m_final_tests.log 2>&1; tail -2 gpurun_out/r4m_final_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r4m_bench_final.json 2> gpurun_out/r4m_bench_final.err; echo bench rc=$?
python tools/show_bench.py gpurun_out/r4m_bench_final.json > gpurun_out/r4m_bench_final.txt 2>&1
mkdir -p gpurun_out/prof_r04m
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r04m/configs" -- python3 "$ROOT/tools/run_configs.py" > "$ROOT/gpurun_out/prof_r04m/configs.log" 2>&1; echo configs rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_r04m/multi" -- python3 "$ROOT/tools/multi_sched_trace.py" --repeat 1 > "$ROOT/gpurun_out/prof_r04m/multi.log" 2>&1; echo multi rc=$?
cd "$ROOT"
for d in configs multi; do f=$(find gpurun_out/prof_r04m/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04m_kernel_stats_$d.csv; done
ls gpurun_out/r04m_kernel_stats_*.csv
timeout -k 10 400 python tests/soak.py 120 101 > gpurun_out/r04m_parity_soak.txt 2>&1; tail -1 gpurun_out/r04m_parity_soak.txt | cut -c1-200
