#!/bin/bash
# Parity + throughput of every inner-loop formulation of the register scan kernel
# (LS_NQ_SCAN_VARIANT: 0 = default fp32 product test, 1 = packed u16 min3).
set -o pipefail
mkdir -p gpurun_out
for v in ${VARIANTS:-0 1}; do
  echo "=== variant $v"
  LS_NQ_SCAN_VARIANT=$v timeout -k 10 200 python -m pytest tests/test_gpu_nqueens.py -m gpu -x -q 2>&1 | tail -2 || exit 1
  LS_NQ_SCAN_VARIANT=$v timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null \
     | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v value %.4g n/s  ms/step %.2f  launch_ms %.2f frac %.2f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))" || exit 1
done
