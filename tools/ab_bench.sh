#!/bin/bash
# In-box A/B of bench.py under different environments, interleaved:  bash tools/ab_bench.sh ROUNDS "ENV_A" "ENV_B" ... [-- bench args]
# (boxes differ by ~3 % in clocks: only numbers from one call on one box compare)
R=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $R); do
  for e in "${ENVS[@]}"; do
    out=$(env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-hbm-table "$@" 2>/dev/null | tail -1)
    python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-60s %9.1f samples/s %7.3f ms  PCIe-incl %s  phase %s  gemm %s TF' % (sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('pcie_inclusive_samples_per_s'), d['config'].get('hw_queue_phase'), d['roofline'].get('achieved')))" "$out" "$e"
  done
done
