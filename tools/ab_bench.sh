#!/bin/bash
# interleaved A/B of bench.py tuning sets:  gpurun -- 'bash tools/ab_bench.sh "nt_p8=0" "nt_p8=1" ...'
R=$PWD
for rep in 1 2 3; do
  for T in "$@"; do
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --tune "$T" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.4f ms  (dominant %.4f)' % ('$T', d['ms_per_step'], d['roofline']['launch_ms']))"
  done
done
