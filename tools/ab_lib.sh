#!/bin/bash
# interleaved A/B of two builds of the library: gpurun -- 'bash tools/ab_lib.sh voicemap_amd/lib_ab/old.so'
for rep in 1 2 3 4; do
  for L in "$1" ""; do
    VOICEMAP_HIP_LIB=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %.4f ms' % ('${L:-default}', d['ms_per_step']))"
  done
done
