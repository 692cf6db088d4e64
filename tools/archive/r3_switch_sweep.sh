#!/bin/bash
# are the engine's stream / fusion switches still at their best settings?  gpurun -- 'bash tools/r3_switch_sweep.sh'
for rep in 1 2; do
for T in "" "split_towers=0" "wgrad_after_dgrad=1" "overlap_wgrad=0" "fold_pairs=0"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras ${T:+--tune $T} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s %.4f ms' % ('${T:-default}', d['ms_per_step']))"
done; done
