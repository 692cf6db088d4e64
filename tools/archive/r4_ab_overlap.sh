#!/bin/bash
# step time of tuning sets (comma-separated key=value lists, bench.py --tune), interleaved twice: bash tools/r4_ab_overlap.sh "a=1,b=2" "a=0" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do for t in "$@"; do
  python bench.py --no-extras --no-cpu-baseline --blocks 3 --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-40s step %.3f' % ('$t', d['ms_per_step']))"
done; done
