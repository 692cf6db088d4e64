#!/bin/bash
# Interleaved A/B of two checkouts of the repository on ONE box: bash tools/ab_trees.sh <dirA> <dirB> [rounds]   (each with its own built library)
A=$1; B=$2; ROUNDS=${3:-3}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; : > $O/ab_trees.txt
for rep in $(seq 1 $ROUNDS); do for d in $A $B; do
  (cd $R/$d && python bench.py --no-extras --no-cpu-baseline --blocks 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['ms_per_step'], ' '.join('%s=%.3f' % (k[8:], v['ms_per_step']) for k, v in d['roofline'].get('families_serial', {}).items()))") | tee -a $O/ab_trees.txt
done; done
python - $O/ab_trees.txt <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    d[p[0]].append(float(p[1]))
for k, v in d.items():
    print("%-12s median %.4f ms  (%s)" % (k, statistics.median(v), " ".join("%.4f" % x for x in v)))
PY
