#!/bin/bash
# Interleaved A/B of library builds on ONE box: bash tools/ab_libs.sh <tag> "<default|variant> <variant> ..." [rounds]
# (variants: voicemap_amd/lib/libvoicemap_hip_<variant>.so from tools/build_variant.sh).  Prints ms per step of every run and the medians.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; ROUNDS=${3:-3}
for rep in $(seq 1 $ROUNDS); do for v in $2; do
  L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_$v.so
  VOICEMAP_HIP_LIB=$L python bench.py --no-extras --no-cpu-baseline --blocks 5 2>> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], ' '.join('%s=%.3f' % (k[8:], v['ms_per_step']) for k, v in d['roofline'].get('families_serial', {}).items()))" | tee -a $OUT/runs.txt
done; done
python - $OUT/runs.txt <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    d[p[0]].append(float(p[1]))
for k, v in d.items():
    print("%-12s median %.4f ms  (%s)" % (k, statistics.median(v), " ".join("%.4f" % x for x in v)))
PY
