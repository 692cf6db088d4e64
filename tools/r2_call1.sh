#!/bin/bash
# round 2, GPU call 1: gate on the new cfg-A geometry parity tests, w4 / w4p probes, nt_w4 A/B in the step
TAG=${1:-r2c1}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_$TAG.log
(timeout 120 tools/probe/conv_w4_probe; timeout 120 tools/probe/conv_w4p_probe) > $O/w4_probes_$TAG.txt 2>&1; cat $O/w4_probes_$TAG.txt
for rep in 1 2; do
  for t in nt_w4=0 nt_w4=1 nt_w4=2 nt_korder=1 nt_w4=1,nt_korder=1; do
    timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('serial $t', round(d['ms_per_step'],4), {k:round(v,4) for k,v in r['family_ms_per_step'].items()})" | tee -a $O/ab_$TAG.txt
    timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('overlap $t', round(d['ms_per_step'],4), {k:round(v,4) for k,v in r['family_ms_per_step'].items()})" | tee -a $O/ab_$TAG.txt
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --breakdown $O/breakdown_$TAG.csv > $O/bench_$TAG.log 2>&1; cat $O/breakdown_$TAG.csv
