#!/bin/bash
TAG=${1:-n2a}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "two_workgroups" > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_$TAG.log
for rep in 1 2; do
  for t in nt_n2=0 nt_n2=1 nt_n2=2 nt_n2=3; do
    timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('serial $t', round(d['ms_per_step'],4), {k:round(v,4) for k,v in r['family_ms_per_step'].items()})" | tee -a $O/ab_$TAG.txt
    timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --tune $t 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('overlap $t', round(d['ms_per_step'],4), {k:round(v,4) for k,v in r['family_ms_per_step'].items()})" | tee -a $O/ab_$TAG.txt
  done
done
