#!/bin/bash
# Gate + launch-geometry sweep + evidence run in ONE gpurun call (box time is the scarce resource):
#   gpurun --timeout 1000 -- 'bash tools/round_run.sh <tag>'
TAG=${1:-rr}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv1_fused" > $O/gate_$TAG.log 2>&1
rc=$?; tail -3 $O/gate_$TAG.log; if [ $rc -ne 0 ]; then echo "GATE(kernels) FAILED"; tail -40 $O/gate_$TAG.log; exit 1; fi
timeout 300 python -m pytest tests/test_gpu_e2e.py -x -q > $O/gate_e2e_$TAG.log 2>&1
rc=$?; tail -3 $O/gate_e2e_$TAG.log; if [ $rc -ne 0 ]; then echo "GATE(e2e) FAILED"; tail -40 $O/gate_e2e_$TAG.log; exit 1; fi
for kv in f1_fwd_blocks=2048 f1_fwd_blocks=4096 f1_fwd_blocks=8192 f1_blocks=1024 f1_blocks=4096; do
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --tune $kv --breakdown $O/bd_${TAG}_$kv.csv > $O/sweep_${TAG}_$kv.log 2>&1
  echo "$kv: $(grep -h 'conv1_fused' $O/bd_${TAG}_$kv.csv | tr '\n' ' ') $(tail -1 $O/sweep_${TAG}_$kv.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)"
done
bash tools/final_run.sh $TAG
