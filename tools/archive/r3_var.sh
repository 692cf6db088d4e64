#!/bin/bash
# interleaved A/B of experiment builds of the library (tools/build_variant.sh): tools/r3_var.sh <tag> <variant> [<variant> ...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do for v in default "$@"; do
  if [ $v = default ]; then unset VOICEMAP_HIP_LIB; else export VOICEMAP_HIP_LIB=$R/voicemap_amd/lib/libvoicemap_hip_$v.so; fi
  timeout 200 python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_${TAG}_$v.log 2>&1
  echo "$v rc=$? $(tail -1 $O/bench_${TAG}_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["timing"]["block_ms_per_step"])')"
done; done
