#!/bin/bash
# per-kernel timing of the nt8 kernel under its ablation flags:  gpurun -- 'bash tools/ablate_nt8.sh "0 2 4 8 16 24 28"'
export TMPDIR=/tmp; R=$PWD; cd /tmp
for A in $1; do
  rm -rf /tmp/tr_$A
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$A -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --tune nt_ablate=$A$2 > /dev/null 2>&1
  echo "== ablate $A $2"; python $R/tools/trace_kernels.py /tmp/tr_$A ${3:-conv_nt8}
done
