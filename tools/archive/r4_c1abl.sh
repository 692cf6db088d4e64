#!/bin/bash
# block-1 ablation timings: vm_conv1_fused_{fwd,bwd} rows of the entry-point breakdown for library variants built with
# tools/build_variant.sh c1abl<bits> -DVM_C1_ABL=<bits> conv1_fused.hip      gpurun -- 'bash tools/r4_c1abl.sh default 1 2 ...'
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/c1abl; mkdir -p $O
for v in "$@"; do
  L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_c1abl$v.so
  VOICEMAP_HIP_LIB=$L timeout 120 python bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline --no-extras --allow-nonfinite --breakdown $O/bd_$v.csv > /dev/null 2> $O/err_$v.txt
  printf "%-8s fwd %s  bwd %s\n" $v "$(grep '^vm_conv1_fused_fwd' $O/bd_$v.csv | cut -d, -f3)" "$(grep '^vm_conv1_fused_bwd' $O/bd_$v.csv | cut -d, -f3)"
done
