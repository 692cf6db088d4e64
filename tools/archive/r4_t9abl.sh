#!/bin/bash
# wgrad ablation timings (conv_tn9_kernel): families_serial of bench.py for library variants built with
# tools/build_variant.sh t9abl<bits> -DVM_TN9_ABL=<bits> conv_wgrad.hip       gpurun -- 'bash tools/r4_t9abl.sh default 1 2 ...'
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in "$@"; do
  L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_t9abl$v.so
  VOICEMAP_HIP_LIB=$L timeout 120 python bench.py --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-extras --allow-nonfinite 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families_serial']
print('%-8s step %.3f | wgrad %s = %.3f' % ('$v', d['ms_per_step'], ' '.join('%.0f'%(l['ms']*1e3) for l in f['vm_conv_wgrad']['launches']), f['vm_conv_wgrad']['ms_per_step']))"
done
