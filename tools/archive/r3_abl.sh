#!/bin/bash
# families_serial of bench.py for a list of library variants (voicemap_amd/lib/libvoicemap_hip_<name>.so; "default" = the product)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in "$@"; do
  L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_$v.so
  VOICEMAP_HIP_LIB=$L timeout 120 python bench.py --steps 10 --warmup 3 --blocks 3 --no-cpu-baseline --no-extras --allow-nonfinite 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families_serial']
print('%-10s step %.3f ms | fwd %s = %.3f | dgrad %s = %.3f | wgrad %.3f' % ('$v', d['ms_per_step'], ' '.join('%.0f'%(l['ms']*1e3) for l in f['vm_conv_fwd']['launches']), f['vm_conv_fwd']['ms_per_step'], ' '.join('%.0f'%(l['ms']*1e3) for l in f['vm_conv_dgrad']['launches']), f['vm_conv_dgrad']['ms_per_step'], f['vm_conv_wgrad']['ms_per_step']))"
done
