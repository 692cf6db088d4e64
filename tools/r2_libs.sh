#!/bin/bash
# same-box comparison of several builds of the library:  bash tools/r2_libs.sh "" _abl _pad64 ...   (suffixes of voicemap_amd/lib/libvoicemap_hip*.so)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2; do
for sfx in "$@"; do
  [ "$sfx" = "base" ] && sfx=""
  VOICEMAP_HIP_LIB=$R/voicemap_amd/lib/libvoicemap_hip$sfx.so timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune split_towers=0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['families_serial']
print('%-10s step %.4f  fwd %.4f  dgrad %.4f  wgrad %.4f' % ('$sfx' or 'base', d['ms_per_step'], r['vm_conv_fwd']['ms_per_step'], r['vm_conv_dgrad']['ms_per_step'], r['vm_conv_wgrad']['ms_per_step']))"
done; done
