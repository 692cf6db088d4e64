#!/bin/bash
# effective GPU clock of the wgrad kernel under ablation flags: GRBM_GUI_ACTIVE cycles / kernel duration
export TMPDIR=/tmp; R=$PWD; cd /tmp
for A in $1; do
  rm -rf /tmp/ck_$A
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/ck_$A -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-wgrad --tune nt_ablate=$A > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/ck_$A/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if '${2:-conv_tn8x}' in r['Kernel_Name']:
            k=r['Grid_Size']; agg[k][r['Counter_Name']].append(float(r['Counter_Value'])); agg[k]['dur'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    print('ablate $A grid',k,'dur %.1f us  GUI_ACTIVE %.3g -> %.2f GHz  MFMA_BUSY/SIMD-cycles %.2f' % (m['dur'], m['GRBM_GUI_ACTIVE'], m['GRBM_GUI_ACTIVE']/m['dur']/1e3, m['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*m['GRBM_GUI_ACTIVE'])))
PY
done
