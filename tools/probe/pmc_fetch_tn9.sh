#!/bin/bash
# FETCH_SIZE (KB per launch, gfx950 counts half of a wide streaming read) of the conv_tn9_kernel launches of a serial cfg-A step, for one
# or more library builds: bash tools/probe/pmc_fetch_tn9.sh <tag> "<default|variant> ..."   (variants: voicemap_amd/lib/libvoicemap_hip_<v>.so)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-fetch}; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for v in $2; do
  L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_$v.so
  rm -rf $O/pmc_$v
  VOICEMAP_HIP_LIB=$L timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$v -- python $R/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0 > $O/pmc_$v.log 2>&1
  python - $O/pmc_$v $v <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_tn9" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            d[r["Grid_Size"]].append(float(r["Counter_Value"]))
for g, v in sorted(d.items()):
    print("%-8s conv_tn9 grid %-8s launches %3d  FETCH_SIZE %.1f MB per launch (x2 for wide reads = %.1f MB)" % (sys.argv[2], g, len(v), sum(v) / len(v) / 1e3, 2 * sum(v) / len(v) / 1e3))
PY
  rm -rf $O/pmc_$v
done
