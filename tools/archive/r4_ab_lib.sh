#!/bin/bash
# interleaved A/B of library builds with the GEMM family times: bash tools/r4_ab_lib.sh <tag> "<lib1|default> <lib2> ..." [pytest -k expr]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; OUT=gpurun_out/${1:-r4abl}; mkdir -p $OUT
if [ -n "$3" ]; then python -m pytest tests/test_gpu_fold.py tests/test_gpu_kernels.py tests/test_gpu_golden_step.py -x -q -m gpu -k "$3" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log; fi
i=0
for rep in 1 2; do for v in $2; do
  i=$((i+1)); L=""; [ "$v" != "default" ] && L=$R/voicemap_amd/lib/libvoicemap_hip_$v.so
  VOICEMAP_HIP_LIB=$L python bench.py --no-extras --no-cpu-baseline --blocks 3 > "$OUT/bench_$i.$v.json" 2>> $OUT/bench.err
done; done
python - $OUT <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    fam=d["roofline"]["families_serial"]
    print("%-24s" % f.split("/")[-1][:-5], "step %.3f" % d["ms_per_step"], " | ".join("%s %.3f [%s]" % (k[8:], v["ms_per_step"], " ".join("%.0f" % (l["ms"]*1e3) for l in v["launches"])) for k,v in fam.items()))
PY
