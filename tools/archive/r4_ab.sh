# interleaved A/B of tuning sets on one box: bash tools/r4_ab.sh <tag> "<tune1> <tune2> ..." [pytest -k expression]
OUT=gpurun_out/${1:-r4ab}; mkdir -p $OUT
if [ -n "$3" ]; then python -m pytest tests/test_gpu_fold.py tests/test_gpu_kernels.py -x -q -m gpu -k "$3" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log; fi
i=0
for t in $2; do
  i=$((i+1)); python bench.py --no-extras --no-cpu-baseline --blocks 3 --tune $t > "$OUT/bench_$i.$t.json" 2>> $OUT/bench.err
done
python - $OUT <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    fam=d["roofline"]["families_serial"]
    print("%-34s" % f.split("/")[-1][:-5], "step %.3f" % d["ms_per_step"], " | ".join("%s %.3f [%s]" % (k[8:], v["ms_per_step"], " ".join("%.0f" % (l["ms"]*1e3) for l in v["launches"])) for k,v in fam.items()))
PY
