# A/B of the conv_nt3_kernel variants on one box: parity tests of the entry points that take packed weights, then interleaved bench runs
# (tuning keys: nt3 = bit 0 forward / bit 1 dgrad use conv_nt3_kernel; nt3_pipe = the interleaved K loop)
OUT=gpurun_out/${1:-r4ab}; mkdir -p $OUT
python -m pytest tests/test_gpu_fold.py tests/test_gpu_kernels.py -x -q -m gpu -k "fold or bnred or fwd_pool or pack_nt" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
i=0
TUNES="${2:-nt3_lean=3 nt3_lean=0 nt3=0 nt3_lean=3 nt3_lean=0 nt3=0}"
for t in $TUNES; do
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
