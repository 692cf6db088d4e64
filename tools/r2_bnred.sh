#!/bin/bash
# fused BatchNorm-backward sums in the dgrad epilogue: parity tests, then same-box A/B of the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bnred or from_sums" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x 2>&1 | tail -8
for rep in 1 2 3; do
for t in fused_bn_reduce=0 fused_bn_reduce=1; do
  timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --tune "$t" 2>/dev/null | tail -1 > $O/bnred_$t.json
  python - "$O/bnred_$t.json" "$t" <<'PY'
import sys, json
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("== %s: %.4f ms/step  final_loss %r  " % (sys.argv[2], d["ms_per_step"], d["config"]["final_loss"]) + "  ".join("%s %.4f" % (k[8:], f["ms_per_step"]) for k, f in r["families_serial"].items()))
PY
done; done
