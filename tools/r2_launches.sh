#!/bin/bash
# per-launch serial attribution for a list of tuning sets:  bash tools/r2_launches.sh TAG "nt_n2=0" "nt_n2=3" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for t in "$@"; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune "$t" 2>/dev/null | tail -1 > $O/launch_${TAG}_$t.json
  python - "$O/launch_${TAG}_$t.json" "$t" <<'PY'
import sys, json
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("== %s: %.4f ms/step  hbm_frac %.3f  final_loss %r" % (sys.argv[2], d["ms_per_step"], r["step_hbm_frac"], d["config"]["final_loss"]))
for nm, f in r["families_serial"].items():
    print("  %-14s %.4f ms  (%.3f of peak)  " % (nm, f["ms_per_step"], f["frac_of_mfma_peak"]) + "  ".join("L%d %d->%d: %.1f us %.0f TF" % (l["L"], l["c_in"], l["c_out"], l["ms"] * 1e3, l["tflops"]) for l in f["launches"]))
PY
done
