#!/bin/bash
# Folded-BatchNorm path: its own tests, the end-to-end / full-size tests that now run through it, and an interleaved A/B of the step.
#   gpurun --timeout 1200 -- 'bash tools/r3_fold.sh <tag> [pytest targets]'
TAG=${1:-f1}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=${2:-"tests/test_gpu_fold.py tests/test_gpu_e2e.py tests/test_gpu_kernels.py::test_prep_conv_weights_batch_equals_single"}
timeout 500 python -m pytest $T -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
tail -30 $O/pytest_$TAG.log
for t in ${VARIANTS:-"fold_affine=0" "fold_pairs=0" "fold_pairs=1" "fold_pairs=1,split_towers=0" "fold_affine=0" "fold_pairs=0" "fold_pairs=1" "fold_pairs=1,split_towers=0"}; do
  timeout 200 python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $t > $O/bench_${TAG}_$t.log 2>&1
  echo "$t rc=$? $(tail -1 $O/bench_${TAG}_$t.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["timing"]["block_ms_per_step"], {k:(v["ms_per_step"], v["frac_of_mfma_peak"]) for k,v in d["roofline"]["families_serial"].items()})')"
done
