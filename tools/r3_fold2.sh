#!/bin/bash
# Folded path: tests that changed + kernel timelines (default streams and serial) of the folded step.
TAG=${1:-f2}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_fold.py tests/test_gpu_e2e.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest_$TAG.log
B="python $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-extras"
S="$B --no-overlap-wgrad --tune split_towers=0"
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -- $B > /dev/null 2>&1; echo "default trace rc=$?"
python $R/tools/timeline.py $O/trace_$TAG > $O/step_timeline_default_$TAG.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/traces_$TAG -- $S > /dev/null 2>&1; echo "serial trace rc=$?"
python $R/tools/timeline.py $O/traces_$TAG > $O/step_timeline_serial_$TAG.txt 2>&1
rm -rf $O/trace_$TAG $O/traces_$TAG
