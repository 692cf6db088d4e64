#!/bin/bash
# Round-3 gate run: the whole GPU suite (pruned kernel set + f16 storage), smoke, and the bench in both 16-bit modes.
#   gpurun --timeout 1200 -- 'bash tools/r3_run1.sh <tag>'
TAG=${1:-r3a}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/parity_report.csv
timeout 700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short --durations=25 -p no:cacheprovider > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_$TAG.log
cp $O/parity_report.csv $O/parity_report_$TAG.csv 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -6 $O/smoke_$TAG.log
timeout 300 python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --breakdown $O/breakdown_f16_$TAG.csv > $O/bench_f16_$TAG.log 2>&1; echo "bench f16 rc=$?"; tail -c 3000 $O/bench_f16_$TAG.log
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --breakdown $O/breakdown_bf16_$TAG.csv > $O/bench_bf16_$TAG.log 2>&1; echo "bench bf16 rc=$?"; tail -c 1500 $O/bench_bf16_$TAG.log
