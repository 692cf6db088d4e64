#!/bin/bash
# Round-end evidence run on the GPU box:  gpurun --timeout 1500 -- 'bash tools/final_run.sh <tag>'
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --breakdown $O/breakdown_$TAG.csv > $O/bench_$TAG.log 2>&1; echo "bench rc=$?"
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0 > $O/bench_${TAG}_serial.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0 > /dev/null 2>&1
python $R/tools/trace_kernels.py $O/trace_$TAG conv_ 3 > $O/conv_kernels_by_layer_$TAG.txt
# one step, kernel by kernel in launch order (rocprofv3 serialises the queues: durations add up)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$TAG -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/tools/timeline.py $O/tl_$TAG > $O/step_timeline_$TAG.txt; rm -rf $O/tl_$TAG
[ -x $R/tools/probe/launch_gap_probe ] || hipcc --offload-arch=gfx950 -O3 -o $R/tools/probe/launch_gap_probe $R/tools/probe/launch_gap_probe.hip >/dev/null 2>&1
timeout 60 $R/tools/probe/launch_gap_probe > $O/launch_gap_probe_$TAG.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_$TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0 > $O/pmc_${C}_$TAG.log 2>&1; echo "pmc $C rc=$?"
done
cd $R
DB=$(find $O/prof_$TAG -name "*_results.db" | head -1); python tools/rocpd_summary.py $DB $O/kernel_stats_$TAG.csv
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_traffic_$TAG.json
tail -3 $O/pytest_$TAG.log; tail -1 $O/bench_$TAG.log; tail -1 $O/bench_${TAG}_serial.log
