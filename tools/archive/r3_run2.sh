#!/bin/bash
# Round-3 evidence run: GPU suite, smoke, the full bench line (f16 headline + extras + CPU baseline), rocprofv3 kernel stats of the
# default step, the serial per-layer trace, the two PMC traffic passes and one SQ counter pass for the GEMM kernels.
#   gpurun --timeout 1500 -- 'bash tools/r3_run2.sh <tag>'
TAG=${1:-r3b}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/parity_report.csv
timeout 700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short --durations=15 -p no:cacheprovider > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_$TAG.log
cp $O/parity_report.csv $O/parity_report_$TAG.csv 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"
timeout 500 python bench.py --steps 20 --warmup 5 --breakdown $O/breakdown_$TAG.csv > $O/bench_$TAG.log 2> $O/bench_$TAG.err; echo "bench rc=$?"; tail -c 6000 $O/bench_$TAG.log
B="python $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-extras"
S="$B --no-overlap-wgrad --tune split_towers=0"
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -- $B > $O/rocprof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -- $S > /dev/null 2>&1; echo "serial trace rc=$?"
python $R/tools/trace_kernels.py $O/trace_$TAG conv_ 3 > $O/conv_kernels_by_layer_$TAG.txt 2>&1
python $R/tools/timeline.py $O/trace_$TAG > $O/step_timeline_serial_$TAG.txt 2>&1
P="python $R/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_$TAG -- $P > $O/pmc_${C}_$TAG.log 2>&1; echo "pmc $C rc=$?"
done
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_SQ_$TAG -- $P > $O/pmc_SQ_$TAG.log 2>&1; echo "pmc SQ rc=$?"
cd $R
DB=$(find $O/prof_$TAG -name "*_results.db" | head -1); python tools/rocpd_summary.py $DB $O/kernel_stats_$TAG.csv 2>&1 | tail -2
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_traffic_$TAG.json 2>&1 | tail -1
python tools/pmc_summary.py $O/pmc_SQ_$TAG > $O/pmc_sq_counters_$TAG.csv 2>&1
rm -rf $O/trace_$TAG $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_SQ_$TAG
head -12 $O/pmc_sq_counters_$TAG.csv
