#!/bin/bash
# cfg-A evidence of the tree (round 6): bench line, rocprofv3 kernel stats of the default step,
# serial per-layer trace + timeline, the two PMC traffic passes, one SQ counter pass.   gpurun --timeout 1200 -- 'bash tools/r6_evidence.sh <tag>'
TAG=${1:-r6p}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 500 python bench.py --steps 20 --warmup 5 --breakdown $O/breakdown_$TAG.csv > $O/bench_$TAG.log 2> $O/bench_$TAG.err; echo "bench rc=$?"; head -c 700 $O/bench_$TAG.log; echo
B="python $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-extras"
S="$B --no-overlap-wgrad --tune split_towers=0"
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -- $B > $O/rocprof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -- $S > /dev/null 2>&1; echo "serial trace rc=$?"
python $R/tools/trace_kernels.py $O/trace_$TAG conv_ 3 > $O/conv_kernels_by_layer_$TAG.txt 2>&1
python $R/tools/timeline.py $O/trace_$TAG > $O/step_timeline_serial_$TAG.txt 2>&1
# the overlapped (default streams) step and the reference's own cfg-B step at 32 pairs: timelines from rocpd databases
timeout 200 rocprofv3 --kernel-trace -d $O/tl_cfgA_$TAG -o t -- python $R/tools/probe/steps_for_profile.py cfgA 128 30 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl_cfgA_$TAG -name "*_results.db" | head -1) > $O/step_timeline_overlapped_$TAG.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d $O/tl_cfgB_$TAG -o t -- python $R/tools/probe/steps_for_profile.py cfgB 32 50 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl_cfgB_$TAG -name "*_results.db" | head -1) > $O/step_timeline_cfgB_32pairs_$TAG.txt 2>&1
rm -rf $O/tl_cfgA_$TAG $O/tl_cfgB_$TAG
P="python $R/bench.py --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-extras --no-overlap-wgrad --tune split_towers=0"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_$TAG -- $P > $O/pmc_${C}_$TAG.log 2>&1; echo "pmc $C rc=$?"
done
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_SQ_$TAG -- $P > $O/pmc_SQ_$TAG.log 2>&1; echo "pmc SQ rc=$?"
cd $R
DB=$(find $O/prof_$TAG -name "*_results.db" | head -1); python tools/rocpd_summary.py $DB $O/kernel_stats_$TAG.csv 2>&1 | tail -1
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_traffic_$TAG.json 2>&1 | tail -1
python tools/pmc_summary.py $O/pmc_SQ_$TAG > $O/pmc_sq_counters_$TAG.csv 2>&1
rm -rf $O/trace_$TAG $O/pmc_FETCH_SIZE_$TAG $O/pmc_WRITE_SIZE_$TAG $O/pmc_SQ_$TAG $O/prof_$TAG
head -3 $O/step_timeline_serial_$TAG.txt; head -6 $O/pmc_sq_counters_$TAG.csv
# round 6: power / clocks per launch shape, conv_nt3 slot timelines (experiment build, if present), pairdist shard, launch times
cd $R
timeout 120 python tools/probe/kernel_power.py 2>&1 | grep -v amdgpu.ids > $O/kernel_power_$TAG.txt; tail -4 $O/kernel_power_$TAG.txt
timeout 60 python tools/probe/pairdist_time.py 2>&1 | grep -v amdgpu.ids > $O/pairdist_$TAG.txt; cat $O/pairdist_$TAG.txt
timeout 60 python tools/probe/nt3_launch_times.py 2>&1 | grep -v amdgpu.ids > $O/nt3_launch_times_$TAG.txt
if [ -f voicemap_amd/lib/libvoicemap_hip_prof.so ]; then
  VOICEMAP_HIP_LIB=$R/voicemap_amd/lib/libvoicemap_hip_prof.so timeout 100 python tools/probe/nt3_slots.py 2>&1 | grep -v amdgpu.ids > $O/nt3_slots_$TAG.txt; head -2 $O/nt3_slots_$TAG.txt | cut -c1-300
fi
