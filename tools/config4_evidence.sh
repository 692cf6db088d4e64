#!/bin/bash
# BASELINE.json config 4 evidence (log-mel + 2-D CNN step): entry-point breakdown + rocprofv3 kernel stats.
TAG=${1:-r3c4}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 200 python tools/spectro_profile.py --dtype f16 --breakdown $O/logmel_breakdown_$TAG.csv > $O/logmel_step_$TAG.json 2> $O/logmel_step_$TAG.err; echo "rc=$?"; cat $O/logmel_step_$TAG.json
timeout 200 python tools/spectro_profile.py --dtype bf16 > $O/logmel_step_bf16_$TAG.json 2>/dev/null; cat $O/logmel_step_bf16_$TAG.json
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_logmel_$TAG -- python $R/tools/spectro_profile.py --dtype f16 --steps 10 > $O/rocprof_logmel_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R
DB=$(find $O/prof_logmel_$TAG -name "*_results.db" | head -1); python tools/rocpd_summary.py $DB $O/logmel_kernel_stats_$TAG.csv 2>&1 | tail -1
rm -rf $O/prof_logmel_$TAG
cat $O/logmel_breakdown_$TAG.csv; head -25 $O/logmel_kernel_stats_$TAG.csv | cut -c1-160
