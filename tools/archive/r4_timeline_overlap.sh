#!/bin/bash
# kernel timeline of the DEFAULT (overlapped) step: which stream runs what, where the main stream waits.  gpurun -- 'bash tools/r4_timeline_overlap.sh <tag>'
TAG=${1:-ovl}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -- python $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo "trace rc=$?"
cd $R; python tools/timeline.py $O/trace_$TAG > $O/step_timeline_overlapped_$TAG.txt 2>&1; rm -rf $O/trace_$TAG; head -5 $O/step_timeline_overlapped_$TAG.txt
