#!/bin/bash
# bench (3 reps) + one-step serial timeline under rocprofv3:  bash tools/r2_tl.sh TAG
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2 3; do timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f loss %r' % (d['ms_per_step'], d['config']['final_loss']))"; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$TAG -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $TUNE > /dev/null 2>&1
cd $R; python tools/timeline.py $O/tl_$TAG > $O/timeline_$TAG.txt; head -3 $O/timeline_$TAG.txt; rm -rf $O/tl_$TAG
