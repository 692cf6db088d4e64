#!/bin/bash
# same-box A/B of bench tuning sets, 3 interleaved repetitions:  bash tools/r2_ab.sh "a=0" "a=1" ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2 3 4; do for t in "$@"; do
  timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --tune "$t" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s ms/step %.4f loss %r' % ('$t', d['ms_per_step'], d['config']['final_loss']))"
done; done
