#!/bin/bash
# Is the step power-limited?  Runs ~6 s of cfg-A steps and samples rocm-smi (power, clocks, limits) beside them.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
echo "--- under load"
python tools/probe/steps_for_profile.py cfgA 128 2500 > /dev/null 2>&1 &
PID=$!
sleep 3.5
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ';'; echo; sleep 0.4; done
wait $PID
echo "--- amd-smi"
amd-smi metric -p -c 2>&1 | head -40
