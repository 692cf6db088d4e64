R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $O/tl_cfgB -o t -- python $R/tools/probe/steps_for_profile.py cfgB 32 50 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl_cfgB -name "*_results.db" | head -1) > $O/step_timeline_cfgB_32pairs_r6a.txt 2>&1
rm -rf $O/tl_cfgB
timeout 200 rocprofv3 --kernel-trace -d $O/tl_cfgA -o t -- python $R/tools/probe/steps_for_profile.py cfgA 128 30 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $O/tl_cfgA -name "*_results.db" | head -1) > $O/step_timeline_overlapped_r6a.txt 2>&1
rm -rf $O/tl_cfgA
