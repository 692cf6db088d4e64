#!/bin/bash
# SQ / cache counters of the config-4 front-end and first-layer kernels (own passes, no tracing beside --pmc).
TAG=${1:-r3k}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
P="python $R/tools/spectro_profile.py --dtype f16 --steps 3"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_A_$TAG -- $P > $O/pmc_A_$TAG.log 2>&1; echo "rc=$?"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $O/pmc_B_$TAG -- $P > $O/pmc_B_$TAG.log 2>&1; echo "rc=$?"
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_C_$TAG -- $P > $O/pmc_C_$TAG.log 2>&1; echo "rc=$?"
cd $R
for X in A B C; do python tools/pmc_summary.py $O/pmc_${X}_$TAG > $O/logmel_pmc_${X}_$TAG.csv 2>&1; rm -rf $O/pmc_${X}_$TAG; grep -E "^kernel|stft|conv2d_first" $O/logmel_pmc_${X}_$TAG.csv; tail -3 $O/pmc_${X}_$TAG.log; done
