#!/bin/bash
# PMC passes for the NT GEMM kernels (separate passes, kernel-trace only): gpurun -- 'bash tools/pmc_nt8.sh [bench args]'
export TMPDIR=/tmp; R=$PWD; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
P3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS"
P4="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $@ > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
done
python $R/tools/pmc_summary.py /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 /tmp/pmc_4 > $R/gpurun_out/pmc_nt8.csv
grep -E "^kernel|nt8|conv_nt" $R/gpurun_out/pmc_nt8.csv
