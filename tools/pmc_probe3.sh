#!/bin/bash
# usage: tools/pmc_probe3.sh <tag> [bench args...] — issue / wait / texture-path / launch counters of one bench
# run, one --pmc pass per group (the environment is inherited)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
dirs=""
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
         "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_CSN_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc3_${tag}_$i -o out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/pmc3_${tag}_$i.log 2>&1 || tail -3 $R/gpurun_out/pmc3_${tag}_$i.log
  dirs="$dirs gpurun_out/pmc3_${tag}_$i"
done
cd $R && python3 tools/pmc_summary.py $dirs > gpurun_out/pmc3_${tag}_summary.txt
find gpurun_out -name "*.db" -path "*pmc3_${tag}_*" -delete
grep -A60 "^k_nodes_m\|^k_edges_v\|^k_corr_" gpurun_out/pmc3_${tag}_summary.txt | head -150
