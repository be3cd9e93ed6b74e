#!/bin/bash
# usage: tools/pmc_probe4.sh <tag> [bench args...] — texture-path (TA / TCP), vector-memory issue and L2 counters of the
# table passes of one bench run, one --pmc pass per group (the environment is inherited)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
dirs=""
for C in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc4_${tag}_$i -o out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/pmc4_${tag}_$i.log 2>&1 || tail -3 $R/gpurun_out/pmc4_${tag}_$i.log
  dirs="$dirs gpurun_out/pmc4_${tag}_$i"
done
cd $R && python3 tools/pmc_summary.py $dirs > gpurun_out/pmc4_${tag}_summary.txt
find gpurun_out -name "*.db" -path "*pmc4_${tag}_*" -delete
grep -A22 "^k_nodes_m\|^k_edges_v\|^k_corr_" gpurun_out/pmc4_${tag}_summary.txt | head -120
