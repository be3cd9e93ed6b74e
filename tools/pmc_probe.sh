#!/bin/bash
# usage: tools/pmc_probe.sh <tag> [bench args...]  — SQ and TCC counter passes over one bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in sq tcc; do
  if [ $pass = sq ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; else C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; fi
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${tag}_$pass -o out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${tag}_$pass.log 2>&1
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_tcc > gpurun_out/pmc_${tag}_summary.txt; cat gpurun_out/pmc_${tag}_summary.txt
