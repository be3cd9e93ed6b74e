#!/bin/bash
# usage: tools/pmc_probe2.sh <tag> "<counters>" [bench args...] — one counter pass over one bench run
tag=$1; shift; C=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${tag} -o out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${tag}.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}_summary.txt
