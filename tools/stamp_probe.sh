#!/bin/bash
# usage: tools/stamp_probe.sh — where a tile of the table pass of amg_build_f.hip spends its time (s_memtime stamps per
# phase; needs the EXPERIMENTS build, made here on the GPU box).  One launch (AMG_FUSED=1) and the two halves (=2).
cd $GRAFT_REPO_ROOT
make -C amira_amd/csrc clean > /dev/null; make -C amira_amd/csrc -j32 EXPERIMENTS=1 2>&1 | grep -E "error" 
for cfg in "1 1" "2 1" "2 2"; do
  set -- $cfg
  echo "AMG_FUSED=$1 stamps of phase $2"
  AMG_FUSED=$1 AMG_F_STAMPS=$2 python3 tools/sweep_probe.py cfg3-sweep 2>&1 | grep "k_graph_x phases" | head -3
done
