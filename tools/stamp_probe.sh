#!/bin/bash
# usage: tools/stamp_probe.sh — where a tile of the fused table pass spends its time (s_memtime stamps per phase;
# needs the EXPERIMENTS build, made here on the GPU box)
cd $GRAFT_REPO_ROOT
make -C amira_amd/csrc clean > /dev/null; make -C amira_amd/csrc -j32 EXPERIMENTS=1 2>&1 | grep -E "error" 
AMG_F_STAMPS=1 python3 tools/sweep_probe.py cfg3-sweep 2>&1 | grep "k_graph_x phases" | head -6
