#!/bin/bash
# usage: tools/bucket_probe.sh [P ...] — node table pass per build of the cfg 3 sweep: hashed slots only (k_nodes_v),
# then minimiser buckets (k_nodes_m) with AMG_BUCKET_PROBES = each P given (variant builds of amg_build_x.o on the box)
cd $GRAFT_REPO_ROOT
show() { python3 tools/sweep_probe.py cfg3 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());t=d['t']
print('[$1] sweep_ms', round(d['sweep_ms'],2), 'nodes', d['nodes'])
for b in ('build1','build2','build3'):
    s=t[b+'_stages']; print('  ',b,round(t[b],2),{k:s[k] for k in ('node_upsert_head','node_upsert','node_rank','edge_upsert','node_count','edge_count') if k in s})"; }
AMG_NODE_BUCKETS=0 show "hashed only"
show "buckets, shipped P"
for P in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-function -DAMG_EXPERIMENTS=0 -DAMG_BUCKET_PROBES=$P $EXTRA -c amira_amd/csrc/amg_build_x.hip -o amira_amd/csrc/amg_build_x.o 2>&1 | grep -E "error"
  make -C amira_amd/csrc > /dev/null 2>&1
  show "buckets P=$P $EXTRA"
done
