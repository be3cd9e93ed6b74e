#!/bin/bash
# usage: tools/builds_probe.sh ["ENV=1 ENV2=1" ...] — per-build stage times of the cfg 3 sweep (HIP events, ms), once per
# environment given (default: none)
cd $GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- ""
for e in "$@"; do env $e python3 tools/sweep_probe.py cfg3 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());t=d['t']
print('[$e] sweep_ms', round(d['sweep_ms'],2))
for b in ('build1','build2','build3'): print('  ',b,round(t[b],2),t[b+'_stages'])"; done
