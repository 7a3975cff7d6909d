#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== default (BC by rule, presplit)"; python scripts/bwd_probe.py wgrad window
echo "== OCCF_WG_BC=64"; OCCF_WG_BC=64 python scripts/bwd_probe.py wgrad
echo "== OCCF_WG_BC=128 PRESPLIT=0"; OCCF_WG_PRESPLIT=0 python scripts/bwd_probe.py wgrad
echo "== train bench"; timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}); [print(k,v) for k,v in list(d['kernels'].items())[:8]]"
