#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== kernels"; python scripts/bwd_probe.py window
echo "== forward bench"; timeout 600 python bench.py --mode forward --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}); [print(k,v) for k,v in list(d['kernels'].items())[:12]]"
echo "== train bench"; timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}); [print(k,v) for k,v in list(d['kernels'].items())[:8]]"
