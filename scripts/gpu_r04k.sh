#!/bin/bash
# round 4, visit k: msda3d value-gradient tiles -- LDS budget / thread-count sweep (scripts/bwd_probe.py msda)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
for kb in 124 140 156; do for th in 1024 768; do
  echo -n "OCCF_MSDA_LDS_KB=$kb OCCF_MSDA_TILE_THREADS=$th: "
  OCCF_MSDA_LDS_KB=$kb OCCF_MSDA_TILE_THREADS=$th timeout 120 python scripts/bwd_probe.py msda 2>/dev/null | tail -1
done; done | tee $O/r04k_msda_lds_sweep.txt
cd /tmp && export TMPDIR=/tmp
for kb in 124 156; do
OCCF_MSDA_LDS_KB=$kb timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04k_prof$kb -- python $R/scripts/bwd_probe.py msda > /dev/null 2>&1
python $R/scripts/summarize_prof.py $R/$O/r04k_prof$kb | grep -i "msda\|kernel " | head -8 | cut -c1-160
done
