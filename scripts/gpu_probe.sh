#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== LDS atomics"; scripts/build/lds_atomic_probe
echo "== wgrad default"; python scripts/bwd_probe.py wgrad
echo "== wgrad OCCF_WG_SWZ=1"; OCCF_WG_SWZ=1 python scripts/bwd_probe.py wgrad
echo "== wgrad OCCF_WG_TARGET=2048"; OCCF_WG_TARGET=2048 python scripts/bwd_probe.py wgrad
echo "== wgrad OCCF_WG_TARGET=512"; OCCF_WG_TARGET=512 python scripts/bwd_probe.py wgrad
echo "== window mfma / valu"; python scripts/bwd_probe.py window; OCCF_WATTN_BWD_MFMA=0 python scripts/bwd_probe.py window
echo "== msda tiled / plain"; python scripts/bwd_probe.py msda; OCCF_MSDA_TILED=0 python scripts/bwd_probe.py msda
