#!/bin/bash
# round 4, visit b: the kitti_effb7_128 training-parity test alone, with and without OCCF_TEST_POISON (full output)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
OCCF_TEST_POISON=1 timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_128" > $O/r04b_kitti128_poison.log 2>&1
tail -60 $O/r04b_kitti128_poison.log | cut -c1-400
timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_128" > $O/r04b_kitti128_plain.log 2>&1
tail -12 $O/r04b_kitti128_plain.log | cut -c1-600
