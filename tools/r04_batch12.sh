#!/bin/bash
# Round 4, twelfth call: bisecting the cfg 5 slow-down between rounds 2 and 3 over library snapshots of round-3 commits.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1000 python3 tools/version_ab.py cfg5 2 r02 c_bf2936c c_8de6360 c_25b2139 c_608e507 c_ee14a14 r03 2>&1 | grep -v amdgpu.ids | tee $O/r04_bisect_cfg5.txt
echo "batch12 done"
