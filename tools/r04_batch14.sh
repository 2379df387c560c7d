#!/bin/bash
# Round 4, fourteenth call: does the probe need the other wave? one-wave workgroups (no cross-wave barrier at all).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in 1024 512; do
  echo "== cfg5 $g"; timeout -k 10 500 python3 tools/rows_ab.py --workload cfg5 --grid $g base RWAVES=1 RWAVES=1+RWBRICKS=4 RWAVES=1+RWBRICKS=1 RWAVES=4 RWAVES=1+NSUB=4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_rwaves_cfg5.txt
  echo "== cfg2 $g"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg2 --grid $g base RWAVES=1 RWAVES=1+RWBRICKS=4 RWAVES=4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_rwaves_cfg2.txt
done
echo "batch14 done"
