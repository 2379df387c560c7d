#!/bin/bash
# Round 4, eleventh call: what made cfg 5 slower between rounds 2 and 3 (same-box A/B: 3.12 vs 3.33 ms) — probe variants.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in 1024 512; do
  echo "== cfg5 $g"; timeout -k 10 500 python3 tools/rows_ab.py --workload cfg5 --grid $g base NOSIMT NSUB=1 NSUB=4 NSUB=16 NOHOIST base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_cfg5_probe_variants.txt
  echo "== cfg2 $g"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg2 --grid $g base NOSIMT NSUB=4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_cfg2_probe_variants.txt
done
echo "batch11 done"
