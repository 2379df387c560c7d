#!/bin/bash
# Round 4, thirteenth call: two-stage probe + single-plane block rows — A/B on cfg 5 / cfg 2 at both grid sizes, mask tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in 1024 512; do
  echo "== cfg5 $g"; timeout -k 10 500 python3 tools/rows_ab.py --workload cfg5 --grid $g base TWOSTAGE=0 REFINE_MIN=1 REFINE_MIN=2 REFINE_MIN=4 NOSIMT base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_twostage_cfg5.txt
  echo "== cfg2 $g"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg2 --grid $g base TWOSTAGE=0 REFINE_MIN=1 REFINE_MIN=2 base 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_twostage_cfg2.txt
done
echo "== mask / bit-exactness tests"
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mask or bit_exact or culling or row_block or brick" -o faulthandler_timeout=400 2>&1 | tail -3
echo "batch13 done"
