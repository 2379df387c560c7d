#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -o faulthandler_timeout=400 -k "row_block or brick or masks or baseline_size or flat_baseline or far_from or sharded or largest" > $O/r03d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03d_pytest.txt
for g in 512 1024; do timeout -k 10 300 python tools/rows_ab.py --grid $g --reps 20 base NOSIMT base NOSIMT NSUB=8 noplanes:NOSIMT 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 300 python tools/rows_ab.py --workload cfg4 --grid 16384 --reps 10 base NOSIMT base NOSIMT NSUB=8 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/rows_ab.py --workload cfg5 --grid 1024 --reps 10 base NOSIMT NSUB=16 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/rows_ab.py --workload cfg5 --grid 512 --reps 20 base NOSIMT NSUB=16 2>&1 | grep -v amdgpu.ids
for g in 512 1024; do timeout -k 10 120 python tools/row_mask_stats.py cfg2 $g 2>&1 | grep -v amdgpu.ids; done
