#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest rows"; timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_block or brick or sharded or baseline_size or masks" > $O/r03b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03b_pytest.txt
for g in 512 1024; do timeout -k 10 200 python tools/rows_ab.py --grid $g --reps 20 base noplanes:base base noplanes:base 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 200 python tools/rows_ab.py --workload cfg5 --grid 1024 --reps 10 base noplanes:base 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/rows_ab.py --workload cfg5 --grid 512 --reps 20 base noplanes:base 2>&1 | grep -v amdgpu.ids
