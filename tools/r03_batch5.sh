#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -o faulthandler_timeout=400 -k "row_block or brick or masks or flat_baseline" > $O/r03e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03e_pytest.txt
timeout -k 10 400 python tools/rows_ab.py --workload cfg4 --grid 16384 --reps 10 base CHAIN_LOOP_MIN=999 base CHAIN_LOOP_MIN=999 NOSIMT CHAIN_LOOP_MIN=999+NOSIMT 2>&1 | grep -v amdgpu.ids
for g in 512 1024; do timeout -k 10 300 python tools/rows_ab.py --grid $g --reps 20 base CHAIN_LOOP_MIN=4 base CHAIN_LOOP_MIN=4 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 300 python tools/rows_ab.py --workload cfg5 --grid 1024 --reps 10 base CHAIN_LOOP_MIN=3 2>&1 | grep -v amdgpu.ids
