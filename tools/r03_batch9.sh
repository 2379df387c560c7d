#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest chain"; timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -o faulthandler_timeout=300 -k "chain_mode or very_large" > $O/r03i_pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/r03i_pytest.txt
echo "== big union"; timeout -k 10 500 python tools/big_union_bench.py --spheres 1000 --grid 512 --json $O/r03i_union1000_513.json 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/big_union_bench.py --spheres 200 --grid 512 --json $O/r03i_union200_513.json 2>&1 | grep -v amdgpu.ids | tail -1
echo "== cfg4 in chain mode (forced) vs specialised"; SDFK_CHAIN_MIN=8 timeout -k 10 300 python tools/rows_ab.py --workload cfg4 --grid 16384 --reps 10 base 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/rows_ab.py --workload cfg4 --grid 16384 --reps 10 base 2>&1 | grep -v amdgpu.ids
