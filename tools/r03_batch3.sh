#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== repro (must finish now)"; bash tools/diag/run_import_repro.sh 2>&1 | tail -8
echo "== pytest"; timeout -k 10 700 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=300 > $O/r03c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r03c_pytest.txt
for g in 512 1024; do timeout -k 10 200 python tools/rows_ab.py --grid $g --reps 20 base noplanes:base base noplanes:base 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 200 python tools/rows_ab.py --workload cfg5 --grid 1024 --reps 10 base noplanes:base 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/rows_ab.py --workload cfg5 --grid 512 --reps 20 base noplanes:base 2>&1 | grep -v amdgpu.ids
