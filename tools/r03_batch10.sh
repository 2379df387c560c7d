#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== example budget"; timeout -k 10 300 python tests/test_example_scenes.py --write-budget 2>&1 | grep -v amdgpu.ids; cp tests/golden/example_budget.json $O/example_budget.json
echo "== parity report + budget"; timeout -k 10 600 python tests/report_gpu_parity.py --write-budget > $O/r03_parity_report.txt 2>&1; tail -12 $O/r03_parity_report.txt; cp tests/golden/parity_budget.json $O/parity_budget.json
echo "== fuzz mods (the round-2 seed range and a fresh one)"; timeout -k 10 400 python tests/fuzz_mods.py gpu 70000 300 2>&1 | tail -4; timeout -k 10 400 python tests/fuzz_mods.py gpu 110000 300 2>&1 | tail -4
echo "== pytest"; timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=400 > $O/r03j_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r03j_pytest.txt
