#!/bin/bash
# Round 4, eighteenth call: cfg 3 after the VALU diet of sd_mod / sd_sincos / op_bend (A/B against the previous library on one box), then the GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
echo "== cfg3 A/B"; timeout -k 10 500 python3 tools/version_ab.py cfg3 3 c_3c916b5 now 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg3_valu_diet.txt
echo "== tests"; timeout -k 10 1000 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=500 > $O/r04_pytest_diet.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r04_pytest_diet.txt | cut -c1-200
echo "batch18 done"
