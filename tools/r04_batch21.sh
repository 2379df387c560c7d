#!/bin/bash
# Round 4, twenty-first call: the build tiers on the GPU box — full GPU suite, the chain / selection fuzzer (its big programs now get
# specialised kernels), random trees.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
echo "== tests"; timeout -k 10 1100 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=500 > $O/r04_pytest_tiers.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r04_pytest_tiers.txt | cut -c1-200
[ $rc -eq 0 ] || exit $rc
run () { echo "== ${*:2}"; timeout -k 10 "$1" "${@:2}" > $O/r04_fuzz_tmp.txt 2>&1; rc=$?; grep -v amdgpu.ids $O/r04_fuzz_tmp.txt | tail -2 | cut -c1-300; echo "rc=$rc"; }
run 600 python tests/fuzz_chain_select.py 71000 40 | tee $O/r04_fuzz3_raw.txt
run 300 python tests/fuzz_random_trees.py 72000 60 3 | tee -a $O/r04_fuzz3_raw.txt
echo "batch21 done"
