#!/bin/bash
# Round 4, nineteenth call: differential fuzzers on the binary with the device-function diet (sd_mod, sd_sincos, op_bend) and
# the three-level point tree — new seeds.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
: > $O/r04_fuzz2_raw.txt
run () { echo "== ${*:2}" | tee -a $O/r04_fuzz2_raw.txt; timeout -k 10 "$1" "${@:2}" > $O/r04_fuzz_tmp.txt 2>&1; rc=$?; grep -v amdgpu.ids $O/r04_fuzz_tmp.txt | tail -3 | cut -c1-300 | tee -a $O/r04_fuzz2_raw.txt; echo "rc=$rc" | tee -a $O/r04_fuzz2_raw.txt; }
run 300 python tests/fuzz_mods.py gpu 61000 400
run 200 python tests/fuzz_prims.py gpu 62000 300
run 200 python tests/fuzz_instancing.py 63000 40
run 200 python tests/fuzz_random_trees.py 64000 60 3
run 200 python tests/fuzz_staged.py 65000 40
run 200 python tests/fuzz_vector.py gpu 66000 200
run 200 python tests/fuzz_consumers.py 67000 30
echo "batch19 done"
