#!/bin/bash
# Round 4, twenty-fourth call: a longer soak of every differential fuzzer on the final binary (new seeds).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
: > $O/r04_fuzz4_raw.txt
run () { echo "== ${*:2}" | tee -a $O/r04_fuzz4_raw.txt; timeout -k 10 "$1" "${@:2}" > $O/r04_fuzz_tmp.txt 2>&1; rc=$?; grep -v amdgpu.ids $O/r04_fuzz_tmp.txt | tail -1 | cut -c1-300 | tee -a $O/r04_fuzz4_raw.txt; echo "rc=$rc" | tee -a $O/r04_fuzz4_raw.txt; }
run 500 python tests/fuzz_mods.py gpu 91000 1200
run 400 python tests/fuzz_prims.py gpu 92000 900
run 400 python tests/fuzz_instancing.py 93000 100
run 500 python tests/fuzz_random_trees.py 94000 200 3
run 400 python tests/fuzz_staged.py 95000 120
run 400 python tests/fuzz_vector.py gpu 96000 600
run 300 python tests/fuzz_consumers.py 97000 80
run 400 python tests/fuzz_row_layouts.py 200
run 900 python tests/fuzz_chain_select.py 98000 100
echo "batch24 done"
