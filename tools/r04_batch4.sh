#!/bin/bash
# Round 4, fourth call: candidate lists (cells) of chain mode — parity tests, cfg4 A/B, the big unions.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== chain tests"
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or nested or body_minus or staged_operator or largest" -o faulthandler_timeout=300 > $O/r04_cells_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r04_cells_pytest.txt
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_consumers.py -m gpu -x -q -k "fused" -o faulthandler_timeout=200 > $O/r04_cells_pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -3 $O/r04_cells_pytest2.txt
echo "== cfg4"
timeout -k 10 300 python3 tools/rows_ab.py --workload cfg4 --grid 16384 base NO_CELLS RWAVES=4 RWAVES=4+NO_CELLS base 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg4_cells.txt
echo "== unions"
for n in 200 1000 4096; do
  timeout -k 10 300 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp --json $O/r04_union${n}_513.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
  SDFK_CELLS=0 timeout -k 10 300 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp --json $O/r04_union${n}_513_nocells.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
done
echo "batch4 done"
