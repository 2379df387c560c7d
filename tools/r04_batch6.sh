#!/bin/bash
# Round 4, sixth call: candidate lists, second version (records, prefetch, pool sizing) — tests, statistics, timings.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== tests"
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or nested or body_minus or staged_operator or thousands or candidate or two_row" -o faulthandler_timeout=400 > $O/r04_cells2_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r04_cells2_pytest.txt
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_consumers.py -m gpu -x -q -k "fused" -o faulthandler_timeout=200 2>&1 | tail -2
echo "== cfg4 (legacy path expected)"
timeout -k 10 300 python3 tools/rows_ab.py --workload cfg4 --grid 16384 base 2>&1 | grep -v amdgpu.ids
echo "== unions"
for n in 100 200 1000 4096 16384; do
  SDFK_CELLS_TRACE=1 timeout -k 10 400 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp --json $O/r04b_union${n}_513.json 2>&1 | grep -v amdgpu.ids | grep "sdfk cells\|^culled\|^plain\|bit_identical" | sort -u | cut -c1-330 | tail -4
done
echo "== kernel trace 1000 / 4096"
for n in 1000 4096; do
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04b_prof_union$n -- python3 $R/tools/big_union_bench.py --spheres $n --grid 512 --no-interp > /dev/null 2> $O/r04b_prof_union$n.log )
  cp $O/r04b_prof_union$n/*/*kernel_stats.csv $O/r04b_union${n}_kernel_stats.csv 2>/dev/null
  echo "-- $n"; cut -d, -f1-4 $O/r04b_union${n}_kernel_stats.csv | head -5
done
echo "batch6 done"
