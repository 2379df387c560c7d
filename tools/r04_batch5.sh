#!/bin/bash
# Round 4, fifth call: where the time of the candidate-list path goes (kernel trace), list sizes, the two-row (xy) path.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== xy test"; timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_row" -o faulthandler_timeout=200 2>&1 | tail -3
echo "== cfg4, lists off at run time (new fold, every member)"
SDFK_CELLS_MIN=64 timeout -k 10 300 python3 tools/rows_ab.py --workload cfg4 --grid 16384 base NO_CELLS 2>&1 | grep -v amdgpu.ids
echo "== kernel traces"
for n in 1000 4096; do
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof_union$n -- python3 $R/tools/big_union_bench.py --spheres $n --grid 512 --no-interp > /dev/null 2> $O/r04_prof_union$n.log )
  cp $O/r04_prof_union$n/*/*kernel_stats.csv $O/r04_union${n}_kernel_stats.csv 2>/dev/null
  echo "-- $n"; cut -d, -f1-4,7 $O/r04_union${n}_kernel_stats.csv | head -8
done
echo "== cell sizes (1000 spheres)"
for f in "3,1,0" "2,1,0" "3,0,0" "2,0,0" "4,1,0" "3,1,1"; do
  SDFK_CELL_FINE=$f timeout -k 10 200 python3 tools/big_union_bench.py --spheres 1000 --grid 512 --no-interp 2>&1 | grep "^culled" | cut -c1-60 | sed "s/^/fine $f: /"
done
SDFK_CELL_COARSE=off timeout -k 10 200 python3 tools/big_union_bench.py --spheres 1000 --grid 512 --no-interp 2>&1 | grep "^culled" | cut -c1-60 | sed "s/^/no coarse: /"
for c in "4,2,1" "5,3,2" "6,4,3"; do
  SDFK_CELL_COARSE=$c timeout -k 10 200 python3 tools/big_union_bench.py --spheres 4096 --grid 512 --no-interp 2>&1 | grep "^culled" | cut -c1-60 | sed "s/^/4096 coarse $c: /"
done
echo "batch5 done"
