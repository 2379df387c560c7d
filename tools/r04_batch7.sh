#!/bin/bash
# Round 4, seventh call: candidate lists after the out-of-range fix, sharded pool heads, list-mode overflow.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== tests"
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or nested or body_minus or staged_operator or thousands or candidate or two_row" -o faulthandler_timeout=400 > $O/r04_cells3_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r04_cells3_pytest.txt | cut -c1-300
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_consumers.py tests/test_example_scenes.py -m gpu -x -q -k "fused or terrain" -o faulthandler_timeout=200 2>&1 | tail -2
echo "== unions"
for n in 100 200 1000 4096 16384; do
  SDFK_CELLS_TRACE=1 timeout -k 10 400 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp --json $O/r04c_union${n}_513.json 2>&1 | grep -v amdgpu.ids | grep "sdfk cells\|^culled\|bit_identical" | sort -u | cut -c1-250 | tail -3
done
echo "== variants (1000 / 4096)"
for n in 1000 4096; do
  for v in "SDFK_RWBRICKS=2" "SDFK_RTC_DEFS=-DSDFK_STAGE_MIN=8" "SDFK_RTC_DEFS=-DSDFK_STAGE_MIN=64" "SDFK_RTC_DEFS=-DSDFK_CHUNK=32" "SDFK_CELL_FINE=4,1,0" "SDFK_CELL_FINE=2,1,0"; do
    env $v timeout -k 10 200 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp 2>&1 | grep "^culled" | cut -c1-50 | sed "s/^/$n $v: /"
  done
done
echo "== kernel trace 1000"
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04c_prof_union1000 -- python3 $R/tools/big_union_bench.py --spheres 1000 --grid 512 --no-interp > /dev/null 2> $O/r04c_prof_union1000.log )
cut -d, -f1-4 $O/r04c_prof_union1000/*/*kernel_stats.csv | head -5
echo "batch7 done"
