#!/bin/bash
# Round 4, sixteenth call: 4-wave workgroups as the default — XCD run lengths and bricks per wave around it, unions, tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== cfg4"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg4 --grid 16384 base XGROUP=2 XGROUP=4 XGROUP=16 RWBRICKS=1 RWBRICKS=3 RWBRICKS=4 RWAVES=2 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_w4_cfg4.txt
echo "== cfg2 1024"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg2 --grid 1024 base XGROUP=4 XGROUP=16 RWBRICKS=1 RWBRICKS=3 RWAVES=2 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_w4_cfg2.txt
echo "== cfg5 1024"; timeout -k 10 400 python3 tools/rows_ab.py --workload cfg5 --grid 1024 base XGROUP=4 XGROUP=16 RWBRICKS=1 RWBRICKS=3 NSUB=4 RWAVES=2 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_w4_cfg5.txt
echo "== unions"
for n in 200 1000 4096; do
  for v in "X=1" "SDFK_RTC_DEFS=-DSDFK_RWAVES=2"; do
    env $v timeout -k 10 200 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp 2>&1 | grep "^culled\|bit_identical" | cut -c1-60 | tr '\n' ' ' | sed "s/^/$n $v: /"; echo
  done
done
echo "== tests"; timeout -k 10 1000 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=500 > $O/r04_pytest_full2.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r04_pytest_full2.txt | cut -c1-200
echo "batch16 done"
