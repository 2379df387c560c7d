#!/bin/bash
# Round 4, twentieth call: kernels built without LLVM's CodeGenPrepare pass (it is what makes hiprtc's time grow with the square of
# the program: 458 of 630 s on a 599-instruction program) — do they run as fast?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
X="extra=-mllvm,-disable-cgp:base"
for w in "cfg2 1024" "cfg5 1024" "cfg4 16384" "cfg3 1024" "cfg1 1024"; do
  set -- $w
  echo "== $1"; timeout -k 10 500 python3 tools/rows_ab.py --workload $1 --grid $2 base $X base $X 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_nocgp.txt
done
echo "== union 1000"; for v in "" "-mllvm -disable-cgp"; do SDFK_RTC_EXTRA="$v" timeout -k 10 200 python3 tools/big_union_bench.py --spheres 1000 --grid 512 --no-interp 2>&1 | grep "^culled\|bit_identical" | cut -c1-70 | tr '\n' ' '; echo " [$v]"; done | tee -a $O/r04_nocgp.txt
echo "batch20 done"
