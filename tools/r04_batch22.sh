#!/bin/bash
# Round 4, twenty-second call: compiler scheduling strategies on the four kernel shapes (CodeGenPrepare turned out to be worth 12 % on
# cfg 5: is there more in the backend's options?).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
V="base extra=-mllvm,-amdgpu-sched-strategy=max-ilp:base extra=-mllvm,-amdgpu-sched-strategy=max-memory-clause:base extra=-mllvm,-enable-post-misched=0:base extra=-mllvm,-amdgpu-enable-max-ilp-scheduling-strategy:base base"
: > $O/r04_sched_sweep.txt
for w in "cfg5 1024" "cfg2 1024" "cfg4 16384" "cfg3 1024"; do
  set -- $w
  echo "== $1" | tee -a $O/r04_sched_sweep.txt; timeout -k 10 700 python3 tools/rows_ab.py --workload $1 --grid $2 $V 2>&1 | grep -v amdgpu.ids | cut -c1-170 | tee -a $O/r04_sched_sweep.txt
done
echo "batch22 done"
