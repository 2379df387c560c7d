#!/bin/bash
# Round 4, fifteenth call: waves per workgroup of the row-block kernel, all BASELINE configs that use it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for w in "cfg2 1024" "cfg5 1024" "cfg2 512" "cfg5 512" "cfg4 16384" "cfg5 2048"; do set -- $w
  echo "== $1 $2"; timeout -k 10 600 python3 tools/rows_ab.py --workload $1 --grid $2 base RWAVES=3 RWAVES=4 RWAVES=6 RWAVES=8 RWAVES=4+XGROUP=16 RWAVES=4+XGROUP=4 base 2>&1 | grep -v amdgpu.ids | sed "s/^/$1 $2: /" | tee -a $O/r04_rwaves_sweep.txt
done
echo "batch15 done"
