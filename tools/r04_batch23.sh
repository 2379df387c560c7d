#!/bin/bash
# Round 4, twenty-third call: the gradient kernel's scheduling knobs in the SLOW process state. Under rocprofv3 every process is in
# that state (profiles/r04_gradient_states.json), so the kernel trace of one process per setting compares the settings there.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
: > $O/r04_gradient_knobs.txt
one () {   # one <label> : env already set
  local label=$1 d=$O/r04_prof_grad_$1
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 $R/tools/consumers_bench.py 1024 > /dev/null 2> "$d.log" )
  local line=$(python3 - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "sdfk_gradient_carry_kernel" in r["Name"]:
            kind = "direction" if "true>" in r["Name"] else "raw"
            print("%s %s calls avg %.3f min %.3f ms;" % (kind, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6), end=" ")
PY
)
  echo "$label: $line" | tee -a $O/r04_gradient_knobs.txt
  rm -rf "$d"
}
for g in 8 0 1 2 4 16 32 64; do export SDFK_GC_GROUP=$g; unset SDFK_GC_SEG; one "group_$g"; done
export SDFK_GC_GROUP=8
for s in 8 16 64; do export SDFK_GC_SEG=$s; one "seg_$s"; done
unset SDFK_GC_SEG SDFK_GC_GROUP
echo "== un-profiled, three processes (which state?)" | tee -a $O/r04_gradient_knobs.txt
for i in 1 2 3; do timeout -k 10 120 python3 tools/consumers_bench.py 1024 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('gradient_direction', d['gradient_direction']['ms'], 'gradient_raw', d['gradient_raw']['ms'])" | tee -a $O/r04_gradient_knobs.txt; done
echo "batch23 done"
