#!/bin/bash
# Round 4, ninth call: gradient kernel in both process states with cache counters, nearest-point tree re-timed with the
# reference's own cloud, host path with the huge-page hint, fuzz soak of the chain / selection paths.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== gradient: 6 processes under TCC counters + kernel trace"
for i in 1 2 3 4 5 6; do
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/r04_grad_pmc/p$i -- python3 $R/tools/consumers_bench.py 1024 > $O/r04_grad_p$i.json 2> $O/r04_grad_p$i.log )
done
python3 - <<'PY'
import csv, glob, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
rows = []
for i in range(1, 7):
    d = O + "/r04_grad_pmc/p%d" % i
    cnt, dur = {}, []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gradient" in r["Kernel_Name"]:
                cnt.setdefault(r["Counter_Name"], []).append((r["Dispatch_Id"], float(r["Counter_Value"])))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gradient" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    per = {}
    for k, v in cnt.items():
        by = {}
        for did, val in v: by[did] = by.get(did, 0.0) + val
        per[k] = sum(by.values()) / max(1, len(by))
    rec = {"process": i, "gradient_ms_under_counters": round(sorted(dur)[len(dur) // 2], 3) if dur else None,
           "l2_hit_rate": round(per.get("TCC_HIT_sum", 0) / max(1.0, per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0)), 4),
           "TCC_MISS_M": round(per.get("TCC_MISS_sum", 0) / 1e6, 2), "EA_RDREQ_M": round(per.get("TCC_EA0_RDREQ_sum", 0) / 1e6, 2),
           "EA_WRREQ_M": round(per.get("TCC_EA0_WRREQ_sum", 0) / 1e6, 2)}
    rows.append(rec); print(rec)
json.dump(rows, open(O + "/r04_gradient_states.json", "w"), indent=1)
PY
echo "== neartree"; for g in 256 512; do timeout -k 10 300 python3 tools/neartree_bench.py $g 16384 terrain > $O/r04_neartree_$g.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/r04_neartree_$g.json')); print($g, {k: (round(v['ms'],2) if isinstance(v, dict) else v) for k, v in d.items() if k in ('box_tree','full_scan','bit_identical','speedup','cloud')})"; done
echo "== host path"; timeout -k 10 300 python3 tools/host_path_ab.py 96 2>&1 | grep -v "amdgpu.ids\|sdfk host" | head -12
echo "== fuzz"
run () { echo "== $*" | tee -a $O/r04_fuzz_raw.txt; timeout -k 10 "$1" "${@:2}" > $O/r04_fuzz_tmp.txt 2>&1; rc=$?; grep -v amdgpu.ids $O/r04_fuzz_tmp.txt | tail -3 | tee -a $O/r04_fuzz_raw.txt; echo "rc=$rc" | tee -a $O/r04_fuzz_raw.txt; }
run 500 python tests/fuzz_chain_select.py 51000 60
run 200 python tests/fuzz_random_trees.py 52000 80 3
run 150 python tests/fuzz_row_layouts.py 80
echo "batch9 done"
