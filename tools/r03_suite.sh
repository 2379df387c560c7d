#!/bin/bash
# full GPU suite + driver-style bench (one line each into gpurun_out/<tag>_*)
set -u
tag=${1:-r03x}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=400 > $O/${tag}_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/${tag}_pytest.txt
echo "== bench"; timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$O/${tag}_bench.json") if l.startswith("{")][-1])
print("headline", round(d["ms_per_step"],3), round(d["roofline"]["frac"],3), "grid512", round(d["grid_512"]["roofline_frac"],3), round(d["grid_512"]["kernel_ms_min"],3))
for k,v in d.get("other_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("roofline_frac"), v.get("verified"), v.get("error"))
print(d.get("next_rows")); print(d.get("first_call")); print(d.get("extras_s"))
PY
