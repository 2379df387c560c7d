#!/bin/bash
# Developer tool: per-kernel durations of one Python script under rocprofv3 (run on the GPU box).
#   tools/kernel_times.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>_kernel_stats.csv + a short table on stdout
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/kt_$tag
rm -rf "$out"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/$1" "${@:2}" > "$out.log" 2>&1 ) || { tail -5 "$out.log"; exit 1; }
f=$(ls "$out"/*/*kernel_stats.csv | head -1)
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s  avg %10.1f us  min %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
