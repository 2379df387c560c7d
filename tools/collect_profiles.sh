#!/bin/bash
# Developer tool (GPU box): everything profiles/ quotes for the headline workload, in one call.
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python bench.py > "$OUT/cfg2_bench.json" 2> "$OUT/cfg2_bench.err"
python bench.py --no-rows --cpu-seconds 0 > "$OUT/cfg2_bench_line_bricks.json" 2>> "$OUT/cfg2_bench.err"
python bench.py --mode nocull --cpu-seconds 0 > "$OUT/cfg2_bench_nocull.json" 2>> "$OUT/cfg2_bench.err"
python bench.py --workload cfg1 --cpu-seconds 0 > "$OUT/cfg1_bench.json" 2>> "$OUT/cfg2_bench.err"
python bench.py --workload cfg3 --cpu-seconds 0 > "$OUT/cfg3_bench.json" 2>> "$OUT/cfg2_bench.err"
python bench.py --workload cfg5 --cpu-seconds 0 > "$OUT/cfg5_bench.json" 2>> "$OUT/cfg2_bench.err"
python bench.py --workload cfg4 --grid 16384 --cpu-seconds 0 > "$OUT/cfg4_bench.json" 2>> "$OUT/cfg2_bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -- python "$R/bench.py" --steps 20 --warmup 3 --cpu-seconds 0 --no-host-path --no-next-rows > "$OUT/cfg2_bench_under_rocprof.json" 2> "$OUT/stats.err"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/cfg2_kernel_stats.csv"
bash "$R/tools/pmc_collect.sh" "$TAG/pmc"
python "$R/tools/pmc_summarize.py" "$OUT/pmc" sdfk_spec_r "$OUT/cfg2_pmc_summary.json" > /dev/null
echo done
