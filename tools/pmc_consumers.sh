#!/bin/bash
# Developer tool (GPU box): L2-miss read / write bytes of the field-consumer kernels (separate PMC passes, no tracing
# domains besides the kernel trace).   usage: tools/pmc_consumers.sh <out-subdir-under-gpurun_out> [resolution]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1
RES=${2:-1024}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/fetch" -- python "$R/tools/consumers_bench.py" $RES > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/write" -- python "$R/tools/consumers_bench.py" $RES > "$OUT/write.json" 2> "$OUT/write.err"
for k in sdfk_gradient_kernel sdfk_select_count_kernel sdfk_select_scatter_kernel; do
    python "$R/tools/pmc_summarize.py" "$OUT" $k "$OUT/pmc_$k.json" > /dev/null
done
cat "$OUT"/pmc_*.json
