#!/bin/bash
# Developer tool (GPU box): collect PMC counters for the bench kernel in separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/runtime tracing).
# usage: tools/pmc_collect.sh <out-subdir-under-gpurun_out> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --no-host-path --no-next-rows $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/fetch" -- python "$R/bench.py" $ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/write" -- python "$R/bench.py" $ARGS > "$OUT/write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -f csv -d "$OUT/sq" -- python "$R/bench.py" $ARGS > "$OUT/sq.json" 2> "$OUT/sq.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -f csv -d "$OUT/grbm" -- python "$R/bench.py" $ARGS > "$OUT/grbm.json" 2> "$OUT/grbm.err"
find "$OUT" -name "*counter_collection.csv" | head
