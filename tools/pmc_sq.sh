#!/bin/bash
# Developer tool (GPU box): instruction-mix counters of the bench kernel, two SQ passes + FETCH_SIZE.
# usage: tools/pmc_sq.sh <out-subdir-under-gpurun_out> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --no-host-path --no-next-rows $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --kernel-trace -f csv -d "$OUT/sq1" -- python "$R/bench.py" $ARGS > "$OUT/sq1.json" 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --kernel-trace -f csv -d "$OUT/sq2" -- python "$R/bench.py" $ARGS > "$OUT/sq2.json" 2> "$OUT/sq2.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/fetch" -- python "$R/bench.py" $ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
