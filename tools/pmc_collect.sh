#!/bin/bash
# Developer tool (GPU box): rocprofv3 counter passes for one command, each counter set in its own run, CSV output.
#   tools/pmc_collect.sh OUTDIR -- python3 tools/rows_ab.py --reps 2 base
# Passes: SQ issue counters; FETCH_SIZE; WRITE_SIZE (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Never combined with --sys-trace / --hip-trace (gpurun refuses that combination).
set -u
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out.trace.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- "$@" > "$out.sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- "$@" > "$out.fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- "$@" > "$out.write.log" 2>&1
