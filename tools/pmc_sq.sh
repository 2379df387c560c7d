#!/bin/bash
# Developer tool (GPU box): ONE rocprofv3 counter pass (SQ issue counters) for a command, CSV output.
#   tools/pmc_sq.sh OUTDIR -- python3 tools/rows_ab.py --reps 2 base
set -u
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$out/sq" -- "$@" > "$out.sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d "$out/sq2" -- "$@" > "$out.sq2.log" 2>&1
