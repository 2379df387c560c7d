#!/bin/bash
# Developer tool (GPU box): VALU wave-instructions per launch of sdfk_spec_r for a list of kernel variants
# (tools/rows_ab.py variant syntax), one rocprofv3 counter pass per variant.
#   tools/pmc_valu.sh OUTDIR variant [variant ...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  d="$out/$(echo "$v" | tr '+=:' '___')"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$d" -- python3 "$R/tools/rows_ab.py" --reps 1 "$v" > "$d.log" 2>&1
  python3 "$R/tools/pmc_summarize.py" "$d" sdfk_spec_r | python3 -c "
import sys, json
r = json.load(sys.stdin)['per_launch']
print('%-50s VALU %.1fM  SALU %.1fM  LDS %.1fM  SMEM %.1fM' % ('$v', r['SQ_INSTS_VALU']/1e6, r['SQ_INSTS_SALU']/1e6, r['SQ_INSTS_LDS']/1e6, r['SQ_INSTS_SMEM']/1e6))"
done
