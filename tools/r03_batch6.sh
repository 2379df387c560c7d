#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 500 python tools/rows_ab.py --workload cfg4 --grid 16384 --reps 10 base RWBRICKS=2 RWBRICKS=3 RWBRICKS=2+RWAVES=4 RWBRICKS=1+RWAVES=4 RWBRICKS=2+RWAVES=1 ABLATE_EVAL ABLATE_EVAL+ABLATE_PROBE RWBRICKS=2+ABLATE_EVAL base 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/rows_ab.py --grid 512 --reps 30 base ABLATE_EVAL ABLATE_EVAL+NOSIMT NOSIMT RWBRICKS=1 RWBRICKS=1+RWAVES=4 RWBRICKS=2+RWAVES=4 RWBRICKS=4+RWAVES=1 base 2>&1 | grep -v amdgpu.ids
