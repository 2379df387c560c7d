#!/bin/bash
# Round 4, third call: host-path breakdown, VALU-class counters available on gfx950, ablations of the cfg4 / cfg5 kernels.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== host path"; timeout -k 10 300 python3 tools/host_path_ab.py 96 > $O/r04_host_path_ab3.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/r04_host_path_ab3.txt | tail -60
echo "== counters"; ( cd /tmp && export TMPDIR=/tmp; rocprofv3 -L > $O/r04_counters_list.txt 2>&1 ); grep -c . $O/r04_counters_list.txt; grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*" $O/r04_counters_list.txt | sort -u | tr '\n' ' '
echo; echo "== cfg4 ablations"
timeout -k 10 400 python3 tools/rows_ab.py --workload cfg4 --grid 16384 base ABLATE_EVAL ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE+ABLATE_BOUNDS WPE=8 WPE=7 RWAVES=1 RWAVES=4 RWBRICKS=1 RWBRICKS=3 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg4_ablate.txt
echo "== cfg5 ablations"
timeout -k 10 400 python3 tools/rows_ab.py --workload cfg5 --grid 1024 base ABLATE_EVAL ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE+ABLATE_BOUNDS WPE=7 WPE=8 NSUB=4 RWBRICKS=4 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg5_ablate.txt
echo "batch3 done"
