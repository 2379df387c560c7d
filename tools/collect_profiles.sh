#!/bin/bash
# Developer tool (GPU box): the evidence set of one round. Writes gpurun_out/<tag>_*; copy what is to be judged to profiles/.
#   tools/collect_profiles.sh r02
set -u
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
B="python3 $R/bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0"
# 1. the driver's command, with every extra
python3 bench.py --steps 20 --warmup 5 > "$O/${tag}_cfg2_bench.json" 2> "$O/${tag}_cfg2_bench.err"
# 2. the same timed region under rocprofv3 (kernel trace + stats), then the counter passes (each in its own run)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof" -- $B > "$O/${tag}_cfg2_bench_under_rocprof.json" 2> "$O/${tag}_prof.log"
cp "$O/${tag}_prof"/*/*kernel_stats.csv "$O/${tag}_cfg2_kernel_stats.csv"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/${tag}_pmc/sq" -- $B > /dev/null 2> "$O/${tag}_pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_pmc/fetch" -- $B > /dev/null 2> "$O/${tag}_pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_pmc/write" -- $B > /dev/null 2> "$O/${tag}_pmc_write.log"
cd "$R"
python3 tools/pmc_summarize.py "$O/${tag}_pmc" sdfk_spec_r "$O/${tag}_cfg2_pmc_summary.json" > /dev/null
# 3. the other BASELINE configs on the same box (headline + verification only)
for w in cfg1 cfg3 cfg5; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > "$O/${tag}_${w}_bench.json" 2>/dev/null
done
python3 bench.py --workload cfg4 --grid 16384 --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg4_bench.json" 2>/dev/null
# the 2-D config's issue counters (what bounds it: VALU or memory)
C4="python3 $R/bench.py --workload cfg4 --grid 16384 --steps 20 --warmup 5 --no-extras --cpu-seconds 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_cfg4" -- $C4 > /dev/null 2> "$O/${tag}_prof_cfg4.log"
cp "$O/${tag}_prof_cfg4"/*/*kernel_stats.csv "$O/${tag}_cfg4_kernel_stats.csv"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/${tag}_pmc_cfg4/sq" -- $C4 > /dev/null 2> "$O/${tag}_pmc_cfg4_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_pmc_cfg4/fetch" -- $C4 > /dev/null 2> "$O/${tag}_pmc_cfg4_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_pmc_cfg4/write" -- $C4 > /dev/null 2> "$O/${tag}_pmc_cfg4_write.log"
cd "$R"
python3 tools/pmc_summarize.py "$O/${tag}_pmc_cfg4" sdfk_spec_r "$O/${tag}_cfg4_pmc_summary.json" > /dev/null
python3 bench.py --mode nocull --steps 10 --warmup 3 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg2_bench_nocull.json" 2>/dev/null
python3 bench.py --mode interpret --steps 5 --warmup 2 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg2_bench_interpreter.json" 2>/dev/null
echo "collected $tag"
