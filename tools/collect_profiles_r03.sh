#!/bin/bash
# Developer tool (GPU box): the evidence set of round 3. Writes gpurun_out/r03_*; copy what is to be judged to profiles/.
#   collect_profiles_r03.sh [a|g|b|c|all]   a: bench lines + counters of cfg2;  g: counters of the 513^3 grid;  b: counters of cfg3 / cfg5 / cfg4 + the other
#   bench lines;  c: consumers, fused selection, big unions, mask statistics  (one gpurun call of <= 20 minutes each)
set -u
tag=r03
part=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
pmc_set () {   # pmc_set <name> <kernel> <command...>: kernel-trace stats + the three counter passes of one command
  local name=$1 kernel=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_$name" -- "$@" > /dev/null 2> "$O/${tag}_prof_$name.log"
    rocprofv3 --pmc $SQ --output-format csv -d "$O/${tag}_pmc_$name/sq" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_sq.log"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_pmc_$name/fetch" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_fetch.log"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_pmc_$name/write" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_write.log"
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/${tag}_pmc_$name/tcc" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_tcc.log" )
  cp "$O/${tag}_prof_$name"/*/*kernel_stats.csv "$O/${tag}_${name}_kernel_stats.csv" 2>/dev/null
  python3 tools/pmc_summarize.py "$O/${tag}_pmc_$name" "$kernel" "$O/${tag}_${name}_pmc_summary.json" > /dev/null
  echo "pmc $name done"
}
B="--steps 20 --warmup 5 --no-extras --cpu-seconds 0"
if [ "$part" = a ] || [ "$part" = all ]; then
# 1. the driver's command, with every extra
python3 bench.py --steps 20 --warmup 5 > "$O/${tag}_cfg2_bench.json" 2> "$O/${tag}_cfg2_bench.err"; echo "bench rc=$?"
python3 bench.py $B > "$O/${tag}_cfg2_bench_noextras.json" 2>/dev/null
# 2. counters: the headline workload, the 513^3 grid, and the other BASELINE configs
pmc_set cfg2 sdfk_spec_r python3 $R/bench.py $B
fi
if [ "$part" = g ] || [ "$part" = all ]; then
# (a 0.4 ms kernel: 200 launches after 50 of warm-up, or the averages are those of a GPU still raising its clocks)
pmc_set grid512 sdfk_spec_r python3 $R/bench.py --grid 512 --steps 200 --warmup 50 --no-extras --cpu-seconds 0
python3 bench.py --grid 512 --steps 200 --warmup 50 --no-extras --cpu-seconds 0 > "$O/${tag}_grid512_bench.json" 2>/dev/null
fi
if [ "$part" = b ] || [ "$part" = all ]; then
pmc_set cfg3 sdfk_spec_v4 python3 $R/bench.py --workload cfg3 $B
pmc_set cfg5 sdfk_spec_r python3 $R/bench.py --workload cfg5 $B
pmc_set cfg4 sdfk_spec_r python3 $R/bench.py --workload cfg4 --grid 16384 --steps 100 --warmup 30 --no-extras --cpu-seconds 0
for w in cfg1 cfg3 cfg5; do python3 bench.py --workload $w $B > "$O/${tag}_${w}_bench.json" 2>/dev/null; done
python3 bench.py --workload cfg4 --grid 16384 --steps 100 --warmup 30 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg4_bench.json" 2>/dev/null
python3 bench.py --mode nocull --steps 10 --warmup 3 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg2_bench_nocull.json" 2>/dev/null
python3 bench.py --mode interpret --steps 5 --warmup 2 --no-extras --cpu-seconds 0 > "$O/${tag}_cfg2_bench_interpreter.json" 2>/dev/null
fi
if [ "$part" = c ] || [ "$part" = all ]; then
# 3. the consumers of the field
python3 tools/consumers_bench.py 1024 > "$O/${tag}_consumers_1025.json" 2>/dev/null
( cd /tmp && export TMPDIR=/tmp
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "$SQ"; do
    d=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/${tag}_pmc_consumers/$d" -- python3 $R/tools/consumers_bench.py 1024 > /dev/null 2> "$O/${tag}_pmc_consumers_$d.log"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_consumers" -- python3 $R/tools/consumers_bench.py 1024 > /dev/null 2> "$O/${tag}_prof_consumers.log" )
cp "$O/${tag}_prof_consumers"/*/*kernel_stats.csv "$O/${tag}_consumers_kernel_stats.csv" 2>/dev/null
for k in "~sdfk_gradient" "~sdfk_select_count_kernel" sdfk_flags_scatter_kernel; do
  python3 tools/pmc_summarize.py "$O/${tag}_pmc_consumers" "$k" "$O/${tag}_consumers_pmc_$(echo $k | tr -d '~').json" > /dev/null
done
# 3b. point_cloud without the field: the evaluation kernel writes flag bits, compaction from the flags
python3 tools/fused_select_bench.py 1024 cfg2 2>/dev/null > "$O/${tag}_fused_select_1025.txt"
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_fused" -- python3 $R/tools/fused_select_bench.py 1024 cfg2 > /dev/null 2> "$O/${tag}_prof_fused.log" )
cp "$O/${tag}_prof_fused"/*/*kernel_stats.csv "$O/${tag}_fused_select_kernel_stats.csv" 2>/dev/null
# 4. big n-ary unions, mask statistics
python3 tools/big_union_bench.py --spheres 1000 --grid 512 --json "$O/${tag}_union1000_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 200 --grid 512 --json "$O/${tag}_union200_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 1000 --groups 20 --grid 512 --json "$O/${tag}_clusters1000_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --body --grid 512 --json "$O/${tag}_porous500_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --clip --grid 512 --json "$O/${tag}_clipped500_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --blend --grid 512 --json "$O/${tag}_blended500_513.json" > /dev/null 2>&1
for g in 512 1024; do python3 tools/row_mask_stats.py cfg2 $g 2>&1 | grep -v amdgpu.ids; done > "$O/${tag}_row_mask_stats.txt"
python3 tools/row_mask_stats.py cfg5 1024 2>&1 | grep -v amdgpu.ids >> "$O/${tag}_row_mask_stats.txt"
fi
echo "collected $tag $part"
