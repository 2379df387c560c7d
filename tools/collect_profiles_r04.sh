#!/bin/bash
# Developer tool (GPU box): the evidence set of round 4. Writes gpurun_out/r04e_*; copy what is to be judged to profiles/.
#   collect_profiles_r04.sh [a|b|c|cfg3|consumers|all]   a: the driver's command + counters of cfg2 and the 513^3 grid;  b: counters and bench
#   lines of cfg3 / cfg5 / cfg4 / cfg1;  c: unions (kernel trace + list statistics), consumers, fused selection
set -u
tag=r04e
part=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
CLS="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"
pmc_set () {   # pmc_set <name> <kernel> <command...>: kernel-trace stats + the counter passes of one command (each set in its own run)
  local name=$1 kernel=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_$name" -- "$@" > /dev/null 2> "$O/${tag}_prof_$name.log"
    rocprofv3 --pmc $SQ --output-format csv -d "$O/${tag}_pmc_$name/sq" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_sq.log"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_pmc_$name/fetch" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_fetch.log"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_pmc_$name/write" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_write.log"
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/${tag}_pmc_$name/tcc" -- "$@" > /dev/null 2> "$O/${tag}_pmc_${name}_tcc.log"
    rocprofv3 --pmc $CLS --output-format csv -d "$O/${tag}_cls_$name" -- "$@" > /dev/null 2> "$O/${tag}_cls_$name.log" )
  cp "$O/${tag}_prof_$name"/*/*kernel_stats.csv "$O/${tag}_${name}_kernel_stats.csv" 2>/dev/null
  python3 tools/pmc_summarize.py "$O/${tag}_pmc_$name" "$kernel" "$O/${tag}_${name}_pmc_summary.json" > /dev/null
  python3 tools/valu_roof.py summarize "$O/${tag}_cls_$name" "$kernel" "$O/${tag}_${name}_valu_roof.json" > /dev/null
  echo "pmc $name done"
}
B="--steps 20 --warmup 5 --no-extras --cpu-seconds 0"
if [ "$part" = a ] || [ "$part" = all ]; then
python3 bench.py --steps 20 --warmup 5 > "$O/${tag}_cfg2_bench.json" 2> "$O/${tag}_cfg2_bench.err"; echo "bench rc=$?"
python3 bench.py $B > "$O/${tag}_cfg2_bench_noextras.json" 2>/dev/null
pmc_set cfg2 sdfk_spec_r python3 $R/bench.py $B
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
if [ "$part" = cfg3 ]; then                                   # after the instruction diet of cfg 3: its own set again
pmc_set cfg3 sdfk_spec_v4 python3 $R/bench.py --workload cfg3 $B
python3 bench.py --workload cfg3 $B > "$O/${tag}_cfg3_bench.json" 2>/dev/null
fi
if [ "$part" = consumers ]; then                              # after the gradient kernel's new run length
python3 tools/consumers_bench.py 1024 > "$O/${tag}_consumers_1025.json" 2>/dev/null
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_consumers" -- python3 $R/tools/consumers_bench.py 1024 > /dev/null 2> "$O/${tag}_prof_consumers.log" )
cp "$O/${tag}_prof_consumers"/*/*kernel_stats.csv "$O/${tag}_consumers_kernel_stats.csv" 2>/dev/null
fi
if [ "$part" = c ] || [ "$part" = all ]; then
for n in 200 1000 4096 16384; do
  SDFK_CELLS_TRACE=1 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp --json "$O/${tag}_union${n}_513.json" 2>&1 | grep "sdfk cells" | sort -u | tail -1 > "$O/${tag}_union${n}_lists.txt"
done
python3 tools/big_union_bench.py --spheres 1000 --groups 20 --grid 512 --json "$O/${tag}_clusters1000_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --body --grid 512 --json "$O/${tag}_porous500_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --clip --grid 512 --json "$O/${tag}_clipped500_513.json" > /dev/null 2>&1
python3 tools/big_union_bench.py --spheres 500 --blend --grid 512 --json "$O/${tag}_blended500_513.json" > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_union1000" -- python3 $R/tools/big_union_bench.py --spheres 1000 --grid 512 --no-interp > /dev/null 2> "$O/${tag}_prof_union1000.log"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_union4096" -- python3 $R/tools/big_union_bench.py --spheres 4096 --grid 512 --no-interp > /dev/null 2> "$O/${tag}_prof_union4096.log" )
cp "$O/${tag}_prof_union1000"/*/*kernel_stats.csv "$O/${tag}_union1000_kernel_stats.csv" 2>/dev/null
cp "$O/${tag}_prof_union4096"/*/*kernel_stats.csv "$O/${tag}_union4096_kernel_stats.csv" 2>/dev/null
python3 tools/consumers_bench.py 1024 > "$O/${tag}_consumers_1025.json" 2>/dev/null
python3 tools/fused_select_bench.py 1024 cfg2 2>/dev/null > "$O/${tag}_fused_select_1025.txt"
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${tag}_prof_fused" -- python3 $R/tools/fused_select_bench.py 1024 cfg2 > /dev/null 2> "$O/${tag}_prof_fused.log" )
cp "$O/${tag}_prof_fused"/*/*kernel_stats.csv "$O/${tag}_fused_select_kernel_stats.csv" 2>/dev/null
fi
echo "collected $tag $part"
