#!/bin/bash
# Developer tool (GPU box): what closes round 3 — N = 2 / 4 rehearsal of bench.py on one GPU, the parity report, and a soak of
# every differential fuzzer on fresh seed ranges (last line of each into gpurun_out/r03_fuzz_raw.txt).
#   r03_final_checks.sh [rehearse|parity|fuzz1|fuzz2|all]
set -u
part=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run () { echo "== $*" >> $O/r03_fuzz_raw.txt; timeout -k 10 "$1" "${@:2}" 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/r03_fuzz_raw.txt; echo "rc=${PIPESTATUS[0]}" >> $O/r03_fuzz_raw.txt; echo "done: ${*:2}"; }
if [ "$part" = rehearse ] || [ "$part" = all ]; then
  for n in 2 4; do
    SDFK_BENCH_REHEARSE=1 timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $((29520 + n)) bench.py --gpus $n --steps 5 --warmup 2 > $O/r03_rehearse_n$n.json 2> $O/r03_rehearse_n$n.err
    echo "rehearse n=$n rc=$?"
  done
fi
if [ "$part" = parity ] || [ "$part" = all ]; then
  timeout -k 10 600 python tests/report_gpu_parity.py > $O/r03_parity_report.txt 2>&1; echo "parity rc=$?"; tail -3 $O/r03_parity_report.txt
fi
if [ "$part" = fuzz1 ] || [ "$part" = all ]; then
  run 400 python tests/fuzz_random_trees.py 41000 200 3
  run 300 python tests/fuzz_random_trees.py 42000 80 4
  run 300 python tests/fuzz_row_layouts.py 200
  run 1000 python tests/fuzz_chain_select.py 43000 70
fi
if [ "$part" = fuzz2 ] || [ "$part" = all ]; then
  run 400 python tests/fuzz_prims.py gpu 44000 400
  run 400 python tests/fuzz_mods.py gpu 45000 400
  run 300 python tests/fuzz_staged.py 46000 40
  run 300 python tests/fuzz_consumers.py 47000 200
  run 400 python tests/fuzz_vector.py gpu 48000 300
  run 300 python tests/fuzz_instancing.py 49000 60
fi
echo "final checks $part done"
