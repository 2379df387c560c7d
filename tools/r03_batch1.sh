#!/bin/bash
# round-3 GPU batch 1: suite, driver-style bench, N=2 rehearsal, A/B ablations at 513 and 1025, mask statistics
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03a_pytest.txt
echo "== bench N=1"; timeout -k 10 420 python bench.py --steps 20 --warmup 5 > $O/r03a_bench.json 2> $O/r03a_bench.err; echo "bench rc=$?"
echo "== rehearse N=2"; SDFK_BENCH_REHEARSE=1 timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r03a_rehearse_n2.json 2> $O/r03a_rehearse_n2.err; echo "rehearse rc=$?"
echo "== A/B 513"; timeout -k 10 300 python tools/rows_ab.py --grid 512 --reps 20 --json $O/r03a_ab_513.json base ABLATE_EVAL ABLATE_PROBE ABLATE_EDGE ABLATE_EVAL+ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE+ABLATE_BOUNDS mode=nocull:base > $O/r03a_ab_513.txt 2>&1; echo "ab513 rc=$?"
echo "== A/B 1025"; timeout -k 10 300 python tools/rows_ab.py --grid 1024 --reps 10 --json $O/r03a_ab_1025.json base ABLATE_EVAL ABLATE_PROBE ABLATE_EDGE ABLATE_EVAL+ABLATE_PROBE ABLATE_EVAL+ABLATE_PROBE+ABLATE_BOUNDS > $O/r03a_ab_1025.txt 2>&1; echo "ab1025 rc=$?"
echo "== masks"; for g in 512 1024; do timeout -k 10 120 python tools/row_mask_stats.py cfg2 $g >> $O/r03a_masks.txt 2>&1; done; timeout -k 10 120 python tools/row_mask_stats.py cfg4 16384 >> $O/r03a_masks.txt 2>&1; timeout -k 10 120 python tools/row_mask_stats.py cfg5 1024 >> $O/r03a_masks.txt 2>&1
cat $O/r03a_ab_513.txt $O/r03a_ab_1025.txt $O/r03a_masks.txt
