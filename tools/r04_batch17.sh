#!/bin/bash
# Round 4, seventeenth call: the final binary — full GPU suite, the default bench line, the N = 2 rehearsal of the closing sequence.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')" 2>&1 | tail -1
echo "== tests"; timeout -k 10 1100 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=500 > $O/r04_pytest_final.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/r04_pytest_final.txt | cut -c1-200
[ $rc -eq 0 ] || exit $rc
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
echo "== bench default"; ( time timeout -k 10 600 python3 bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err ) 2>&1 | grep real; echo "rc=$?"; cut -c1-400 $O/r04_bench_default.json
echo "== rehearsal N=2"; SDFK_BENCH_REHEARSE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 > $O/r04_rehearse_n2.json 2> $O/r04_rehearse_n2.err; echo "rc=$?"; cut -c1-300 $O/r04_rehearse_n2.json
echo "batch17 done"
