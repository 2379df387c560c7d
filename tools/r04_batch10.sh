#!/bin/bash
# Round 4, tenth call: the libraries of rounds 2, 3 and 4 on ONE box (cfg 5, cfg 2), the gradient kernel's process states
# without a profiler, the whole GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== versions cfg5"; timeout -k 10 500 python3 tools/version_ab.py cfg5 3 2>&1 | grep -v amdgpu.ids | tee $O/r04_version_ab_cfg5.txt
echo "== versions cfg2"; timeout -k 10 400 python3 tools/version_ab.py cfg2 2 2>&1 | grep -v amdgpu.ids | tee $O/r04_version_ab_cfg2.txt
echo "== gradient, 8 plain processes"
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 100 python3 tools/consumers_bench.py 1024 2>/dev/null | python3 -c "
import sys, json; d = json.load(sys.stdin); print('process $i gradient_direction %.3f ms  raw %.3f ms' % (d['gradient_direction']['ms'], d['gradient_raw']['ms']))"; done | tee $O/r04_gradient_plain_states.txt
echo "== pytest -m gpu"; timeout -k 10 1000 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=500 > $O/r04_pytest_full.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r04_pytest_full.txt | cut -c1-300
echo "batch10 done"
