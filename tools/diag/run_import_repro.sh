#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
gcc -O1 -g -shared -fPIC -rdynamic -o $O/bt_on_signal.so tools/diag/bt_on_signal.c || exit 1
for variant in "" warm; do
  for rep in 1 2; do
    log=$O/r03h_import_${variant:-cold}_$rep.txt
    python tools/diag/import_during_build.py $variant > $log 2>&1 &
    pid=$!
    for i in $(seq 1 45); do sleep 1; kill -0 $pid 2>/dev/null || break; done
    if kill -0 $pid 2>/dev/null; then
      echo "variant=${variant:-cold} rep=$rep: STUCK after 45 s -> native stacks"; python tools/diag/dump_threads.py $pid; sleep 2; kill -9 $pid
    else
      echo "variant=${variant:-cold} rep=$rep: $(tail -1 $log)"
    fi
  done
done
