run() { local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
  echo "== ${envs[*]} $*"; env "${envs[@]}" timeout -k 10 250 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.3f frac %.3f grid_ms %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['grid_path']['ms']))"; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_block or brick or sharded or baseline_size" 2>&1 | tail -5
run A=1
run SDFK_RTC_DEFS="-DSDFK_ABLATE_EVAL -DSDFK_ABLATE_PROBE"
run A=1 -- --workload cfg5
run A=1 -- --workload cfg4 --grid 16384
