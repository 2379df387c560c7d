# Developer tool (GPU box): one bench line per environment variation.
#   run VAR=value ... [-- bench.py arguments]
# e.g. run SDFK_NP=2 SDFK_RWBRICKS=4 ; run SDFK_RTC_DEFS="-DSDFK_ABLATE_EVAL -DSDFK_ABLATE_PROBE" ; run A=1 -- --workload cfg5
run() { local envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" = "--" ] && shift
  echo "== ${envs[*]} $*"; env "${envs[@]}" timeout -k 10 250 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.3f frac %.3f grid_ms %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['grid_path']['ms']))"; }
run A=1
