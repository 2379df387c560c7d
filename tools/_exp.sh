run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.3f' % d['roofline']['kernel_ms'], d.get('grid_path'))"; }
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
run A=1
run BENCH_ARGS="--workload cfg5"
run BENCH_ARGS="--workload cfg4 --grid 16384"
run BENCH_ARGS="--workload cfg4 --grid 16384 --no-rows"
run BENCH_ARGS="--workload cfg1"
run BENCH_ARGS="--workload cfg3"
