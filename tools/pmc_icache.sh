#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of one bench workload.  tools/pmc_icache.sh cfg4 16384
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
w=$1; g=${2:-1024}
out=$R/gpurun_out/pmc_icache_$w
rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$out" -- python3 $R/bench.py --workload $w --grid $g --steps 5 --warmup 2 --no-extras --cpu-seconds 0 > /dev/null 2> "$out.log"
cd $R
python3 tools/pmc_summarize.py "$out" sdfk_spec_r "$out.json" | python3 -c "
import sys,json
d=json.load(sys.stdin)['per_launch']
print('$w', {k: '%.4g' % v for k,v in d.items()})
if 'SQC_ICACHE_REQ' in d: print('  icache miss rate %.3f, misses per wave %.1f' % (d['SQC_ICACHE_MISSES']/max(d['SQC_ICACHE_REQ'],1), d['SQC_ICACHE_MISSES']/max(d.get('SQ_WAVES',1),1)))
"
