#!/bin/bash
# Round 4, eighth call: chain tests once more, cell sizes without staging, fused selection, the default bench line,
# VALU class counters (calibration + the BASELINE configs).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== tests"
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or nested or body_minus or staged_operator or thousands or candidate or two_row or survivor or importing" -o faulthandler_timeout=400 > $O/r04_cells4_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r04_cells4_pytest.txt | cut -c1-300
timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_consumers.py -m gpu -x -q -o faulthandler_timeout=200 2>&1 | tail -2
echo "== cell sizes, staging off"
for n in 1000 4096; do
  for f in "3,1,0" "4,1,0" "4,2,0" "4,2,1" "5,1,0" "5,2,0"; do
    SDFK_CELL_FINE=$f timeout -k 10 200 python3 tools/big_union_bench.py --spheres $n --grid 512 --no-interp 2>&1 | grep "^culled" | cut -c1-42 | sed "s/^/$n fine $f: /"
  done
done
echo "== fused select"; timeout -k 10 300 python3 tools/fused_select_bench.py 1024 cfg2 2>&1 | grep -v amdgpu.ids | tail -6
echo "== bench"; timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/r04_bench_a.json 2> $O/r04_bench_a.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
d = json.loads([l for l in open(O + "/r04_bench_a.json") if l.startswith("{")][-1])
print("headline %.3f ms frac %.3f  grid512 %.3f" % (d["ms_per_step"], d["roofline"]["frac"], d["grid_512"]["roofline_frac"]))
for k, v in d.get("other_configs", {}).items(): print(k, round(v.get("kernel_ms", 0), 3), round(v.get("roofline_frac", 0), 3), v.get("verified"), v.get("same_field_as_the_three_row_call"), v.get("error"))
print("host_path", d.get("host_path")); print("next_rows", json.dumps(d.get("next_rows"))[:900]); print("extras_s", d.get("extras_s"))
PY
echo "== valu class counters"
CLS="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc $CLS --output-format csv -d $O/r04_valu_calib_cls -- $R/tools/bin/valu_calib 1024 > $O/r04_valu_calib_cls.log 2>&1; echo "calib pmc rc=$?"
  for w in cfg2 cfg3 cfg5; do
    timeout -k 10 300 rocprofv3 --pmc $CLS --output-format csv -d $O/r04_valu_cls_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-extras --cpu-seconds 0 > /dev/null 2> $O/r04_valu_cls_$w.log; echo "$w rc=$?"
  done
  timeout -k 10 300 rocprofv3 --pmc $CLS --output-format csv -d $O/r04_valu_cls_cfg4 -- python3 $R/bench.py --workload cfg4 --grid 16384 --steps 50 --warmup 20 --no-extras --cpu-seconds 0 > /dev/null 2> $O/r04_valu_cls_cfg4.log; echo "cfg4 rc=$?" )
timeout -k 10 100 tools/bin/valu_calib 4096 > $O/r04_valu_calib3.jsonl 2>&1
python3 tools/valu_roof.py calib $O/r04_valu_calib3.jsonl $O/r04_valu_calib_cls $O/r04_valu_calib_classes.json | head -60
for w in cfg2:sdfk_spec_r cfg3:sdfk_spec_v4 cfg5:sdfk_spec_r cfg4:sdfk_spec_r; do
  python3 tools/valu_roof.py summarize $O/r04_valu_cls_${w%%:*} ${w##*:} $O/r04_valu_roof_${w%%:*}.json | python3 -c "
import sys, json; r = json.load(sys.stdin); print('${w%%:*}', 'insts %.0fM' % (r['valu_wave_instructions_per_launch']/1e6), 'busy', {k: round(v, 3) for k, v in r['valu_busy_frac'].items()}, 'mean cycles %.2f' % r['mean_issue_cycles_per_instruction'], {k[14:]: round(v/1e6) for k, v in r['class_counts'].items()}, 'unclassified %.0fM' % (r['unclassified']/1e6))"
done
echo "batch8 done"
