#!/bin/bash
# Developer tool (GPU box), round 4, first call: VALU calibration (plain + counters), host-path A/B, per-config baselines
# of the shipped kernels on THIS box, the RCCL world-1 child by hand, the watchdog rehearsal.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== valu_calib"; timeout -k 10 120 tools/bin/valu_calib 4096 > $O/r04_valu_calib.jsonl 2>&1; echo "rc=$?"; tail -3 $O/r04_valu_calib.jsonl
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r04_valu_calib_pmc -- $R/tools/bin/valu_calib 2048 > $O/r04_valu_calib_pmc.log 2>&1; echo "pmc rc=$?" )
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(O + "/r04_valu_calib_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, c in sorted(acc.items()):
    # per dispatch: counters are summed over dimensions already per row? sum rows per dispatch / number of dispatches (8)
    n = 8.0
    v = {name: sum(vals) / n for name, vals in c.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    v["cycles_per_launch"] = cyc
    if cyc:
        v["insts_per_simd_cycle"] = v.get("SQ_INSTS_VALU", 0) / 1024.0 / cyc
        v["active_x4_per_simd_cycle"] = v.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / 1024.0 / cyc
    out[k] = v
json.dump(out, open(O + "/r04_valu_calib_pmc.json", "w"), indent=1)
for k, v in out.items():
    print(k[:40], {a: round(b, 3) for a, b in v.items() if a in ("insts_per_simd_cycle", "active_x4_per_simd_cycle", "cycles_per_launch")})
PY
echo "== host path"; timeout -k 10 300 python3 tools/host_path_ab.py 96 > $O/r04_host_path_ab.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/r04_host_path_ab.txt
echo "== baselines"
for w in "cfg2 1024" "cfg2 512" "cfg3 1024" "cfg5 1024" "cfg4 16384"; do set -- $w
  timeout -k 10 300 python3 tools/rows_ab.py --workload $1 --grid $2 base 2>&1 | grep -v amdgpu.ids | sed "s/^/$1 $2: /" | tee -a $O/r04_baselines.txt; done
echo "== rccl world 1"; timeout -k 10 240 python3 tests/rccl_world1_child.py 128 29633 > $O/r04_rccl_world1.json 2> $O/r04_rccl_world1.err; echo "rc=$?"; tail -c 1500 $O/r04_rccl_world1.json; tail -5 $O/r04_rccl_world1.err
echo "== watchdog rehearsal"
SDFK_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 3 --warmup 1 --extras-timeout 8 --cpu-seconds 0 > $O/r04_watchdog_n2.json 2> $O/r04_watchdog_n2.err; echo "watchdog rc=$?" | tee -a $O/r04_watchdog_n2.err
echo "batch1 done"
