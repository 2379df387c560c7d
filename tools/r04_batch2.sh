#!/bin/bash
# Round 4, second call: pipelined plain kernel (SDFK_VTILES) on cfg3 / cfg1, per-class VALU issue rates, cfg3 counters,
# host path with the prefaulted result array.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== cfg3 pipelined"
timeout -k 10 400 python3 tools/rows_ab.py --workload cfg3 --grid 1024 base VTILES=2 VTILES=4 VTILES=8 VTILES=16 VTILES=64 VTILES=512 VTILES=8+V4_WPE=8 VTILES=16+V4_WPE=8 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg3_vtiles.txt
echo "== cfg1 pipelined"
timeout -k 10 300 python3 tools/rows_ab.py --workload cfg1 --grid 1024 base VTILES=2 VTILES=4 VTILES=8 VTILES=16 VTILES=64 VTILES=512 base 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg1_vtiles.txt
echo "== cfg2 nocull pipelined (v4 kernel, VALU-heavy)"
timeout -k 10 300 python3 tools/rows_ab.py --workload cfg2 --grid 1024 mode=nocull:base mode=nocull:VTILES=8 mode=nocull:VTILES=16+V4_WPE=8 2>&1 | grep -v amdgpu.ids | tee $O/r04_cfg2_nocull_vtiles.txt
echo "== valu_calib"; timeout -k 10 200 tools/bin/valu_calib 4096 > $O/r04_valu_calib2.jsonl 2>&1; echo "rc=$?"
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for l in open(O + "/r04_valu_calib2.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        if d["waves_per_simd"] == 8: print("%-26s %.2f cycles per instr per SIMD at 2.4 GHz" % (d["kind"], d["cycles_per_instr_at_2400MHz"]))
PY
echo "== cfg3 counters"
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r04_cfg3_pmc/sq -- python3 $R/bench.py --workload cfg3 --steps 10 --warmup 3 --no-extras --cpu-seconds 0 > $O/r04_cfg3_pmc.log 2>&1; echo "pmc rc=$?" )
python3 tools/pmc_summarize.py $O/r04_cfg3_pmc sdfk_spec_v4 $O/r04_cfg3_pmc_summary.json | head -40
echo "== host path"; timeout -k 10 300 python3 tools/host_path_ab.py 96 > $O/r04_host_path_ab2.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/r04_host_path_ab2.txt
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
echo "batch2 done"
