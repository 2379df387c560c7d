"""Developer tool: the calibrated VALU roof of a kernel from counter passes.

  1. calibration (tools/valu_calib.hip run plain and under the class counters): issue cycles per wave-instruction and
     SIMD for each instruction class at 8 waves per SIMD, and what each SQ_INSTS_VALU_* class counter counts;
  2. a kernel's class counts (same counters, one rocprofv3 --pmc pass) -> issue cycles it needs on the chip's 1024 SIMDs,
     against the cycles the launch took (GRBM_GUI_ACTIVE / 8):  valu_busy = sum(count_c * cycles_c) / (1024 * cycles).

    python tools/valu_roof.py summarize <pmc dir> <kernel name> <calib json> [out.json]
    python tools/valu_roof.py calib <valu_calib.jsonl of the plain run> <pmc dir of the counter run> [out.json]
    python tools/valu_roof.py reprice <stored record.json> [...]        (after a change of the cost table)
"""
import collections
import csv
import glob
import json
import os
import sys

CLASSES = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
           "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT")


def read(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"].split("(")[0].strip()][row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
    return {k: {c: sum(d.values()) / len(d) for c, d in v.items()} for k, v in acc.items()}


def calib(jsonl, pmc_dir, out=None):
    wall = {}
    for line in open(jsonl):
        if line.startswith("{"):
            d = json.loads(line)
            if d["waves_per_simd"] == 8:
                wall[d["kind"]] = d
    pm = read(pmc_dir)
    rec = {"issue_cycles_per_wave_instruction_per_simd_at_2400MHz": {k: v["cycles_per_instr_at_2400MHz"] for k, v in wall.items()},
           "class_counters_per_wave_instruction": {}}
    # kernels are void calib<KIND, 8>: map KIND -> kind name by order of the sweep in valu_calib.hip
    kinds = list(wall)
    for name, c in sorted(pm.items()):
        if not name.startswith("void calib<") or not name.rstrip(">").endswith(", 8"):
            continue
        k = int(name[len("void calib<"):].split(",")[0])
        total = c.get("SQ_INSTS_VALU", 0.0)
        rec["class_counters_per_wave_instruction"][kinds[k] if k < len(kinds) else str(k)] = {
            x: round(c.get(x, 0.0) / max(total, 1.0), 3) for x in CLASSES}
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


# issue cost of a class counter's instructions (cycles per wave-instruction per SIMD at 8 waves per SIMD, measured:
# profiles/r04_valu_calib.jsonl and the second sweep r04_valu_calib_forms.jsonl). A class counter does not say which FORM an
# instruction had, and the form decides: v_add / v_mul / v_sub with VGPR (or inline-constant) operands 2.4, the same with
# an SGPR operand 4.2; v_fmaak / v_fmamk (literal) and v_fma with an inline constant 2.3, v_fmac 3.8, v_fma with three
# registers 3.7, with an SGPR operand 4.2, every packed form 4.2; integer add / and / xor / ashr 2.4, lshl / bfi 4.2,
# bitop3 3.7; conversions, v_rndne, v_floor 4.2; v_min / v_max / v_max3 / compares / v_cndmask_e64 4.2; transcendentals
# 8.2. Hence a band per class: (cheapest form, dearest form); what the class counters do not cover (moves, selects,
# compares, min / max) is priced 2.4 .. 4.2. low / high = every instruction in its cheapest / dearest form, mid = the mean.
COST = {"SQ_INSTS_VALU_ADD_F32": (2.4, 4.2), "SQ_INSTS_VALU_MUL_F32": (2.4, 4.2), "SQ_INSTS_VALU_FMA_F32": (2.3, 4.2),
        "SQ_INSTS_VALU_TRANS_F32": (8.2, 8.2), "SQ_INSTS_VALU_INT32": (2.4, 4.2), "SQ_INSTS_VALU_CVT": (4.2, 4.2),
        "other": (2.4, 4.2)}


def price(class_counts, total, cycles):
    known = sum(class_counts.get(x, 0.0) for x in CLASSES)
    rest = max(0.0, total - known)
    lo = sum(class_counts.get(x, 0.0) * COST[x][0] for x in CLASSES) + rest * COST["other"][0]
    hi = sum(class_counts.get(x, 0.0) * COST[x][1] for x in CLASSES) + rest * COST["other"][1]
    mid = 0.5 * (lo + hi)
    return {"unclassified": rest,
            "issue_cycles_per_simd": {"low": lo / 1024.0, "mid": mid / 1024.0, "high": hi / 1024.0},
            "valu_busy_frac": {"low": lo / 1024.0 / cycles, "mid": mid / 1024.0 / cycles, "high": hi / 1024.0 / cycles},
            "mean_issue_cycles_per_instruction": mid / total,
            "cost_model": {k: list(v) for k, v in COST.items()}}


def reprice(path):
    """Re-price a stored record (per-kernel file or the bench's static record) with the current cost table."""
    rec = json.load(open(path))
    if "workloads" in rec:
        for w in rec["workloads"].values():
            r = price(w["class_counts"], w["valu_wave_instructions"], w["cycles_per_launch"])
            w.update({k: r[k] for k in ("unclassified", "valu_busy_frac", "mean_issue_cycles_per_instruction")})
        rec["cost_model"] = {k: list(v) for k, v in COST.items()}
    else:
        rec.update(price(rec["class_counts"], rec["valu_wave_instructions_per_launch"], rec["cycles_per_launch"]))
    json.dump(rec, open(path, "w"), indent=1)
    print(path, json.dumps(rec.get("valu_busy_frac") or {k: v["valu_busy_frac"] for k, v in rec["workloads"].items()}))


def summarize(pmc_dir, kernel, out=None):
    pm = read(pmc_dir)
    c = pm.get(kernel) or next((v for k, v in pm.items() if kernel in k), None)
    if c is None:
        raise SystemExit("kernel %r not in %s (%s)" % (kernel, pmc_dir, sorted(pm)[:8]))
    total = c["SQ_INSTS_VALU"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    rec = {"kernel": kernel, "valu_wave_instructions_per_launch": total, "cycles_per_launch": cycles,
           "class_counts": {x: c.get(x, 0.0) for x in CLASSES}}
    rec.update(price(rec["class_counts"], total, cycles))
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "reprice":
        for path in sys.argv[2:]:
            reprice(path)
    else:
        {"calib": calib, "summarize": summarize}[sys.argv[1]](*sys.argv[2:])
