"""Developer tool: reduce the counter_collection CSVs written by tools/pmc_collect.sh to one JSON
(per-launch averages for the named kernel), applying the gfx950 FETCH_SIZE correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of wide coalesced reads)."""
import csv
import glob
import json
import os
import sys


def collect(root, kernel):
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name_ = row.get("Kernel_Name", "").split("(")[0].strip()
                if kernel.startswith("~"):                                          # "~text": any kernel whose name contains text
                    if kernel[1:] not in name_:                                     # (template kernels: "void f<1>")
                        continue
                elif name_ != kernel:                                               # exact name, not a prefix
                    continue
                name, val = row["Counter_Name"], float(row["Counter_Value"])
                did = row.get("Dispatch_Id")
                acc.setdefault(name, {}).setdefault((path, did), 0.0)
                acc[name][(path, did)] += val
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main(root, kernel="sdfk_spec_v4", out=None):
    avg, n = collect(root, kernel)
    rec = {"kernel": kernel, "launches_seen": n, "per_launch": avg}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        # units: KiB per the rocprofv3 counter definition; gfx950 correction x2 on FETCH_SIZE
        rec["hbm_read_bytes_per_launch"] = avg["FETCH_SIZE"] * 1024 * 2
        rec["hbm_write_bytes_per_launch"] = avg["WRITE_SIZE"] * 1024
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        rec["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
    if "SQ_INSTS_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
        # (rounds 1-3 derived "valu_active_frac" = SQ_ACTIVE_INST_VALU * 4 / SIMD-cycles from here. That counter ticks once per
        #  instruction — it reads 1.02 x SQ_INSTS_VALU on every kernel, and 0.96 on a pure v_fma loop that saturates the SIMDs
        #  (profiles/r04_valu_calib_pmc.json) — while instructions cost 2.4 to 8.2 cycles by class, so the quotient is a
        #  fraction only for an all-FMA kernel: 1.40 on cfg3. The calibrated roof is tools/valu_roof.py.)
        rec["valu_instructions_per_simd_cycle"] = avg["SQ_INSTS_VALU"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)
    print(json.dumps(rec, indent=1))
    if out:
        with open(out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
