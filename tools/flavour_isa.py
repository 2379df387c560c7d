"""Developer tool (no GPU needed): build ONE kernel flavour of a BASELINE workload with hiprtc exactly as the library does
(same options, optional extra -D switches), pull the code object out of a private on-disk cache and print the kernel's
resources (VGPRs, SGPRs, LDS, scratch), its instruction histogram, and optionally write the disassembly.
    python tools/flavour_isa.py cfg4 8 [--defs "-DSDFK_X=1"] [--asm /tmp/cfg4_rows2d.s]
flavours: 0 plain array, 5 row blocks (array), 6 row blocks (grid), 8 flat row blocks (array); + 256: flag-writing build"""
import argparse
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("flavour", type=int)
    ap.add_argument("--defs", default="")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--spheres", type=int, default=0, help="workload 'union': n-ary union of this many spheres")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="sdfk-isa-")
    os.environ["SDFK_CACHE_DIR"] = tmp
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    if args.workload == "union":
        tree = workloads.sphere_union(ns, count=args.spheres or 1000)
    else:
        tree = workloads.build(args.workload, ns)[0]
    low = lower_geometry(tree)
    _engine.lib().sdfk_debug_set_rtc_defs(args.defs.encode())
    prog = _engine.Program.from_lowered(low)
    size, secs = prog.compile_flavour(args.flavour)
    files = glob.glob(os.path.join(tmp, "sdfk-*.co"))
    assert files, "no code object in the private cache"
    co = max(files, key=os.path.getmtime)
    blob = open(co, "rb").read()[:-24]
    raw = os.path.join(tmp, "kernel.co")
    open(raw, "wb").write(blob)
    print("flavour %d of %s: %d instructions of bytecode, code object %d bytes, built in %.2f s" % (
        args.flavour, args.workload, low.code.shape[0], size, secs))
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", raw], capture_output=True, text=True).stdout
    for name in re.findall(r"\.name:\s+(\S+)", notes):
        blk = notes[notes.index(".name:           " + name) - 1200:notes.index(".name:           " + name) + 800] if (".name:           " + name) in notes else ""
        get = lambda k: (re.findall(r"\.%s:\s+(\d+)" % k, blk) or ["?"])[-1]
        print("  %-18s vgpr %s  sgpr %s  lds %s B  scratch %s B  wavefront %s" % (
            name, get("vgpr_count"), get("sgpr_count"), get("group_segment_fixed_size"), get("private_segment_fixed_size"),
            get("wavefront_size")))
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", raw], capture_output=True, text=True).stdout
    ops = collections.Counter(re.findall(r"^\s+([vs]_[a-z0-9_]+|global_\w+|ds_\w+|buffer_\w+|scratch_\w+)", dis, re.M))
    print("  static instructions: %d (VALU %d, SALU %d, LDS %d, global %d)" % (
        sum(ops.values()), sum(n for k, n in ops.items() if k.startswith("v_")), sum(n for k, n in ops.items() if k.startswith("s_")),
        sum(n for k, n in ops.items() if k.startswith("ds_")), sum(n for k, n in ops.items() if k.startswith("global_"))))
    print("  " + "  ".join("%s %d" % kv for kv in ops.most_common(14)))
    if args.asm:
        open(args.asm, "w").write(dis)


if __name__ == "__main__":
    main()
