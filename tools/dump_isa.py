"""Developer tool: write the generated HIP source of a scene's specialised kernel and its gfx950 ISA,
and print VALU instruction statistics of sdfk_spec_v4 (per thread = 4 points)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(scene="tree_cfg2_smooth_union10", kernel="sdfk_spec_v4", outdir="/tmp/isa"):
    import __graft_entry__
    __graft_entry__.build()
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    os.makedirs(outdir, exist_ok=True)
    low = lower_geometry(scenes.SCENES[scene](ns))
    prog = _engine.Program.from_lowered(low)
    src = os.path.join(outdir, scene + ".hip")
    with open(src, "w") as f:
        f.write("#include <hip/hip_runtime.h>\n" + prog.source())
    asm = os.path.join(outdir, scene + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-mno-amdgpu-ieee", "-std=c++17", "-S",
                    "--cuda-device-only", "-Wno-unused-command-line-argument", src, "-o", asm], check=True)
    text = open(asm).read()
    body = text[text.index(kernel + ":"):]
    open(os.path.join(outdir, kernel + ".s"), "w").write(body[:body.index("s_endpgm")])
    body = body[:body.index("s_endpgm")]
    ops = collections.Counter(re.findall(r"^\s+([vs]_[a-z0-9_]+)", body, re.M))
    valu = sum(n for k, n in ops.items() if k.startswith("v_"))
    trans = sum(n for k, n in ops.items() if re.match(r"v_(sqrt|rcp|rsq|exp|log|sin|cos)_", k))
    meta = re.search(r"sdfk_spec_v4.*?NumVgprs: (\d+).*?NumSgprs: (\d+)|; NumSgprs: (\d+)\n; NumVgprs: (\d+)", text, re.S)
    m2 = re.findall(r"; (NumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", text[text.index(kernel + ":"):])[:5]
    print("scene %s kernel %s: %d instructions, %d bytecode ops" % (scene, kernel, sum(ops.values()), low.code.shape[0]))
    print("VALU per thread %d (= %.1f per point), of which transcendental %d, spill lanes %d" % (
        valu, valu / 4, trans, ops.get("v_readlane_b32", 0) + ops.get("v_writelane_b32", 0)))
    print("resources:", m2)
    for k, n in ops.most_common(24):
        print("  %5d %s" % (n, k))


if __name__ == "__main__":
    main(*sys.argv[1:])
