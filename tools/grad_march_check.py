"""march vs flat gradient kernel on a few shapes: first mismatch"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from aegolius_amd import DeviceField
    out = {}
    for shape in [(13, 15, 11), (3, 70, 129), (35, 5, 201), (65, 37, 130)]:
        rng = np.random.default_rng(sum(shape))
        f = rng.normal(size=int(np.prod(shape))).astype(np.float32)
        dev = DeviceField.from_host(f)
        out[str(shape)] = dev.gradient(shape, normalize=False)
    np.savez(sys.argv[1], **out)
else:
    env = dict(os.environ)
    env.pop("SDFK_GRADIENT_MARCH", None)
    subprocess.check_call([sys.executable, __file__, "/tmp/g_flat.npz"], env=env)
    env["SDFK_GRADIENT_MARCH"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/g_march.npz"], env=env)
    a, b = np.load("/tmp/g_march.npz"), np.load("/tmp/g_flat.npz")
    for k in a.files:
        x, y = a[k], b[k]
        bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
        print(k, "mismatches", len(bad), "of", x.size)
        shape = eval(k)
        for comp, idx in bad[:12]:
            print("   comp", comp, "idx", idx, np.unravel_index(idx, shape), "march", x[comp, idx], "flat", y[comp, idx])
