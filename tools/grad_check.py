"""Developer tool (GPU): the gradient kernels against each other on a few shapes — flat (SDFK_GRADIENT_FLAT=1) and register
carry (default): first mismatches, raw and normalised."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from aegolius_amd import DeviceField
    out = {}
    for shape in [(13, 15, 11), (3, 70, 129), (35, 5, 201), (65, 37, 130), (70, 9, 8), (2, 2, 9), (40, 33, 36), (67, 3, 1023),
                  (34, 129, 130), (100, 7, 11), (5, 200, 200), (9, 40, 161), (33, 20, 513), (6, 12, 1025), (4, 30, 322),
                  (40, 3, 160), (3, 9, 2051)]:
        rng = np.random.default_rng(sum(shape))
        f = rng.normal(size=int(np.prod(shape))).astype(np.float32)
        f[rng.random(f.size) < 0.2] = 0.25
        dev = DeviceField.from_host(f)
        out[str(shape)] = dev.gradient(shape, normalize=False)
        out[str(shape) + " unit"] = dev.gradient(shape, normalize=True)
    np.savez(sys.argv[1], **out)
else:
    env = dict(os.environ)
    env["SDFK_GRADIENT_FLAT"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/g_flat.npz"], env=env)
    env.pop("SDFK_GRADIENT_FLAT")
    subprocess.check_call([sys.executable, __file__, "/tmp/g_carry.npz"], env=env)
    b = np.load("/tmp/g_flat.npz")
    for tag in ("carry",):
        a = np.load("/tmp/g_%s.npz" % tag)
        for k in a.files:
            x, y = a[k], b[k]
            bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
            print(tag, k, "mismatches", len(bad), "of", x.size)
            shape = eval(k.replace(" unit", ""))
            for comp, idx in bad[:6]:
                print("   comp", comp, "idx", idx, np.unravel_index(idx, shape), tag, x[comp, idx], "flat", y[comp, idx])
