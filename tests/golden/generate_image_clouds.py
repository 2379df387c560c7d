"""Fixture generator (build container only): the point clouds the reference's image examples extract from its own test
images (Files/test_images/*.png through `Points.from_image`, C/geom.py), kept as DATA in tests/golden/image_clouds.npz —
x and y of every point, float64, bit for bit (pixel lattices: they compress to tens of kilobytes). The example builders of
tests/example_scenes.py / example_pipelines.py read them; `generate_example_golden.py` then proves each builder equal
to its script, which pins these arrays to what the script itself extracts.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_image_clouds.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/Code/spomso")
sys.dont_write_bytecode = True
IMAGES = "/root/reference/Files/test_images"

from spomso.cores.geom import Points  # noqa: E402  (the real reference)


def alpha_greyscale(image):
    a = np.asarray(image) / 255.0
    return np.maximum(1 - a[:, :, 1], a[:, :, 0])


def cloud(image, size, threshold):
    p = Points([])
    p.from_image(image, size, binary_threshold=threshold)
    c = np.asarray(p.cloud, dtype=np.float64)
    assert c.shape[0] == 3 and not c[2].any()
    return c[:2].copy()


def main():
    out = {}
    lines = Image.open(os.path.join(IMAGES, "lines_test_handdrawn.png")).convert("L")
    out["lines"] = cloud(lines, (3, 1.5), 0.5)                               # 2D/pointcloud_image_2D.py
    owl = alpha_greyscale(Image.open(os.path.join(IMAGES, "owl_logo.png")).convert("LA"))
    out["owl_exterior"] = cloud(owl, (9, 16), 0.0)                           # 2D/sdf_from_mask_2D.py
    out["owl_interior"] = cloud(1 - owl, (9, 16), 0.0)
    shapes = alpha_greyscale(Image.open(os.path.join(IMAGES, "dilation_erosion.png")).convert("LA"))
    out["shapes"] = cloud(shapes, (3, 1.5), 0.2)                             # 2D/erosion_dilation_image_2D.py
    np.savez_compressed(os.path.join(HERE, "image_clouds.npz"), **out)
    for k, v in out.items():
        print("%-14s %8d points" % (k, v.shape[1]))
    print("%d bytes" % os.path.getsize(os.path.join(HERE, "image_clouds.npz")))


if __name__ == "__main__":
    main()
