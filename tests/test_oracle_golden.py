"""CPU: the oracle (oracle/sdf_oracle.py) and the host-side API mirror, pinned against golden vectors
produced by the REAL reference (tests/golden/generate_golden.py, run in the build container)."""
import numpy as np
import pytest

import scenes
import aegolius_amd.cores as ns
from oracle import sdf_oracle

ALL = sorted(scenes.SCENES)


def rel_err(out, ref):
    both_nan = np.isnan(ref) & np.isnan(out)
    err = np.abs(out - ref) / np.maximum(1.0, np.abs(ref))
    err[both_nan] = 0.0
    return err


def test_golden_covers_every_scene(golden):
    data, meta = golden
    assert set(meta["scenes"]) == set(scenes.SCENES)
    assert set(meta["grid_scenes"]) == set(scenes.GRID_SCENES)
    assert data["inputs"].dtype == np.float32 and data["inputs"].shape == (3, meta["n_points"])
    # inputs are reproducible from the seed (no hidden state in the fixture)
    np.testing.assert_array_equal(data["inputs"], scenes.input_points().astype(np.float32))


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference(name, golden, golden_inputs):
    data, _ = golden
    ref = data["scene/" + name]
    co = golden_inputs.copy()
    with np.errstate(all="ignore"):
        out = sdf_oracle.evaluate(scenes.SCENES[name](ns), co)
    np.testing.assert_array_equal(co, golden_inputs)          # create() never mutates the caller's array
    assert out.shape == ref.shape and out.dtype == np.float64
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    assert rel_err(out, ref).max() <= 1e-12


@pytest.mark.parametrize("name", sorted(scenes.GRID_SCENES))
def test_oracle_matches_reference_on_grid_scenes(name, golden):
    """signed / conv_averaging / conv_edge_detection: evaluated on whole generate_grid clouds (own inputs)."""
    data, meta = golden
    build, key = scenes.GRID_SCENES[name]
    co, res = scenes.grid_inputs(ns, key)
    np.testing.assert_array_equal(co.astype(np.float32), data["gridinputs/" + key])
    ref = data["gridscene/" + name]
    with np.errstate(all="ignore"):
        out = sdf_oracle.evaluate(build(ns, res), co.copy())
    assert list(out.shape) == meta["grid_scenes"][name]["shape"] and out.dtype == np.float64
    assert rel_err(out, ref).max() <= 1e-12
    if "signed" in name and "already" not in name:
        assert (ref < 0).sum() > 30                  # the sign really is recovered


@pytest.mark.parametrize("name", sorted(scenes.CONSUMER_SCENES))
def test_oracle_consumers_match_reference(name, golden):
    """point_cloud (interior extraction) and from_sdf (gradient direction) of the reference, on its own fields."""
    data, meta = golden
    build, key = scenes.CONSUMER_SCENES[name]
    co, res = scenes.grid_inputs(ns, key)
    field = data["consumer/%s/field" % name]
    with np.errstate(all="ignore"):
        assert rel_err(sdf_oracle.evaluate(build(ns, res), co.copy()), field).max() <= 1e-12
    cloud = sdf_oracle.point_cloud(field, co)
    np.testing.assert_array_equal(cloud, data["consumer/%s/cloud" % name])
    assert cloud.shape == (3, meta["consumer_scenes"][name]["interior"])
    field32 = data["consumer/%s/field32" % name]
    np.testing.assert_array_equal(field32, field.astype(np.float32))
    vec = sdf_oracle.from_sdf(field32.astype(np.float64), res)
    np.testing.assert_array_equal(vec, data["consumer/%s/direction" % name])
    lengths = np.linalg.norm(vec, axis=0)
    assert np.all((np.abs(lengths - 1) < 1e-12) | (lengths == 0))


def test_consumer_scenes_cover_the_edge_cases(golden):
    _, meta = golden
    assert set(meta["consumer_scenes"]) == set(scenes.CONSUMER_SCENES)
    assert meta["consumer_scenes"]["consume_empty_interior"]["interior"] == 0
    data = golden[0]
    flat = data["consumer/consume_flat_plateaus/direction"]
    assert (np.linalg.norm(flat, axis=0) == 0).mean() > 0.3        # zero gradients stay zero vectors


@pytest.mark.parametrize("name", ALL)
def test_transform_state_matches_reference(name, golden):
    """EuclideanTransform bookkeeping (scipy Rotation composition, centre, scale) equals the reference's."""
    _, meta = golden
    m = meta["scenes"][name]
    obj = scenes.SCENES[name](ns)
    np.testing.assert_allclose(obj.rotation_matrix, np.asarray(m["rotation_matrix"]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.asarray(obj.center, dtype=float), np.asarray(m["center"]), rtol=0, atol=0)
    assert float(obj.scale) == m["scale"]


def test_aliasing_scenes_are_not_vacuous(golden, golden_inputs):
    """The in-place-mutation fixtures really differ from a copy-semantics evaluation."""
    data, _ = golden
    s = ns.Sphere(0.5)
    s.move_sdf((0.2, 0.0, 0.0))
    s.symmetry(0)
    plain = scenes.placed(s)                      # same chain without the displacement
    with np.errstate(all="ignore"):
        base = sdf_oracle.evaluate(plain, golden_inputs)
    ref = data["scene/alias_symmetry_displacement"]
    # displacement added |x'| (mutated coordinates), so ref - base >= 0 everywhere and > 0 somewhere
    assert np.all(ref - base >= -1e-12) and np.any(ref - base > 0.1)


@pytest.mark.parametrize("gname", ["g3_even", "g3_mixed", "g3_scalar_res", "g2", "g2_scalar_res", "g1"])
def test_generate_grid_matches_reference(gname, golden):
    data, meta = golden
    g = meta["grids"][gname]
    res = g["resolution"] if len(g["resolution"]) > 1 else g["resolution"][0]
    co, r = ns.generate_grid(tuple(g["size"]), res)
    assert list(r) == g["returned_resolution"]
    assert co.dtype == np.float64
    np.testing.assert_array_equal(co, data["grid/" + gname])


def test_resolution_conversion_and_reshape(golden):
    _, meta = golden
    assert [ns.resolution_conversion(k) for k in (128, 129, 512, 1024, 2048, 16384)] == [129, 129, 513, 1025, 2049,
                                                                                         16385]
    assert list(ns.smarter_reshape(np.zeros(9 ** 3), 8).shape) == meta["smarter_reshape"]["129_cubed"]
    assert list(ns.smarter_reshape(np.zeros(9 * 5), (8, 5)).shape) == meta["smarter_reshape"]["2d"]
    with pytest.raises(ValueError):
        ns.smarter_reshape(np.zeros(10), (8, 8, 8))
