"""The reference's own example scripts that are pipelines — the five vector examples, the post-processing functions, the
image -> cloud -> field -> interior points -> cloud -> field example (tests/example_pipelines.py) — as stage-wise
fixtures of the real reference (tests/golden/generate_example_pipeline_golden.py ran every script itself and proved each
workflow equal to it).

A workflow is replayed stage by stage: every intermediate array is compared with the reference's record and the walk
continues FROM THE RECORD, so each stage — SDF tree, 5x5 smoothing, falloff maps, gradient direction, vector chain, six
read-outs — is checked on exactly the reference's input.
  CPU  (this file, not gpu): the float64 oracle on every stage, <= 1e-12 * max(1, |ref|);
  GPU  (-m gpu): aegolius_amd through the product API, |gpu - ref| <= 1e-6 * max(1, |ref|); where a stage is
       ill-conditioned at a point (arccos at the poles, the direction of a vanishing gradient) the bound is 2x what the
       float64 oracle itself moves when the stage's inputs move by one fp32 ulp; an SDF stage is judged against the
       largest intermediate of the tree at the point (coordinates of +-50 here: `sdf_oracle.evaluate_with_magnitude`, the
       rule of tests/test_gpu_parity.py); phi is compared as an angle (+pi and -pi are the same direction).
"""
import json
import os

import numpy as np
import pytest

import example_pipelines as evs
import aegolius_amd.cores as ns
from oracle import sdf_oracle
from oracle import vector_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(n for n, e in evs.EXAMPLES.items() if e["raises"] is None)
RAISING = sorted(n for n, e in evs.EXAMPLES.items() if e["raises"] is not None)
TOL = 1e-6


@pytest.fixture(scope="module")
def golden():
    data = np.load(os.path.join(HERE, "golden", "example_pipelines.npz"))
    with open(os.path.join(HERE, "golden", "example_pipelines_meta.json")) as f:
        return data, json.load(f)


class OracleEvaluator:
    """The same walk with every evaluation handed to oracle/: the trees and chains are the ones aegolius_amd's API
    mirror records (no GPU, no libsdfk)."""

    def __init__(self):
        self._last = None                                       # (expression, coordinates, field): a selection re-uses the field

    def sdf(self, tree, co):
        co = np.asarray(co, dtype=np.float64)
        # expression nodes are immutable: the same node, the same parameter objects, the same transform = the same field
        state = (tree.modified_object, tree._geo_parameters, float(tree.scale),
                 np.asarray(tree.center, dtype=np.float64).tobytes(), np.asarray(tree.rotation_matrix, dtype=np.float64).tobytes())
        last = self._last
        same = last is not None and last[0][0] is state[0] and len(last[0][1]) == len(state[1]) \
            and all(a is b for a, b in zip(last[0][1], state[1])) and last[0][2:] == state[2:] \
            and last[1].shape == co.shape and np.array_equal(last[1], co)
        if not same:
            self._last = (state, co.copy(), sdf_oracle.evaluate(tree, co))
        return self._last[2].copy()

    def vector(self, field, arg, read_out):
        return vo.evaluate(field.vf, np.asarray(arg, dtype=np.float64), field._vf_parameters,
                           "vector" if read_out == "create" else read_out)

    def conv_averaging(self, grid, kernel_size, iterations):
        return sdf_oracle._conv_averaging(np.asarray(grid, dtype=np.float64), kernel_size, iterations)

    def linear_falloff(self, u, amplitude, width):                          # C/post_processing.py linear_falloff
        return np.clip(1 - np.asarray(u, dtype=np.float64) / width, 0, 1) * amplitude

    def smarter_reshape(self, pattern, resolution):
        return sdf_oracle._smarter_reshape(np.asarray(pattern, dtype=np.float64), resolution)

    def batch_normalize(self, vec):
        return vo.unit(np.asarray(vec, dtype=np.float64))

    def post(self, name, u, **kwargs):
        return sdf_oracle.POST_FUNCTIONS[name](np.asarray(u, dtype=np.float64), kwargs)

    def point_cloud(self, tree, co):
        return sdf_oracle.point_cloud(self.sdf(tree, co), co)


def selected(name, data, cloud):
    """Grid indices of the columns of an interior-point cloud (the columns ARE grid points: C/geom.py:62-74)."""
    co = evs.continue_from("coor", data[name + "/coor"])
    index = {(x, y): i for i, (x, y) in enumerate(zip(co[0], co[1]))}
    assert len(index) == co.shape[1]
    cloud = np.asarray(cloud, dtype=np.float64)
    assert cloud.ndim == 2 and cloud.shape[0] == 3 and not cloud[2].any()
    picked = np.array([index[(x, y)] for x, y in zip(cloud[0], cloud[1])], dtype=np.int64)
    assert np.all(np.diff(picked) > 0), "interior points come in grid order, each once"
    return picked


def check_selection(name, data, got, ref, tolerance):
    """Two selections of `field <= 0` may differ only where the reference's field is within `tolerance` of zero."""
    a, b = selected(name, data, got), selected(name, data, ref)
    field = data[name + "/sdf"]
    np.testing.assert_array_equal(b, np.flatnonzero(field <= 0))         # the record is the reference's own selection
    differ = np.setxor1d(a, b)
    assert np.all(np.abs(field[differ]) <= tolerance * np.maximum(1.0, np.abs(field[differ]))), \
        "%d points selected differently, |field| up to %.3g" % (differ.size, np.abs(field[differ]).max())
    return differ.size


def walk(name, data, evaluator, check, perturb=None):
    """Replay one workflow; `check(stage, got, ref)` sees every stage, the walk continues from the record (moved by one
    fp32 ulp per element when `perturb` is a generator: the conditioning runs)."""
    e = evs.EXAMPLES[name]

    def hook(stage, value):
        ref = data[name + "/" + stage]
        check(stage, np.asarray(value), ref)
        go_on = evs.continue_from(stage, ref)
        if perturb is not None and not stage.startswith("out/"):
            return go_on * (1.0 + 6e-8 * perturb.choice([-1.0, 1.0], size=ref.shape))
        return go_on
    with np.errstate(all="ignore"):
        return e["run"](ns, evaluator, hook, e["size"], e["res"], **e["variant"])


def test_fixture_covers_every_example_and_every_workflow_is_its_script(golden):
    data, meta = golden
    assert set(meta["scenes"]) == set(evs.EXAMPLES)
    scripts = {e["script"] for e in evs.EXAMPLES.values()}
    assert scripts == {"vector/buildin_vector_fields.py", "vector/custom_vector_field.py", "vector/from_components.py",
                       "vector/revolve_vector_field.py", "vector/sdf_vector_field.py",
                       "scalar/2D/post_processing_scalar_2D.py", "scalar/2D/erosion_dilation_image_2D.py",
                       "scalar/2D/surface_reconstruction_2D.py"}
    for name in NAMES:
        m = meta["scenes"][name]
        assert m["workflow_equals_script_bit_for_bit"] is True and m["nan"] == 0
        assert {k.split("/", 1)[1] for k in data.files if k.startswith(name + "/")} == set(m["stages"])
    for name in RAISING:
        assert meta["scenes"][name]["script_and_workflow_raise_it"] is True


@pytest.mark.parametrize("name", NAMES)
def test_oracle_replays_every_stage(name, golden):
    data, _ = golden
    seen = []

    def check(stage, got, ref):
        seen.append(stage)
        if stage == "new_cloud":
            assert check_selection(name, data, got, ref, 1e-12) == 0
            return
        assert got.shape == ref.shape, stage
        bound = 1e-12 * np.maximum(1.0, np.abs(ref))
        err = np.abs(got - ref)
        assert not (err > bound).any(), "%s: %d off, worst %.3g" % (stage, (err > bound).sum(), err.max())
    out = walk(name, data, OracleEvaluator(), check)
    assert set(out) == set(evs.EXAMPLES[name]["outputs"]) and len(seen) == len(golden[1]["scenes"][name]["stages"])


@pytest.mark.parametrize("name", RAISING)
def test_variants_the_reference_cannot_evaluate_raise_the_same_error(name):
    e = evs.EXAMPLES[name]
    with pytest.raises(e["raises"]):                            # raised while lowering: before any GPU call
        e["run"](ns, evs.ProductEvaluator(ns), lambda _s, v: v, e["size"], e["res"], **e["variant"])


# ---- GPU ----------------------------------------------------------------------------------------------------------------
def conditioning(name, data, trials=3):
    """Per stage: how far the float64 oracle's output moves when the stage's recorded inputs move by one fp32 ulp."""
    base = {}
    walk(name, data, OracleEvaluator(), lambda stage, got, ref: base.__setitem__(stage, np.array(got, dtype=np.float64)))
    worst = {k: np.zeros_like(v) for k, v in base.items()}
    rng = np.random.default_rng(7)
    for _ in range(trials):
        def note(stage, got, ref):
            if np.shape(got) == base[stage].shape:              # (a selection may change its size under the perturbation)
                worst[stage] = np.fmax(worst[stage], np.abs(np.asarray(got, dtype=np.float64) - base[stage]))
        walk(name, data, OracleEvaluator(), note, perturb=rng)
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_replays_every_stage(name, golden, built):
    built.require_gpu()
    data, _ = golden
    slack = conditioning(name, data, trials=1 if name.startswith("erosion") else 3)   # (200 K-point clouds in the oracle)

    def check(stage, got, ref):
        if stage == "new_cloud":                                # interior points: the same grid points, up to fp32 at field = 0
            check_selection(name, data, got, ref, TOL)
            return
        assert got.shape == ref.shape, stage
        if stage == "coor":
            assert np.array_equal(np.asarray(got).astype(np.float32), ref.astype(np.float32))
            return
        assert np.array_equal(np.isnan(got), np.isnan(ref)), stage
        scale = np.maximum(1.0, np.abs(ref))
        if stage == "sdf":
            scale = np.maximum(scale, magnitudes.pop())
        bound = np.maximum(TOL * scale, 2.0 * slack[stage])
        if stage == "out/theta":                                # arccos(v_z): an fp32 v_z costs 1 / sin(theta) ...
            vz = data[name + "/out/z"]
            bound = np.maximum(bound, 3e-7 / np.sqrt(np.maximum(1.0 - vz * vz, 1e-12)))
            if np.all(np.abs(vz) == 1.0):                       # ... a constant field along z: every point a pole, all exact
                bound = np.zeros_like(bound)
        err = np.abs(np.asarray(got, dtype=np.float64) - ref)
        if stage == "out/phi":                                  # atan2(v_y, v_x): an fp32 (v_x, v_y) costs 1 / its length
            err = np.minimum(err, 2 * np.pi - err)
            planar = np.hypot(data[name + "/out/x"], data[name + "/out/y"])
            bound = np.where(planar > 0, np.maximum(bound, 3e-7 / np.maximum(planar, 1e-12)), bound)
        bad = err > bound
        assert not bad.any(), "%s: %d of %d off, worst %.3g" % (stage, bad.sum(), bad.size, err[bad].max())
        assert np.median(bound / scale) <= 4 * TOL, "%s: a stage that is ill-conditioned everywhere pins nothing" % stage

    class Evaluator(evs.ProductEvaluator):
        def sdf(self, tree, co):                                # the error scale of the tree at every point, for `check`
            with np.errstate(all="ignore"):
                magnitudes.append(sdf_oracle.evaluate_with_magnitude(tree, np.asarray(co, dtype=np.float64))[1])
            return super().sdf(tree, co)
    magnitudes = []
    out = walk(name, data, Evaluator(ns), check)
    assert set(out) == set(evs.EXAMPLES[name]["outputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["waveguide_circle", "revolve_z", "components_cylindrical"])
def test_gpu_end_to_end_without_the_records(name, golden, built):
    """The whole pipeline on the GPU path with NO stage replaced by its record: unit vectors, and directions within 1e-3 of
    the reference's wherever the reference's own pipeline is well-conditioned (a bound for plumbing errors, not parity:
    the stage-wise test above is the parity statement)."""
    built.require_gpu()
    data, _ = golden
    e = evs.EXAMPLES[name]
    out = e["run"](ns, evs.ProductEvaluator(ns), lambda _s, v: v, e["size"], e["res"], **e["variant"])
    ref = data[name + "/out/create"]
    got = np.asarray(out["create"], dtype=np.float64)
    assert got.shape == ref.shape
    lengths = np.linalg.norm(got, axis=0)
    ref_len = np.linalg.norm(ref, axis=0)
    assert np.all(np.abs(lengths - ref_len) < 1e-5)
    cond = conditioning(name, data)["out/create"].max(axis=0)
    well = cond < 1e-6
    assert well.mean() > 0.5
    assert np.abs(got - ref)[:, well].max() < 1e-3
