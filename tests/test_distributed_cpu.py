"""CPU, world_size 2, gloo: the N > 1 path (slab partition + all-gather reassembly). The per-slab
evaluator is the CPU oracle here (test infrastructure); on the GPU box the same code runs with the
libsdfk evaluator (tests/test_gpu_parity.py::test_sharded_evaluation_matches_whole_grid)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_partition_the_index_space():
    from aegolius_amd.distributed import slab_bounds
    for n in (0, 1, 7, 1025 ** 2, 129 ** 3, 1025 ** 3):
        for w in (1, 2, 3, 4, 8):
            spans = [slab_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
    with pytest.raises(ValueError):
        slab_bounds(10, 2, 2)
    # whole grid rows per slab
    for shape in ((9, 7, 11), (1025, 1025, 1025), (33, 1, 129)):
        n, unit = shape[0] * shape[1] * shape[2], shape[2]
        for w in (1, 2, 3, 8):
            spans = [slab_bounds(n, w, r, unit) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            assert all(s % unit == 0 and c % unit == 0 for s, c in spans)
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
    with pytest.raises(ValueError):
        slab_bounds(10, 2, 0, unit=3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, resolution, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scenes
        import aegolius_amd.cores as ns
        from aegolius_amd.distributed import evaluate_grid_sharded, slab_bounds
        from oracle import sdf_oracle

        tree = scenes.cfg5_tree(ns)

        def oracle_slab(axes, start, count):
            n1, n2 = axes[1].size, axes[2].size
            idx = np.arange(start, start + count)
            co = np.stack([axes[0][idx // (n1 * n2)], axes[1][(idx // n2) % n1], axes[2][idx % n2]]).astype(np.float64)
            with np.errstate(all="ignore"):
                return torch.from_numpy(sdf_oracle.evaluate(tree, co).astype(np.float32))

        full, res = evaluate_grid_sharded(tree, (3, 3, 3), resolution, gather=True, evaluate_slab=oracle_slab)
        local, _ = evaluate_grid_sharded(tree, (3, 3, 3), resolution, gather=False, evaluate_slab=oracle_slab)
        n = res[0] * res[1] * res[2]
        s, c = slab_bounds(n, world, rank, res[2])
        ok = full.numel() == n and local.numel() == c and torch.equal(full[s:s + c], local)
        q.put((rank, bool(ok), full.numpy().copy(), tuple(res)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,resolution", [(2, (6, 8, 10)), (2, (8, 8, 8)), (3, (4, 6, 4))])
def test_sharded_evaluation_equals_single_rank(world, resolution):
    """Uneven slabs (odd-converted grids never divide evenly), gather on every rank, slab left distributed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, resolution, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    co, res = ns.generate_grid((3, 3, 3), resolution)
    co = co.astype(np.float32).astype(np.float64)
    with np.errstate(all="ignore"):
        want = sdf_oracle.evaluate(scenes.cfg5_tree(ns), co).astype(np.float32)
    for rank, ok, full, r in got:
        assert ok and r == res
        np.testing.assert_array_equal(full, want)


def test_sharded_evaluator_refuses_user_code_but_takes_signed():
    """Opaque callables need the whole field on one device: the slab evaluator says so instead of leaking a lowering
    exception. `signed` shards (the slabs exchange one byte per point), conv_* shard with a recomputed halo."""
    sys.path.insert(0, ROOT)
    import aegolius_amd.cores as ns
    from aegolius_amd.distributed import _GpuSlabEvaluator
    s = ns.Sphere(0.5)
    s.boundary()
    s.signed((8, 8, 8))
    ev = _GpuSlabEvaluator(s)
    assert ev.staged and ev.exchange
    t = ns.Sphere(0.5)
    t.custom_post_process(lambda u, a: a * u, (2.0,))
    with pytest.raises(NotImplementedError, match="single GPU"):
        _GpuSlabEvaluator(t)


# ---- consumers of the field, sharded ---------------------------------------------------------------
def _oracle_callbacks(tree):
    from oracle import sdf_oracle

    def slab(axes, start, count):
        n1, n2 = axes[1].size, axes[2].size
        idx = np.arange(start, start + count)
        co = np.stack([axes[0][idx // (n1 * n2)], axes[1][(idx // n2) % n1], axes[2][idx % n2]]).astype(np.float64)
        with np.errstate(all="ignore"):
            return torch.from_numpy(sdf_oracle.evaluate(tree, co).astype(np.float32))

    def gradient(ext, shape):
        vec = np.asarray(np.gradient(ext.numpy().astype(np.float64).reshape(shape))).reshape(len(shape), -1)
        m = np.linalg.norm(vec, axis=0)
        keep = m != 0
        vec[:, keep] /= m[keep]
        return torch.from_numpy(vec)

    def select(local, threshold):
        return torch.from_numpy(np.flatnonzero(local.numpy() <= threshold))

    return slab, gradient, select


@pytest.mark.parametrize("size,resolution", [((3, 3, 3), (6, 8, 10)), ((3, 3), (12, 8))])
def test_sharded_consumers_partition_exactly(size, resolution):
    """Slab + halo bookkeeping for any number of ranks (more ranks than planes included): the concatenated slabs are the
    single-rank from_sdf / flatnonzero, bit for bit (the per-slab arithmetic is the oracle's here)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd.distributed import gradient_direction_sharded, interior_indices_sharded
    from oracle import sdf_oracle
    tree = scenes.cfg5_tree(ns)
    slab, gradient, select = _oracle_callbacks(tree)
    co, _ = ns.generate_grid(size, resolution)
    with np.errstate(all="ignore"):
        field = sdf_oracle.evaluate(tree, co.astype(np.float32).astype(np.float64)).astype(np.float32)
    want = sdf_oracle.from_sdf(field.astype(np.float64), resolution)
    for world in (1, 2, 3, 5, 8, 20):
        vec = [gradient_direction_sharded(tree, size, resolution, evaluate_slab=slab, gradient_slab=gradient,
                                          world_rank=(world, r))[0].numpy() for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(vec, axis=1), want)
        idx = [interior_indices_sharded(tree, size, resolution, evaluate_slab=slab, select_slab=select,
                                        world_rank=(world, r))[0].numpy() for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(idx), np.flatnonzero(field <= 0))


def _consumer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import scenes
        import aegolius_amd.cores as ns
        from aegolius_amd.distributed import gradient_direction_sharded, interior_indices_sharded
        tree = scenes.cfg5_tree(ns)
        slab, gradient, select = _oracle_callbacks(tree)
        vec, _ = gradient_direction_sharded(tree, (3, 3, 3), (6, 8, 10), evaluate_slab=slab, gradient_slab=gradient)
        idx, _ = interior_indices_sharded(tree, (3, 3, 3), (6, 8, 10), evaluate_slab=slab, select_slab=select)
        q.put((rank, vec.numpy().copy(), idx.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_sharded_consumers_under_a_process_group():
    """world_size 2 over gloo: rank and world come from the process group; no collective is issued."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_consumer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    co, _ = ns.generate_grid((3, 3, 3), (6, 8, 10))
    with np.errstate(all="ignore"):
        field = sdf_oracle.evaluate(scenes.cfg5_tree(ns), co.astype(np.float32).astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got], axis=1),
                                  sdf_oracle.from_sdf(field.astype(np.float64), (6, 8, 10)))
    np.testing.assert_array_equal(np.concatenate([g[2] for g in got]), np.flatnonzero(field <= 0))


def test_sharded_vector_field_partitions_whole_rows():
    """vector_field_sharded hands every rank a slab of whole grid rows that together cover the cloud exactly once
    (the per-slab evaluator is a recorder here; the GPU evaluator is covered by tests/test_gpu_vector.py)."""
    sys.path.insert(0, ROOT)
    import aegolius_amd.cores as ns
    from aegolius_amd.distributed import vector_field_sharded
    field = ns.RadialSphericalVectorField()
    for size, resolution, row in (((2, 2, 2), (6, 8, 10), 11), ((3, 3), (12, 8), 9)):
        for world in (1, 2, 3, 5):
            seen = []

            def record(f, axes, start, count, out, resident):
                assert f is field and out == "length" and not resident and start % row == 0 and count % row == 0
                seen.append((start, count))
                return np.zeros(count)
            for r in range(world):
                slab, res = vector_field_sharded(field, size, resolution, out="length", evaluate_slab=record,
                                                 world_rank=(world, r))
                assert slab.shape == (seen[-1][1],)
            n = int(np.prod([a for a in ns.generate_grid(size, resolution)[0].shape[1:]]))
            assert seen[0][0] == 0 and sum(c for _, c in seen) == n
            assert all(a[0] + a[1] == b[0] for a, b in zip(seen, seen[1:]))


# ---- reassembly overlapped with the evaluation -------------------------------------------------------
def _overlap_worker(rank, world, port, shape, chunks, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aegolius_amd.distributed import evaluate_gathered_overlapped, gather_slabs, slab_bounds
        n, unit = shape[0] * shape[1] * shape[2], shape[2]
        calls = []

        def evaluate_chunk(start, count, out):                     # "field" = a function of the flat index
            calls.append((start, count))
            out.copy_(torch.arange(start, start + count, dtype=torch.float32) * 0.5 + 1.0)
        res = {}
        for schedule in ("direct", "collective"):
            calls.clear()
            full = torch.full((n,), -7.0)
            evaluate_gathered_overlapped(evaluate_chunk, full, n, unit=unit, chunks=chunks, schedule=schedule,
                                         chunk_unit=2 * unit)
            s, c = slab_bounds(n, world, rank, unit)
            covered = sorted(calls)
            ok = (covered[0][0] == s if covered else c == 0) and sum(k for _, k in covered) == c and \
                all((a - s) % (2 * unit) == 0 or (schedule == "collective" and a - s == (n // unit // world) * unit)
                    for a, k in covered)
            res[schedule] = (ok, full.numpy().copy())
        s, c = slab_bounds(n, world, rank, unit)                   # the plain gather (what bench.py times first)
        local = torch.arange(s, s + c, dtype=torch.float32) * 0.5 + 1.0
        res["after"] = (True, gather_slabs(local, n, unit=unit).numpy().copy())
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,chunks", [(2, (7, 5, 33), 3), (3, (5, 3, 9), 4), (2, (3, 1, 40), 8), (3, (2, 1, 5), 2)])
def test_overlapped_gather_reassembles_the_field(world, shape, chunks):
    """evaluate_gathered_overlapped: slabs of whole rows with an uneven last slab (odd-converted grids never divide),
    more chunks than rows, ranks whose slab is empty; both schedules give the single-rank field on every rank, and
    every rank evaluates exactly its slab in chunks that start at multiples of the chunk unit (two rows here). Also the after-the-fact gather of bench.py."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, shape, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = shape[0] * shape[1] * shape[2]
    want = np.arange(n, dtype=np.float32) * 0.5 + 1.0
    for _rank, res in got:
        for schedule, (ok, full) in res.items():
            assert ok, schedule
            np.testing.assert_array_equal(full, want, err_msg=schedule)


# ---- bench.py's N > 1 extras, walked over gloo (the exact call sequence of the GPU run) ---------------------------------
def _bench_legs_worker(rank, world, port, shape, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from aegolius_amd import distributed as sdist
        n, row_len = shape[0] * shape[1] * shape[2], shape[2]
        start, count = sdist.slab_bounds(n, world, rank, row_len)

        def field(s, c):
            return torch.arange(s, s + c, dtype=torch.float32) * 0.25 - 3.0
        out = torch.empty(count + 5)                              # the rank's field buffer (padded like bench.Run.out)
        out[:count] = field(start, count)
        calls = []

        def evaluate_chunk(cstart, ccount, out_view):
            calls.append((cstart, ccount))
            out_view.copy_(field(cstart, ccount))
        res = bench.reassembly_legs(torch, dist, sdist, out[:count], n, start, count, row_len, evaluate_chunk, dist.barrier,
                                    torch.device("cpu"), 1e-3, chunks=3, chunk_rows=2)
        # every re-evaluation stayed inside the rank's slab; 2 schedules x (warm-up + timed) passes over the slab
        inside = all(start <= a and a + k <= start + count for a, k in calls)
        q.put((rank, res, inside, sum(k for _a, k in calls), count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (7, 5, 33)), (3, (5, 3, 9)), (3, (2, 1, 5))])
def test_bench_reassembly_legs_over_gloo(world, shape):
    """bench.reassembly_legs — warm-up + timed gather, both overlapped schedules with their warm-up, the NaN-fill /
    own-slab check — on CPU tensors over gloo with uneven (and, for the last case, empty) slabs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_legs_worker, args=(r, world, port, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _rank, res, inside, evaluated, count in got:
        assert inside and evaluated == 4 * count
        assert res["after_compute"]["own_slab_intact"] is True
        for leg in ("overlapped_direct", "overlapped_collective"):
            assert "error" not in res[leg], res[leg]
            assert res[leg]["own_slab_intact"] is True
            assert res[leg]["evaluate_and_gather_ms"] > 0
