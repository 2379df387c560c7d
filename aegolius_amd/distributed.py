"""Slab sharding of a grid evaluation across the GPUs of one node.

The path is pointwise, so any partition of the flat point index is exact. Rank r of W evaluates the
contiguous range `slab_bounds(N, W, r)` (an x-slab of the grid: the flat index is
n = (ix*ny + iy)*nz + iz, reference cores/helper_functions.py:90-91) with no data-path collective.
Only reassembling the field needs communication: one all-gather (RCCL over xGMI when the process
group backend is "nccl"; gloo on CPU in the tests).

One process per GPU, launched with torch.distributed.run; torch is plumbing here (device memory,
streams, the process group), the evaluation itself is libsdfk.so.
"""
import numpy as np


def slab_bounds(n_total, world_size, rank, unit=1):
    """(start, count) of rank's contiguous slab; the remainder goes to the last rank. Slabs are whole
    multiples of `unit` points (the grid's row length: the row-block culling kernel wants whole rows)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    if unit < 1 or n_total % unit:
        raise ValueError("%d points are not whole units of %d" % (n_total, unit))
    per = (n_total // unit // world_size) * unit
    start = rank * per
    count = per if rank < world_size - 1 else n_total - start
    return start, count


def gather_slabs(local, n_total, group=None, unit=1):
    """All-gather the per-rank slabs (1-D tensors laid out by slab_bounds) into the full field on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, count = slab_bounds(n_total, world, rank, unit)
    if local.numel() != count:
        raise ValueError("rank %d holds %d values, its slab has %d" % (rank, local.numel(), count))
    per = (n_total // unit // world) * unit
    pad = n_total - (world - 1) * per                         # largest slab (the last one)
    send = local
    if count != pad:
        send = torch.zeros(pad, dtype=local.dtype, device=local.device)
        send[:count] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":          # gloo moves host tensors (several ranks on one GPU: the tests)
        host = torch.empty(world * pad, dtype=local.dtype)
        dist.all_gather_into_tensor(host, send.contiguous().cpu(), group=group)
        full = host.to(local.device)
    else:
        full = torch.empty(world * pad, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, send.contiguous(), group=group)
    if pad * world == n_total:
        return full
    out = torch.empty(n_total, dtype=local.dtype, device=local.device)
    for r in range(world):
        s, c = slab_bounds(n_total, world, r, unit)
        out[s:s + c] = full[r * pad:r * pad + c]
    return out


def chunk_bounds(count, chunks, unit=1):
    """Cut `count` points into at most `chunks` contiguous pieces that start at multiples of `unit`:
    [(offset, length)], equal lengths except the last (which also takes what is left of a partial unit), none empty."""
    if unit < 1:
        raise ValueError("unit must be positive")
    if count <= 0:
        return []
    units = -(-count // unit)
    chunks = max(1, min(int(chunks), units))
    step = -(-units // chunks) * unit
    return [(o, min(step, count - o)) for o in range(0, count, step)]


def evaluate_gathered_overlapped(evaluate_chunk, full, n_total, unit=1, chunks=4, group=None, schedule="direct",
                                 chunk_unit=None, local=None, exercise_transport=False):
    """Evaluate this rank's slab and reassemble the field in `full` (an (n_total,) tensor on every rank) WHILE it is
    being computed.

    The slab (slab_bounds with `unit`) is cut into `chunks` pieces that start at multiples of `chunk_unit` points of
    the slab (default `unit`; the bench passes 32 rows so that every piece starts on a 128-byte line of the rank's own
    buffers). `evaluate_chunk(start, count, out)` enqueues the evaluation of the flat range [start, start + count)
    into `out` — a view of `local`, the rank's own slab buffer — on the current stream. As soon as a piece is done its
    exchange starts on a second stream, so the transfer of piece i runs under the evaluation of piece i + 1 and only
    the last piece's transfer is exposed.

    schedule "direct"     each rank sends the piece to every peer and receives the peers' pieces in place, as ONE
                          grouped point-to-point batch (RCCL send/recv): W - 1 transfers per rank that each use their
                          own xGMI link — xGMI is point to point (7 links per GPU), a ring would relay every byte
                          W - 1 times over one link pair. No staging buffer, no padding for the uneven last slab.
    schedule "collective" all_gather_into_tensor per piece into a (W, piece) staging buffer + one strided copy into
                          the field; the last rank's surplus rows travel by one broadcast.

    Transports: device tensors over RCCL (backend "nccl") go as they are; CPU tensors over gloo (the tests) issue the
    same calls in program order; DEVICE tensors over gloo (several ranks rehearsing on one GPU: gloo's TCP transport
    reads host memory only — handing it device pointers is what stalled the round-2 rehearsal) are staged through host
    copies piece by piece, with the same control flow, streams and events around them.
    Ranks are GROUP-local throughout (`group_peer` / `group_src`), so a sub-group works.
    `exercise_transport` (tests): a world of ONE still issues its calls — the per-piece all-gather of "collective", a
    send-to-self / receive-from-self batch for "direct" whose payload then overwrites the piece in the field — so that
    the RCCL communicator, its stream and the event choreography run on a one-GPU box."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if full.numel() != n_total or not full.is_contiguous():
        raise ValueError("`full` must be a contiguous tensor of n_total elements")
    if schedule not in ("direct", "collective"):
        raise ValueError("unknown schedule %r" % (schedule,))
    chunk_unit = chunk_unit or unit
    spans = [slab_bounds(n_total, world, r, unit) for r in range(world)]
    start, count = spans[rank]
    if local is None:
        local = torch.empty(max(count, 1), dtype=full.dtype, device=full.device)
    if local.numel() < count:
        raise ValueError("`local` is smaller than the rank's slab")
    cuda = full.is_cuda
    via_host = cuda and world > 1 and dist.get_backend(group) == "gloo"
    comm = torch.cuda.Stream(device=full.device) if cuda else None
    works = []

    def on_comm(fn):
        """run fn on the communication stream once everything enqueued so far on the current stream is done"""
        if not cuda:
            fn()
            return
        done = torch.cuda.Event()
        done.record()
        comm.wait_event(done)
        with torch.cuda.stream(comm):
            fn()

    def exchange(i, pieces):
        mine = pieces[rank][i] if i < len(pieces[rank]) else None
        if mine is not None:                                   # the rank's own piece into its place in the field
            o, c = mine
            full[start + o:start + o + c].copy_(local[o:o + c])
        if world == 1 and not exercise_transport:
            return
        if world == 1 and schedule == "direct":
            if mine is not None:                               # the piece travels rank -> rank and lands in the field again
                o, c = mine
                back = torch.full((c,), float("nan"), dtype=full.dtype, device=full.device)
                reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, local[o:o + c], group=group, group_peer=rank),
                                               dist.P2POp(dist.irecv, back, group=group, group_peer=rank)])
                for w in reqs:
                    w.wait()
                full[start + o:start + o + c].copy_(back)
            return
        if schedule == "direct":
            ops, landed = [], []
            send = None
            if mine is not None:
                send = local[mine[0]:mine[0] + mine[1]]
                if via_host:
                    comm.synchronize()                         # (this piece is complete: we run on `comm`)
                    send = send.cpu()
            for peer in range(world):
                if peer == rank:
                    continue
                if send is not None:
                    ops.append(dist.P2POp(dist.isend, send, group=group, group_peer=peer))
                if i < len(pieces[peer]):
                    o, c = pieces[peer][i]
                    ps = spans[peer][0]
                    into = full[ps + o:ps + o + c]
                    if via_host:
                        host = torch.empty(c, dtype=full.dtype)
                        landed.append((into, host))
                        into = host
                    ops.append(dist.P2POp(dist.irecv, into, group=group, group_peer=peer))
            if ops:
                reqs = dist.batch_isend_irecv(ops)
                if via_host:
                    for w in reqs:
                        w.wait()
                    for into, host in landed:
                        into.copy_(host)
                else:
                    works.extend(reqs)
        else:
            o, c = pieces[0][i]                                # the common part: the same piece on every rank
            per = spans[0][1]
            if via_host:
                comm.synchronize()
                host = torch.empty((world, c), dtype=full.dtype)
                dist.all_gather_into_tensor(host.view(-1), local[o:o + c].cpu(), group=group)
                stage = host.to(full.device)
            else:
                stage = torch.empty((world, c), dtype=full.dtype, device=full.device)
                dist.all_gather_into_tensor(stage.view(-1), local[o:o + c], group=group)
            full[:world * per].view(world, per)[:, o:o + c].copy_(stage)

    def surplus(lo, hi):                                       # the last slab's rows beyond the common part
        if via_host:
            comm.synchronize()
            host = full[lo:hi].cpu()
            dist.broadcast(host, group=group, group_src=world - 1)
            full[lo:hi].copy_(host)
        else:
            dist.broadcast(full[lo:hi], group=group, group_src=world - 1)

    if schedule == "direct" or world == 1:
        pieces = [chunk_bounds(c, chunks, chunk_unit) for _, c in spans]
        steps = max(len(p) for p in pieces)
    else:
        per = spans[0][1]                                      # every rank cuts the common part alike;
        pieces = [chunk_bounds(per, chunks, chunk_unit)] * world   # the last rank's surplus follows by broadcast
        steps = len(pieces[0])
    mine = pieces[rank]
    for i in range(steps):
        if i < len(mine):
            o, c = mine[i]
            evaluate_chunk(start + o, c, local[o:o + c])
        on_comm(lambda i=i: exchange(i, pieces))
    if schedule == "collective" and world > 1:
        ls, lc = spans[-1]
        per = spans[0][1]
        if lc > per:                                           # surplus rows of the last slab
            if rank == world - 1:
                evaluate_chunk(ls + per, lc - per, local[per:lc])
                on_comm(lambda: full[ls + per:ls + lc].copy_(local[per:lc]))
            on_comm(lambda: surplus(ls + per, ls + lc))
    for w in works:
        w.wait()
    if cuda:
        torch.cuda.current_stream(full.device).wait_stream(comm)
    return full


def evaluate_grid_sharded(geometry, size, resolution, gather=True, group=None, evaluate_slab=None):
    """Evaluate `geometry` on generate_grid(size, resolution), this rank computing only its slab.

    Returns (field, resolution): the full field on every rank when gather=True, else this rank's slab
    (left distributed — the configuration that scales). `evaluate_slab(axes, start,
    count) -> 1-D torch tensor` may replace the built-in GPU evaluator (the CPU tests pass the oracle).
    """
    import torch.distributed as dist
    from .cores.helper_functions import grid_axes
    axes64, res = grid_axes(size, resolution)
    n_total = int(np.prod([a.size for a in axes64]))
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if evaluate_slab is None:
        evaluate_slab = _GpuSlabEvaluator(geometry, group)
    # whole grid rows per slab (rows run along the last axis longer than one point); trees with grid-neighbourhood
    # operators are cut between planes of the first axis (their slabs carry a halo of planes)
    unit = int(axes64[2].size) if axes64[2].size > 1 else int(axes64[1].size)
    if getattr(evaluate_slab, "staged", False):
        unit = int(axes64[1].size) * int(axes64[2].size)
    start, count = slab_bounds(n_total, world, rank, unit)
    local = evaluate_slab([a.astype(np.float32) for a in axes64], start, count)
    if gather and world > 1:
        return gather_slabs(local, n_total, group, unit), res
    return local, res


class _LocalComm:
    """One slab = the whole grid: nothing to exchange."""

    def allreduce_min(self, value):
        return value

    def allgather_bytes(self, ptr, start, count, total):
        if start != 0 or count != total:
            raise ValueError("a single slab must cover the whole grid")
        return ptr


class _TorchComm:
    """What `signed` needs between the slabs of a sharded grid (aegolius_amd._eval.evaluate_slab_staged): the minimum
    of the field over all ranks and an all-gather of ONE BYTE per point (field < grid spacing) — its scan lines cross
    every slab, but that bit is all they read. RCCL moves device tensors directly; gloo (the tests: several ranks on
    one GPU) goes through host copies."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self._keep = None                                     # the gathered mask: alive while the C side reads it

    def allreduce_min(self, value):
        import torch
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if self.on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t[0])

    def allgather_bytes(self, ptr, start, count, total):
        import torch
        import torch.distributed as dist
        from . import _engine
        dev = "cuda" if self.on_device else "cpu"
        spans = torch.zeros((self.world, 2), dtype=torch.int64, device=dev)
        mine = torch.tensor([start, count], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(spans.view(-1), mine, group=self.group)
        spans = spans.cpu().tolist()
        pad = max(1, max(c for _s, c in spans))
        send = torch.zeros(pad, dtype=torch.uint8, device="cuda")
        if count:
            _engine.check(_engine.lib().sdfk_memcpy_d2d(_engine._vp(send.data_ptr()), _engine._vp(ptr), count), "d2d")
        if self.on_device:
            parts = torch.empty(self.world * pad, dtype=torch.uint8, device="cuda")
            dist.all_gather_into_tensor(parts, send, group=self.group)
        else:
            host = torch.empty(self.world * pad, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
            parts = host.cuda()
        full = torch.empty(total, dtype=torch.uint8, device="cuda")
        for r, (s0, c) in enumerate(spans):
            full[s0:s0 + c] = parts[r * pad:r * pad + c]
        torch.cuda.synchronize()
        self._keep = full
        return full.data_ptr()


class _GpuSlabEvaluator:
    """Evaluates a slab straight from the per-axis tables into a torch tensor on the current device."""

    def __init__(self, geometry, group=None):
        from . import _engine
        from ._eval import program_for
        from ._lower import NeedsStage, lower_geometry
        self._engine = _engine
        self.staged = False
        self.exchange = False                                 # the tree contains `signed`: every rank takes part, empty slab or not
        self._group = group
        try:
            self._prog = program_for(lower_geometry(geometry))
        except NeedsStage:
            # conv_averaging / conv_edge_detection: slabs of whole planes with a recomputed halo; signed: the slabs
            # exchange one byte per point (_TorchComm); opaque user code needs the whole field: refused with the reason
            from ._eval import _halo_planes, _plan_stages
            self._lower = lambda **kw: lower_geometry(geometry, **kw)
            stages, _final, _ = _plan_stages(self._lower)
            for _low, node, _key, _params in stages:
                if node.name in ("signed", "signed_old"):
                    self.exchange = True
                elif node.name not in ("conv_averaging", "conv_edge_detection"):
                    raise NotImplementedError(
                        "the tree contains %r, which needs the whole field on one device: evaluate it with "
                        "geometry.create(co) on a single GPU" % (node.name,)) from None
            self._halo = _halo_planes
            self.staged = True

    def _comm(self):
        import torch.distributed as dist
        if not self.exchange:
            return None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1:
            return _TorchComm(self._group)
        return _LocalComm()

    def __call__(self, axes, start, count):
        import torch
        self._engine.require_gpu()
        out = torch.empty(count, dtype=torch.float32, device="cuda")
        if self.staged:
            from ._eval import evaluate_slab_staged
            plane = int(axes[1].size) * int(axes[2].size)
            if start % plane or count % plane:
                raise ValueError("slabs of a tree with grid-neighbourhood operators must be whole planes of the first axis")
            torch.cuda.synchronize()
            if count or self.exchange:
                evaluate_slab_staged(self._lower, axes, start // plane, count // plane, out.data_ptr(), comm=self._comm())
            return out
        stream = torch.cuda.current_stream().cuda_stream
        self._prog.eval_grid(axes, start, count, out.data_ptr(), stream=stream)
        return out


# ---------------------------------------------------------------------------------------------------
# consumers of the field, sharded the same way: no data-path collective either
# ---------------------------------------------------------------------------------------------------
def _world(group):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _grid(size, resolution):
    from .cores.helper_functions import grid_axes
    axes64, res = grid_axes(size, resolution)
    shape = tuple(int(a.size) for a in axes64 if a.size > 1)
    return [a.astype(np.float32) for a in axes64], res, shape


def interior_indices_sharded(geometry, size, resolution, threshold=0.0, group=None, evaluate_slab=None, select_slab=None,
                             world_rank=None):
    """GenericGeometry.point_cloud's mask, sharded: this rank's ascending GLOBAL flat indices with field <= threshold
    (an int64 torch tensor; the ranks' results, concatenated in rank order, are numpy.flatnonzero of the whole field).
    `world_rank=(W, r)` overrides the process group (slabs emulated on one device)."""
    axes, res, shape = _grid(size, resolution)
    world, rank = world_rank or _world(group)
    n_total = int(np.prod(shape))
    start, count = slab_bounds(n_total, world, rank, shape[-1])
    evaluate_slab = evaluate_slab or _GpuSlabEvaluator(geometry)
    if getattr(evaluate_slab, "staged", False):
        raise NotImplementedError("sharded consumers take pointwise trees (no grid-neighbourhood operators)")
    local = evaluate_slab(axes, start, count)
    return (select_slab or _gpu_select)(local, threshold) + start, res


def gradient_direction_sharded(geometry, size, resolution, group=None, evaluate_slab=None, gradient_slab=None,
                               world_rank=None):
    """vector_functions.from_sdf, sharded: this rank's (D, count) slab of unit gradient vectors, count = its whole
    planes of the first axis. Each rank evaluates its slab plus ONE halo plane per inner side (recomputed, not
    exchanged: 2 planes against N/W) so that the central differences across slab boundaries are the single-GPU ones;
    the one-sided differences stay on the global faces. Left distributed, like the field itself."""
    axes, res, shape = _grid(size, resolution)
    world, rank = world_rank or _world(group)
    n_total = int(np.prod(shape))
    plane = n_total // shape[0]
    start, count = slab_bounds(n_total, world, rank, plane)
    evaluate_slab = evaluate_slab or _GpuSlabEvaluator(geometry)
    if getattr(evaluate_slab, "staged", False):
        raise NotImplementedError("sharded consumers take pointwise trees (no grid-neighbourhood operators)")
    gradient_slab = gradient_slab or _gpu_gradient
    p0, p1 = start // plane, (start + count) // plane
    if p1 == p0:
        return gradient_slab(evaluate_slab(axes, 0, 2 * plane), (2,) + shape[1:])[:, :0], res
    h0, h1 = (1 if p0 > 0 else 0), (1 if p1 < shape[0] else 0)
    if p1 - p0 + h0 + h1 < 2:                                  # a one-plane grid cannot be differentiated (numpy agrees)
        raise ValueError("Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements "
                         "are required.")
    ext = evaluate_slab(axes, (p0 - h0) * plane, (p1 - p0 + h0 + h1) * plane)
    vec = gradient_slab(ext, (p1 - p0 + h0 + h1,) + shape[1:])
    return vec[:, h0 * plane:h0 * plane + count], res


def _gpu_select(local, threshold):
    """sdfk_field_select on a torch CUDA tensor -> int64 CUDA tensor of local indices."""
    import ctypes
    import torch
    from . import _engine
    lib, vp = _engine.lib(), _engine._vp
    torch.cuda.synchronize()
    n = local.numel()
    m = ctypes.c_int64(0)
    scratch = torch.empty(lib.sdfk_field_select_scratch(n), dtype=torch.uint8, device=local.device)
    _engine.check(lib.sdfk_field_select(vp(local.data_ptr()), n, float(threshold), None, 0, ctypes.byref(m),
                                        vp(scratch.data_ptr()), None), "sdfk_field_select")
    index = torch.empty(m.value, dtype=torch.int64, device=local.device)
    if m.value:
        _engine.check(lib.sdfk_field_select_finish(n, m.value, vp(index.data_ptr()), m.value, vp(scratch.data_ptr()), None),
                      "sdfk_field_select_finish")
    return index


def _gpu_gradient(ext, shape):
    """sdfk_field_gradient on a torch CUDA tensor holding a slab of `shape` -> (D, points) view of unit vectors."""
    import torch
    from . import _engine
    lib, vp = _engine.lib(), _engine._vp
    torch.cuda.synchronize()
    n = ext.numel()
    stride = (n + 63) // 64 * 64
    dims = (1,) * (3 - len(shape)) + tuple(shape)
    vec = torch.empty((len(shape), stride), dtype=torch.float32, device=ext.device)
    _engine.check(lib.sdfk_field_gradient(vp(ext.data_ptr()), dims[0], dims[1], dims[2], len(shape), 1,
                                          vp(vec.data_ptr()), stride, None), "sdfk_field_gradient")
    return vec[:, :n]


def vector_field_sharded(field, size, resolution, out="vector", group=None, resident=False, evaluate_slab=None,
                         world_rank=None):
    """VectorField.create (or a read-out: out = x | y | z | phi | theta | length) on generate_grid(size, resolution),
    this rank computing only its slab of whole grid rows: pointwise, so no halo and no collective. Per-point operands
    of the chain (angles, second fields) are given for the WHOLE grid and sliced here; a revolution about the grid's own
    coordinates uses the slab's. Returns (slab, resolution): a host array, or a DeviceVectorField / DeviceField when
    `resident`."""
    axes, res, shape = _grid(size, resolution)
    world, rank = world_rank or _world(group)
    n_total = int(np.prod(shape))
    start, count = slab_bounds(n_total, world, rank, shape[-1])
    if evaluate_slab is None:
        from ._vector import evaluate_slab
    return evaluate_slab(field, axes, start, count, out, resident), res
