#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s of SDF evaluation on the 1024^3-request grid (1025^3 points after the
reference's odd-resolution rule) for the 10-primitive smooth-union tree (BASELINE.json `metric`,
SURVEY.md §8(d) "cfg 2 / north-star" recipe), coordinates resident in HBM, field left in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one evaluation of the whole grid. With N > 1 the flat point index is cut into N
contiguous slabs (= slabs along x), one per rank/GPU, no data-path collective (the path is
pointwise) -> strong scaling of the same grid. Reassembling the field on every rank (RCCL over xGMI) is
timed separately after the timed region and reported under "allgather", both after the fact and
overlapped chunk-wise with the evaluation.

The JSON line carries
  roofline      algorithmic 16 B/point over the live HIP-event kernel time, against the 8 TB/s HBM peak
  verified      a random sample of the field the timed region produced, checked against the oracle
  cpu_baseline  the NumPy oracle timed on this host on a bounded x-slab sample of the same grid
  grid_512      the same tree on the 512^3-request grid (north_star names both sizes)
  first_call    cold / warm-cache latency from program creation to the first field (hiprtc JIT)
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_POINT = 16            # 3 x fp32 coordinate loads + 1 x fp32 store
PMC_RECORDS = [os.path.join("profiles", "r04_pmc_traffic.json"),   # static: counters cannot be read from inside a run
               os.path.join("profiles", "r03_pmc_traffic.json"), os.path.join("profiles", "r02_pmc_traffic.json")]
VALU_ROOF = os.path.join("profiles", "r04_valu_roof.json")          # calibrated VALU roof (tools/valu_calib.hip, tools/valu_roof.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024, help="requested resolution per axis (odd-converted)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--mode", default="auto", choices=["auto", "interpret", "nocull"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the field-consumer extras (selection, gradient)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configs (N = 1 extra)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host_path extra (it launches the "
                    "same kernel on a small array, which skews per-kernel averages under rocprofv3)")
    ap.add_argument("--no-rows", action="store_true", help="do not pass the row-length layout hint (flat 128-point bricks)")
    ap.add_argument("--no-extras", action="store_true", help="headline, roofline and verification only (profiling runs)")
    ap.add_argument("--extras-timeout", type=float, default=240.0, help="N > 1: seconds the collective extras may take "
                    "before the line is printed without them")
    return ap.parse_args()


def build_workload(name, ns):
    from aegolius_amd import workloads
    return workloads.build(name, ns)


def _cpu_plane_worker(args):
    """One x-plane of the grid through the oracle (a worker process of cpu_baseline: NumPy only, no GPU)."""
    workload, x, ax1, ax2 = args
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    tree = _CPU_TREES.get(workload)
    if tree is None:
        tree = _CPU_TREES[workload] = build_workload(workload, ns)[0]
    ny, nz = ax1.size, ax2.size
    plane = np.empty((3, ny * nz))
    plane[0] = x
    plane[1] = np.repeat(ax1.astype(np.float64), nz)
    plane[2] = np.tile(ax2.astype(np.float64), ny)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        sdf_oracle.evaluate(tree, plane)
    return time.perf_counter() - t0


_CPU_TREES = {}


def cpu_baseline(workload, axes, budget_s):
    """Oracle (float64 NumPy restatement of the reference, same operation order and temporaries) on whole x-planes of
    the same grid. The reference's NumPy path is single-threaded but for a 3x3 BLAS product, so the host's cores are used
    the way a user of the reference would use them: one process per core of this GPU's share of the host (16), each
    evaluating planes with one BLAS thread. `single_process` is one process with NumPy's default threading."""
    import multiprocessing as mp
    cores = max(1, min(16, os.cpu_count() or 1))
    ny, nz = int(axes[1].size), int(axes[2].size)
    order = np.linspace(0, axes[0].size - 1, min(axes[0].size, 4096)).astype(int)       # planes spread over the grid
    t1 = _cpu_plane_worker((workload, float(axes[0][order[0]]), axes[1], axes[2]))      # one plane, this process
    single = {"value": ny * nz / t1 / 1e6, "unit": "Mpoints/s", "planes": 1}
    per_core = max(1, int(0.8 * budget_s / max(t1, 1e-3)))
    planes = [float(axes[0][order[(7 * k) % len(order)]]) for k in range(per_core * cores)]
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update({k: "1" for k in saved})
    try:
        # (a ProcessPoolExecutor: a worker that cannot start raises BrokenProcessPool instead of being respawned for ever;
        #  spawn, not fork: this process holds a GPU)
        from concurrent.futures import ProcessPoolExecutor
        pool = ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn"))
        try:
            warm = [(workload, planes[0], axes[1], axes[2])] * cores
            list(pool.map(_cpu_plane_worker, warm, timeout=120 + 20 * t1))                   # imports, tree: not timed
            t0 = time.perf_counter()
            busy = list(pool.map(_cpu_plane_worker, [(workload, x, axes[1], axes[2]) for x in planes], chunksize=1,
                                 timeout=60 + 3 * budget_s))
            wall = time.perf_counter() - t0
            pool.shutdown(wait=True)
        except BaseException:
            # a worker that stalls must not hold this process in shutdown(wait=True): end the workers, then leave
            for proc in list(getattr(pool, "_processes", {}).values()):
                try:
                    proc.kill()
                except Exception:  # noqa: BLE001
                    pass
            pool.shutdown(wait=False, cancel_futures=True)
            raise
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {"value": len(planes) * ny * nz / wall / 1e6, "unit": "Mpoints/s", "cores": cores, "kind": "port",
            "sample": "%d x-planes of %dx%d points of the same grid on %d worker processes (%.1f s wall, %.1f s of "
                      "oracle time)" % (len(planes), ny, nz, cores, wall, float(np.sum(busy))),
            "single_process": single, "host_cpus": os.cpu_count()}


def verify_sample(workload, axes, start, out, count, samples=20000, seed=11):
    """The field the timed region left in `out`, at `samples` random points of this rank's slab, against the oracle
    (float64 on the fp32-rounded coordinates: "identical grids"); tolerance of the parity tests, 1e-6 * max(1, |ref|)."""
    import torch
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    tree, _size, _desc = build_workload(workload, ns)
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(count, size=min(samples, count), replace=False))
    got = out[torch.from_numpy(idx).to(out.device)].cpu().numpy().astype(np.float64)
    flat = idx + start
    n1, n2 = int(axes[1].size), int(axes[2].size)
    co = np.stack([axes[0][flat // (n1 * n2)], axes[1][(flat // n2) % n1], axes[2][flat % n2]]).astype(np.float64)
    with np.errstate(all="ignore"):
        ref = sdf_oracle.evaluate(tree, co)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    return {"points": int(idx.size), "max_rel_err": float(err.max()), "tolerance": 1e-6,
            "violations": int(np.count_nonzero(err > 1e-6)), "against": "oracle/sdf_oracle.evaluate (float64)",
            "what": "random sample of the buffer the timed steps wrote"}


_FIRST_CALL = r"""
import json, sys, time
t_import = time.perf_counter()
sys.path.insert(0, {root!r})
import numpy as np
import bench
import aegolius_amd.cores as ns
from aegolius_amd import _engine
_engine.require_gpu()
co, _ = ns.generate_grid((2, 2, 2), (128, 128, 128))       # configs[0] size: where the reference needs 2.6 s per call
ns.Sphere(0.25).create(co[:, :4096].copy())                  # device context, allocator: not part of the tree's latency
t0 = time.perf_counter()
tree, _size, _desc = bench.build_workload({workload!r}, ns)
first = tree.create(co)
t1 = time.perf_counter()
second = tree.create(co)
t2 = time.perf_counter()
builds, build_s = _engine.jit_stats()
print(json.dumps(dict(first_create_s=t1 - t0, second_create_s=t2 - t1, hiprtc_builds_so_far=builds,
                      same_bits=bool(np.array_equal(first, second)))))
"""


def first_call_latency(workload, budget_s=120.0):
    """Program creation -> first field, in fresh processes (no torch import), on the 129^3 grid through the drop-in
    API (`tree.create(co)` on a generate_grid array): cold with the disk cache off (AUTO: the interpreter kernel
    serves the call while hiprtc builds in the background), cold when the call has to wait for the specialised kernel
    (SDFK_ASYNC_JIT=0), and with a warm on-disk cache."""
    import tempfile
    script = _FIRST_CALL.format(root=ROOT, workload=workload)
    out = {"grid": "129^3 via tree.create(generate_grid(...))"}
    deadline = time.perf_counter() + budget_s                    # one budget for the four processes together

    def run(tag, **env):
        e = dict(os.environ)
        e.update(env)
        left = deadline - time.perf_counter()
        if left < 5.0:
            out[tag] = {"skipped": "first_call budget of %.0f s spent" % budget_s}
            return
        try:
            res = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True,
                                 timeout=min(60.0, left))
            out[tag] = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as exc:  # noqa: BLE001
            out[tag] = {"error": repr(exc)}
    with tempfile.TemporaryDirectory() as tmp:
        run("cold_auto", SDFK_CACHE_DIR="off")
        run("cold_wait_for_specialised", SDFK_CACHE_DIR="off", SDFK_ASYNC_JIT="0")
        run("fill_cache", SDFK_CACHE_DIR=tmp, SDFK_ASYNC_JIT="0")
        run("warm_cache", SDFK_CACHE_DIR=tmp, SDFK_ASYNC_JIT="0")
    out.pop("fill_cache", None)
    return out


class Run:
    """One resident grid on this rank: coordinates, field, and the timed loop."""

    def __init__(self, torch, dist, engine, prog, axes, world, rank, dev, red_dev, mode, use_rows):
        self.torch, self.dist, self.engine, self.prog = torch, dist, engine, prog
        self.axes, self.world, self.rank, self.dev, self.red_dev, self.mode = axes, world, rank, dev, red_dev, mode
        self.n_total = int(axes[0].size) * int(axes[1].size) * int(axes[2].size)
        # contiguous slabs of the flat index, whole grid rows each (row = the last axis longer than 1);
        # remainder to the last rank
        self.row_len = int(axes[2].size) if axes[2].size > 1 else int(axes[1].size)
        self.per = (self.n_total // self.row_len // world) * self.row_len
        self.start = rank * self.per
        self.count = self.per if rank < world - 1 else self.n_total - self.start
        self.stride = (self.count + 255) // 256 * 256            # 16-byte aligned rows -> dwordx4 loads
        self.co = torch.empty((3, self.stride), dtype=torch.float32, device=dev)
        self.out = torch.empty((self.stride,), dtype=torch.float32, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.use_rows = use_rows
        self.xy = False
        self.no_planes = os.environ.get("SDFK_BENCH_NO_PLANES") == "1"      # A/B: blocks of 16 rows regardless of planes
        engine.grid_fill(self.co.data_ptr(), self.stride, axes, self.start, self.count, stream=self.stream)

    def step(self):
        flat = self.axes[2].size == 1
        if self.xy:                                             # two-row call: z = 0 by contract, 12 B/point (flat grids)
            self.prog.eval_device_xy(self.co.data_ptr(), self.count, self.stride, self.out.data_ptr(), stream=self.stream,
                                     mode=self.mode, row_len=self.row_len if self.use_rows else None)
            return
        self.prog.eval_device(self.co.data_ptr(), self.count, self.stride, self.out.data_ptr(), stream=self.stream,
                              mode=self.mode, row_len=self.row_len if self.use_rows else None, flat=flat,
                              plane_rows=None if flat or self.no_planes else int(self.axes[1].size),
                              first_row_in_plane=0 if flat else (self.start // self.row_len) % int(self.axes[1].size))

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """-> (elapsed seconds over `steps`, kernel ms per step, median, min of the per-step ms): MAX over ranks."""
        for _ in range(warmup):
            self.step()
        ev = [self.engine.Event() for _ in range(steps + 1)]
        self.fence()
        t0 = time.perf_counter()
        ev[0].record(self.stream)
        for k in range(steps):
            self.step()
            ev[k + 1].record(self.stream)                      # HIP events on the launch stream
        self.fence()
        elapsed = time.perf_counter() - t0
        kernel_ms = ev[0].elapsed_ms(ev[steps]) / steps
        per_step = sorted(ev[k].elapsed_ms(ev[k + 1]) for k in range(steps))
        t = self.torch.tensor([elapsed, kernel_ms, per_step[len(per_step) // 2], per_step[0]], dtype=self.torch.float64,
                              device=self.red_dev)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0]), float(t[1]), float(t[2]), float(t[3])


def reassembly_legs(torch, dist, sdist, local, n_total, start, count, row_len, evaluate_chunk, fence, red_dev,
                    compute_s, chunks=8, chunk_rows=32, exercise_transport=False):
    """The three ways of putting the whole field on every rank, each warmed up once and timed once (MAX over ranks),
    never part of `value`: the plain all-gather after the evaluation and the two schedules of
    distributed.evaluate_gathered_overlapped, with a check that the rank's own slab arrived intact.
    `local`: this rank's slab (a view of its field buffer); `evaluate_chunk(start, count, out)` re-evaluates a piece.
    Module-level so that the CPU test (gloo, CPU tensors) walks the very sequence the GPU run walks."""
    res = {"bytes_per_rank": int(count * 4), "compute_ms_per_step": compute_s * 1e3}

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    full = sdist.gather_slabs(local, n_total, unit=row_len)           # warm-up (communicators, buffers)
    fence()
    g0 = time.perf_counter()
    full = sdist.gather_slabs(local, n_total, unit=row_len)
    fence()
    gt = max_over_ranks(time.perf_counter() - g0)
    intact = bool(torch.equal(full[start:start + count], local))
    res["after_compute"] = {"gather_ms": gt * 1e3, "schedule": "all_gather_into_tensor after the evaluation",
                            "mpoints_per_s_with_gather": n_total / (compute_s + gt) / 1e6, "own_slab_intact": intact}
    del full
    full = torch.empty(n_total, dtype=local.dtype, device=local.device)
    for schedule in ("direct", "collective"):
        try:
            def once():
                sdist.evaluate_gathered_overlapped(evaluate_chunk, full, n_total, unit=row_len, chunks=chunks,
                                                   schedule=schedule, chunk_unit=chunk_rows * row_len, local=local,
                                                   exercise_transport=exercise_transport)
            full.fill_(float("nan"))
            once()                                                    # warm-up
            fence()
            full.fill_(float("nan"))
            fence()
            o0 = time.perf_counter()
            once()
            fence()
            ot = max_over_ranks(time.perf_counter() - o0)
            # the whole field arrived (no NaN left) and this rank's part of it is the slab it computed
            same = torch.tensor([1.0], dtype=torch.float64, device=red_dev)
            if bool(torch.isnan(full).any()) or not torch.equal(full[start:start + count], local[:count]):
                same[0] = 0.0
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            res["overlapped_" + schedule] = {"evaluate_and_gather_ms": ot * 1e3, "chunks": chunks,
                                             "mpoints_per_s_with_gather": n_total / ot / 1e6,
                                             "own_slab_intact": bool(same[0] > 0)}
        except Exception as exc:  # noqa: BLE001
            res["overlapped_" + schedule] = {"error": repr(exc)}
    return res


class Extras:
    """Everything outside the timed region: a failure is reported in the line, never raised; the line records how
    long each extra took and — should the watchdog fire — which one was in flight."""

    def __init__(self, line):
        self.line, self.in_flight, self.seconds = line, None, {}
        self.lock = threading.Lock()          # `line` is written here and serialised by the watchdog thread

    def __call__(self, name, fn):
        self.in_flight = name
        t0 = time.perf_counter()
        try:
            val = fn()
        except Exception as exc:  # noqa: BLE001
            val = {"error": repr(exc)}
        with self.lock:
            self.seconds[name] = round(time.perf_counter() - t0, 3)
            self.in_flight = None
            if self.line is not None:
                if val is not None:
                    self.line[name] = val
                self.line["extras_s"] = dict(self.seconds)
        return val

    def snapshot(self):
        """-> (JSON text of the line as it stands, name of the extra in flight): what the watchdog prints."""
        with self.lock:
            return (json.dumps(self.line) if self.line is not None else None), self.in_flight


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched through torch.distributed.run" % args.gpus)
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path")
    # SDFK_BENCH_REHEARSE=1: every rank on device 0 with gloo (reductions on the CPU, transfers staged through host
    # memory) — walks the whole N > 1 control flow on a one-GPU box; RCCL refuses several ranks on one device.
    # Never set by the driver.
    rehearse = world > 1 and os.environ.get("SDFK_BENCH_REHEARSE") == "1"
    torch.cuda.set_device(0 if rehearse else local_rank)
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    red_dev = torch.device("cpu") if rehearse else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()                               # no-op when libsdfk.so is current
    if world > 1:
        dist.barrier()
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd import distributed as sdist
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    _engine.lib()  # fail loudly if the HIP extension is missing

    tree, size, desc = build_workload(args.workload, ns)

    def fp32_axes(request, gsize=size):
        axes64, _res = grid_axes(gsize, (request,) * len(gsize))
        return [a.astype(np.float32) for a in axes64]       # fp32-rounded float64 linspace ("identical grids")
    axes = fp32_axes(args.grid)

    low = lower_geometry(tree)
    prog = _engine.Program.from_lowered(low)
    # the timed region always waits for the specialised kernel (AUTO would serve the first calls from the interpreter
    # while hiprtc builds in the background: measured separately under "first_call")
    mode = {"interpret": _engine.MODE_INTERPRET, "nocull": _engine.MODE_NOCULL}.get(args.mode, _engine.MODE_SPECIALIZED)

    def kernel_name(lowered, rows):
        if mode == _engine.MODE_INTERPRET:
            return "sdfk_interp_kernel"
        if mode == _engine.MODE_SPECIALIZED and len(lowered.cull_sites) > 0:
            return ("sdfk_spec_r (hiprtc, topology-specialised, exact culling on 32x16-point row blocks)" if rows else
                    "sdfk_spec_t (hiprtc, topology-specialised, exact culling on 128-point bricks)")
        return "sdfk_spec_v4 (hiprtc, topology-specialised)"
    culled = mode == _engine.MODE_SPECIALIZED and len(low.cull_sites) > 0

    run = Run(torch, dist, _engine, prog, axes, world, rank, dev, red_dev, mode, not args.no_rows)
    n_total, count, start, stride, row_len, stream = run.n_total, run.count, run.start, run.stride, run.row_len, run.stream
    t_build = time.perf_counter()
    run.step()                                                # builds (or loads from the disk cache) this flavour
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    elapsed, kernel_ms_max, median_ms, min_ms = run.timed(args.steps, args.warmup)

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed / 1e6
        achieved = BYTES_PER_POINT * count / (kernel_ms_max * 1e-3) / 1e9
        traffic = valu_busy = traffic_source = None
        for record in PMC_RECORDS:
            pmc = os.path.join(ROOT, record)
            if not os.path.exists(pmc):
                continue
            try:
                rec = json.load(open(pmc))
                if rec.get("points_per_launch") == count and rec.get("workload") == args.workload:
                    traffic = rec.get("hbm_bytes_per_launch")
                    valu_busy = rec.get("valu_active_frac")
                    traffic_source = "%s (static record of a separate rocprofv3 --pmc run of this kernel, commit %s; not " \
                                     "measured by this run)" % (record, rec.get("commit", "?"))
                    break
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "Mpoints/sec SDF eval, 1024^3 grid, 10-prim smooth-union tree",
            "value": value, "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "median_ms": median_ms, "min_ms": min_ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "grid": "%dx%dx%d (request %d per axis), size %s" % (
                axes[0].size, axes[1].size, axes[2].size, args.grid, tuple(size)),
                       "points": n_total, "points_per_gpu": count, "sharding": "contiguous x-slabs, no collective",
                       "kernel": kernel_name(low, not args.no_rows), "instructions": int(low.code.shape[0]),
                       "cull_sites": int(len(low.cull_sites)) if culled else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kernel_ms_max, "kernel_ms_median": median_ms, "kernel_ms_min": min_ms,
                         "bytes_per_point": BYTES_PER_POINT},
            "first_kernel_build_s": t_build,
        }
        # the second roof SURVEY 8(d) asks for beside the HBM one: VALU issue, from per-class instruction counters priced with
        # measured issue costs (a static record of separate counter runs, like `traffic`; low / mid / high = how the half of the
        # instructions no class counter covers is priced). The old `valu_active_frac` (SQ_ACTIVE_INST_VALU * 4 / cycles) is not
        # a fraction — that counter ticks once per instruction, and instructions cost 2.4 to 8.2 cycles — and is gone.
        try:
            vr = json.load(open(os.path.join(ROOT, VALU_ROOF)))["workloads"].get(args.workload)
            if vr and world == 1 and vr.get("points_per_launch") == count:
                clock_ghz = vr["cycles_per_launch"] / (kernel_ms_max * 1e-3) / 1e9     # (counter runs clock a little lower)
                line["roofline"]["valu"] = {"bound": "valu issue", "wave_instructions": vr["valu_wave_instructions"],
                                            "lane_instructions_per_point": vr["lane_instructions_per_point"],
                                            "busy_frac": vr["valu_busy_frac"], "mean_issue_cycles": vr["mean_issue_cycles_per_instruction"],
                                            "implied_clock_ghz_at_this_kernel_time": clock_ghz, "source": VALU_ROOF}
        except Exception:  # noqa: BLE001
            pass

    # Whatever happens in the extras, rank 0 prints the headline line and every rank leaves: a watchdog thread stays armed
    # from here until the line has been printed (N > 1: `--extras-timeout` for the collective extras, re-armed for rank 0's
    # solo extras; N = 1: one generous limit for everything). When it fires the line says which extra was in flight
    # (`extras_timeout`) and the process exits NON-ZERO whatever the state of the headline — a rank may be holding a GPU
    # with a collective in flight, and that must never read as a successful run: 3 when the headline is in the line and
    # its `verified` sample had come back clean (the timed region's number stands on its own), 4 otherwise.
    printed = threading.Event()
    extra = Extras(line)
    state = {}                                                   # what the watchdog needs to know on every rank

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            print(json.dumps(line), flush=True)

    def watchdog(limit_s):
        code = 4
        try:
            v = state.get("verified")                            # (every rank holds the all-reduced verdict; `line` is rank 0's)
            if isinstance(v, dict) and v.get("violations") == 0 and "error" not in v:
                code = 3
            if rank == 0 and not printed.is_set():
                with extra.lock:
                    line["extras_timeout"] = {"in_flight": extra.in_flight, "after_s": limit_s,
                                              "finished": dict(extra.seconds), "exit_code": code}
                text, _ = extra.snapshot()
                printed.set()
                print(text, flush=True)
            sys.stderr.write("[bench] rank %d: extras did not return within %.0f s (in flight: %s); exit code %d\n"
                             % (rank, limit_s, extra.in_flight, code))
            sys.stderr.flush()
        finally:
            os._exit(code)                                       # always reached, whatever the serialisation did

    def arm(limit_s):
        # (ranks other than 0 fire 5 s later: the launcher ends every rank once one has left, and rank 0 prints first)
        t = threading.Timer(limit_s + (0.0 if rank == 0 else 5.0), watchdog, args=(limit_s,))
        t.daemon = True
        t.start()
        return t
    timer = None
    if not args.no_extras or world > 1:
        timer = arm(args.extras_timeout if world > 1 else max(args.extras_timeout, 900.0))

    # ---- the field the timed steps wrote, against the oracle (every rank checks its slab; rank 0 reports) ----
    def verified():
        v = verify_sample(args.workload, axes, start, run.out, count)
        if world > 1:
            t = torch.tensor([v["max_rel_err"], float(v["violations"])], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v["max_rel_err"], v["violations"] = float(t[0]), int(t[1])
            v["what"] += " (every rank samples its own slab; worst rank reported)"
        return v
    state["verified"] = extra("verified", verified)

    if not args.no_extras:
        lib = _engine.lib()

        # ---- plain (3,N)->(N) stream with the same access width: what this box delivers ----
        def stream_probe():
            n4 = count // 4 * 4
            scratch = torch.empty_like(run.out)
            for _ in range(2):
                _engine.check(lib.sdfk_stream_probe(run.co.data_ptr(), n4, stride, scratch.data_ptr(), stream), "probe")
            p0, p1 = _engine.Event(), _engine.Event()
            p0.record(stream)
            for _ in range(5):
                _engine.check(lib.sdfk_stream_probe(run.co.data_ptr(), n4, stride, scratch.data_ptr(), stream), "probe")
            p1.record(stream)
            return BYTES_PER_POINT * n4 / (p0.elapsed_ms(p1) / 5 * 1e-3) / 1e9
        if rank == 0:
            try:
                line["roofline"]["stream_probe_gbps"] = stream_probe()
            except Exception as exc:  # noqa: BLE001
                line["roofline"]["stream_probe_gbps"] = None
                line["roofline"]["stream_probe_error"] = repr(exc)

        # ---- the 512^3-request grid, same tree, same slab partition (north_star names both sizes) ----
        def grid_512():
            ax = fp32_axes(512)
            r2 = Run(torch, dist, _engine, prog, ax, world, rank, dev, red_dev, mode, not args.no_rows)
            # 0.4 ms per step: 150 steps after 75 of warm-up (30 + 60 ms), so that the GPU has reached its clocks before the
            # timed region — 30 steps after 10 read 0.416 ms where 200 after 50 read 0.397 (profiles/r03_grid512_*)
            steps2 = max(150, args.steps)
            e2, k2, med2, min2 = r2.timed(steps2, 75)
            v = verify_sample(args.workload, ax, r2.start, r2.out, r2.count, samples=5000)
            return {"grid": "%dx%dx%d" % (ax[0].size, ax[1].size, ax[2].size), "points": r2.n_total, "steps": steps2,
                    "value": r2.n_total * steps2 / e2 / 1e6, "unit": "Mpoints/s", "ms_per_step": e2 / steps2 * 1e3,
                    "kernel_ms": k2, "kernel_ms_median": med2, "kernel_ms_min": min2,
                    "roofline_frac": BYTES_PER_POINT * r2.count / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "verified": {"points": v["points"], "max_rel_err": v["max_rel_err"], "violations": v["violations"]}}
        if len(size) == 3 and args.grid != 512:
            extra("grid_512", grid_512)

    if not args.no_extras and rank == 0 and world == 1:
        # ---- every other BASELINE config on this box, each with its own verification sample (N = 1 only) ----
        def other_configs():
            from aegolius_amd import workloads
            res = {}
            plan = [("cfg1", 1024, "1024^3 request (the config's own 128^3 grid is 2 M points: launch-bound)"),
                    ("cfg3", 1024, None), ("cfg4", 16384, None), ("cfg5", 1024, "1024^3 request"),
                    ("cfg5", 2048, "the config's own 2048^3 request on ONE GPU (137 GB resident)")]
            for name, request, note in plan:
                if name == args.workload and request == args.grid:
                    continue
                key = name if request != 2048 else name + "_2049"
                try:
                    t_tree, t_size, t_desc = workloads.build(name, ns)
                    ax = fp32_axes(request, t_size)
                    pts = int(ax[0].size) * int(ax[1].size) * int(ax[2].size)
                    free_b, _total_b = torch.cuda.mem_get_info()
                    if 16.0 * pts * 1.02 > free_b:
                        res[key] = {"skipped": "needs %.0f GB, %.0f GB free" % (16e-9 * pts, free_b * 1e-9)}
                        continue
                    t_low = lower_geometry(t_tree)
                    t_prog = _engine.Program.from_lowered(t_low)
                    r = Run(torch, dist, _engine, t_prog, ax, 1, 0, dev, red_dev, mode, not args.no_rows)
                    t0 = time.perf_counter()
                    r.step()
                    torch.cuda.synchronize()
                    build_s = time.perf_counter() - t0
                    # about 30 ms of warm-up and 60 ms of timed launches per config: a 0.9 ms kernel measured over 10
                    # launches after 5 is still on the GPU's way up to its clocks (cfg 4: 0.93-0.98 against 0.89-0.90 ms)
                    probe_e, _k, _med, _mn = r.timed(3, 2)
                    one = max(probe_e / 3, 1e-4)
                    steps_o = int(min(100, max(10, round(0.06 / one))))
                    e, k, med, mn = r.timed(steps_o, int(min(50, max(5, round(0.03 / one)))))
                    v = verify_sample(name, ax, 0, r.out, r.count, samples=5000)
                    res[key] = {"workload": t_desc, "grid": "%dx%dx%d" % (ax[0].size, ax[1].size, ax[2].size),
                                "points": pts, "steps": steps_o, "value": pts * steps_o / e / 1e6, "unit": "Mpoints/s",
                                "ms_per_step": e / steps_o * 1e3, "kernel_ms": k, "kernel_ms_median": med,
                                "kernel_ms_min": mn, "kernel": kernel_name(t_low, not args.no_rows),
                                "roofline_frac": BYTES_PER_POINT * pts / (k * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "first_kernel_build_s": build_s,
                                "verified": {"points": v["points"], "max_rel_err": v["max_rel_err"],
                                             "violations": v["violations"]}}
                    if note:
                        res[key]["note"] = note
                    if ax[2].size == 1:
                        # the same flat grid WITHOUT its z row (sdfk_eval_device_rows2d_xy: z = 0 is the call's contract):
                        # 12 B/point — its own line with its own roofline, never mixed with the 16 B/point one above
                        keep = r.out.clone()
                        r.xy = True
                        r.step()
                        torch.cuda.synchronize()
                        e2, k2, med2, mn2 = r.timed(steps_o, int(min(50, max(5, round(0.03 / one)))))
                        res[key + "_xy"] = {"workload": t_desc + " — x and y rows only (z = 0 by contract)", "points": pts,
                                            "steps": steps_o, "value": pts * steps_o / e2 / 1e6, "unit": "Mpoints/s",
                                            "ms_per_step": e2 / steps_o * 1e3, "kernel_ms": k2, "kernel_ms_median": med2,
                                            "kernel_ms_min": mn2, "bytes_per_point": 12,
                                            "roofline_frac": 12.0 * pts / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                            "same_field_as_the_three_row_call": bool(torch.equal(keep[:r.count], r.out[:r.count]))}
                        del keep
                    del r, t_prog
                    torch.cuda.empty_cache()
                except Exception as exc:  # noqa: BLE001
                    res[key] = {"error": repr(exc)}
            return res
        if not args.no_other_configs:
            extra("other_configs", other_configs)

        # the same evaluation straight from the per-axis tables (no coordinate array: 4 B/point) -- a separate line
        def grid_path():
            scratch = torch.empty_like(run.out)
            for _ in range(2):
                prog.eval_grid(axes, start, count, scratch.data_ptr(), stream=stream, mode=mode)
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            for _ in range(5):
                prog.eval_grid(axes, start, count, scratch.data_ptr(), stream=stream, mode=mode)
            torch.cuda.synchronize()
            gms = (time.perf_counter() - g0) / 5 * 1e3
            return {"ms": gms, "mpoints_per_s": count / gms / 1e3, "bytes_per_point": 4,
                    "note": "sdfk_eval_grid: coordinates expanded in-kernel from three axis tables; includes the "
                            "per-call table upload and stream sync"}
        extra("grid_path", grid_path)

        # end to end through host memory (PCIe inclusive): create()-style call on a plain (3, M) float32 host array of
        # whole x-planes of the same grid, bounded size — a separate line, never `value`
        def host_path():
            planes = max(1, min(int(axes[0].size), int(1.0e8 // (axes[1].size * axes[2].size))))
            m = planes * int(axes[1].size) * int(axes[2].size)
            hco = np.empty((3, m), dtype=np.float32)
            hco[0] = np.repeat(axes[0][:planes], axes[1].size * axes[2].size)
            hco[1] = np.tile(np.repeat(axes[1], axes[2].size), planes)
            hco[2] = np.tile(axes[2], planes * axes[1].size)
            prog.eval_host(hco, device=local_rank, mode=mode)          # warm-up (allocations, pinned staging)
            # The call, not the garbage collection of its result: round 3's 2.1 -> 1.66 Gpoints/s in this line was the
            # un-mapping of the PREVIOUS 400 MB result inside the timed statement (18 ms next to a HIP context, 23 ms with
            # torch's threads around: tools/host_path_ab.py, profiles/r04_host_path_ab.txt) — results are kept until after the
            # clock stops, and what freeing them costs is reported beside it.
            best, keep = 1e30, []
            for _ in range(3):
                h0 = time.perf_counter()
                keep.append(prog.eval_host(hco, device=local_rank, mode=mode))
                best = min(best, (time.perf_counter() - h0) * 1e3)
            f0 = time.perf_counter()
            n_kept = len(keep)
            keep.clear()
            free_ms = (time.perf_counter() - f0) * 1e3 / n_kept
            return {"ms": best, "points": m, "mpoints_per_s": m / best / 1e3, "bytes_over_pcie_per_point": 16,
                    "gbytes_per_s_over_pcie": 16.0 * m / best / 1e6, "free_result_ms": free_ms,
                    "mpoints_per_s_including_the_free": m / (best + free_ms) / 1e3,
                    "note": "sdfk_eval_host: pageable host (3, M) float32 in, (M,) float32 out; free_result_ms = handing one "
                            "result array back to the OS afterwards (not part of the call)"}
        if not args.no_host_path:
            extra("host_path", host_path)

        # the consumers of the field on the field this run has just produced, resident in HBM: separate
        # numbers, never `value`
        def next_rows():
            import ctypes
            vp = ctypes.c_void_p
            out = run.out
            torch.cuda.synchronize()
            scratch = torch.empty(lib.sdfk_field_select_scratch(count), dtype=torch.uint8, device=dev)
            selected = ctypes.c_int64(0)
            _engine.check(lib.sdfk_field_select(vp(out.data_ptr()), count, 0.0, None, 0, ctypes.byref(selected),
                                                vp(scratch.data_ptr()), vp(stream)), "sdfk_field_select")
            index = torch.empty(max(selected.value, 1), dtype=torch.int64, device=dev)
            vec = torch.empty((3, stride), dtype=torch.float32, device=dev)

            def best_ms(fn, reps=3):
                best = 1e30
                for _ in range(reps):
                    e0, e1 = _engine.Event(), _engine.Event()
                    e0.record(stream)
                    fn()
                    e1.record(stream)
                    best = min(best, e0.elapsed_ms(e1))
                return best

            def select():
                _engine.check(lib.sdfk_field_select(vp(out.data_ptr()), count, 0.0, None, 0, ctypes.byref(selected),
                                                    vp(scratch.data_ptr()), vp(stream)), "sdfk_field_select")
                _engine.check(lib.sdfk_field_select_finish(count, selected.value, vp(index.data_ptr()), selected.value,
                                                           vp(scratch.data_ptr()), vp(stream)), "sdfk_field_select_finish")

            flat = axes[2].size == 1                              # 2-D grids: (1, n0, n1), two components
            dims = (1, axes[0].size, axes[1].size) if flat else (axes[0].size, axes[1].size, axes[2].size)

            def gradient():
                _engine.check(lib.sdfk_field_gradient(vp(out.data_ptr()), dims[0], dims[1], dims[2], 2 if flat else 3, 1,
                                                      vp(vec.data_ptr()), stride, vp(stream)), "sdfk_field_gradient")
            # a vector-field chain (§8(f).4) on the run's coordinates with the run's field as the per-point angle:
            # radial-cylindrical field, turned about z by the SDF value, revolved about x, normalised (28 B/point)
            from aegolius_amd import _vector
            tiny = np.zeros((3, 8))
            vf = ns.RadialCylindricalVectorField()
            vf.rotate_phi(np.zeros(8))
            vf.revolution_x(tiny)
            vf.normalize()
            vprog = _vector.program_array(_vector.lower_only(vf.vf, tiny, ())[0])

            def chain():
                _engine.check(lib.sdfk_vec_eval_device(vprog, len(vprog), vp(run.co.data_ptr()), count, stride, vp(out.data_ptr()), 1,
                                                       stride, 0, vp(vec.data_ptr()), stride, vp(stream)), "sdfk_vec_eval_device")
            chain()                                               # builds the kernel of this chain shape: not timed
            torch.cuda.synchronize()
            sel_ms, grad_ms, chain_ms = best_ms(select), best_ms(gradient), best_ms(chain)
            sel_bytes = 4.0 * count + 8.0 * selected.value
            # GenericGeometry.point_cloud's mask WITHOUT a field: evaluation kernels write flag bits, compaction reads them
            fscratch = torch.empty(lib.sdfk_eval_select_scratch(count, row_len), dtype=torch.uint8, device=dev)
            fsel = ctypes.c_int64(0)
            flat_i = 1 if flat else 0

            def fused():
                _engine.check(lib.sdfk_eval_device_select(prog.handle, vp(run.co.data_ptr()), count, stride, row_len, flat_i, 0.0,
                                                          vp(index.data_ptr()), index.numel(), ctypes.byref(fsel),
                                                          vp(fscratch.data_ptr()), vp(stream), mode), "sdfk_eval_device_select")
            fused()
            fused_ms = best_ms(fused)
            fused_same = bool(fsel.value == selected.value)
            fused_bytes = 12.0 * count + 8.0 * selected.value
            return {
                "fused_evaluate_and_select": {"ms": fused_ms, "selected": fsel.value, "same_count_as_field_path": fused_same,
                                              "bytes": fused_bytes, "evaluate_then_select_ms": kernel_ms_max + sel_ms,
                                              "frac_of_hbm_peak": fused_bytes / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                              "note": "sdfk_eval_device_select: 12 B/point in, flag bits + 8 B per selected "
                                                      "point out; evaluation kernel + count + scan + scatter, one stream sync"},
                "interior_selection": {"ms": sel_ms, "selected": selected.value, "bytes": sel_bytes,
                                       "frac_of_hbm_peak": sel_bytes / (sel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "gradient_direction": {"ms": grad_ms, "bytes": (12.0 if flat else 16.0) * count,
                                       "frac_of_hbm_peak": (12.0 if flat else 16.0) * count / (grad_ms * 1e-3) / 1e9
                                       / HBM_PEAK_GBPS},
                "vector_chain": {"ms": chain_ms, "bytes": 28.0 * count,
                                 "frac_of_hbm_peak": 28.0 * count / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "note": "sdfk_field_select (+ _finish), sdfk_field_gradient and a 4-instruction vector-field chain "
                        "(sdfk_vec_eval_device) on the resident coordinates / field of this run"}
        if not args.no_next_rows:
            extra("next_rows", next_rows)

    # ---- reassembling the field on every rank (RCCL over xGMI; rehearsal: gloo through host copies): never `value` ----
    if world > 1 and not args.no_allgather and not args.no_extras:
        def evaluate_chunk(cstart, ccount, out_view):
            off = cstart - start
            flat = axes[2].size == 1
            prog.eval_device(run.co.data_ptr() + 4 * off, ccount, stride, out_view.data_ptr(), stream=stream, mode=mode,
                             row_len=row_len if not args.no_rows else None, flat=flat,
                             plane_rows=None if flat else int(axes[1].size),
                             first_row_in_plane=0 if flat else (cstart // row_len) % int(axes[1].size))
        extra("allgather", lambda: reassembly_legs(torch, dist, sdist, run.out[:count], n_total, start, count, row_len,
                                                   evaluate_chunk, run.fence, red_dev, elapsed / args.steps))

    # ---- N > 1: the last collective is the closing barrier; the process group is taken down BEFORE rank 0's solo extras, so
    # that no peer sits in an RCCL barrier (whose own watchdog would abort it, and torchrun rank 0 with it) while rank 0 times
    # the CPU baseline. The watchdog stays armed over barrier and teardown.
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if timer is not None:
        timer.cancel()
    if rank == 0:
        # rank 0 alone from here on: no collectives below, every leg bounded by its own time-outs, and a watchdog of its own
        # over the two of them (first_call <= 4 fresh processes, cpu_baseline ~ --cpu-seconds + imports)
        solo = arm(240.0 + 4.0 * args.cpu_seconds)
        if not args.no_extras:
            extra("first_call", lambda: first_call_latency(args.workload))
        if args.cpu_seconds > 0:
            extra("cpu_baseline", lambda: cpu_baseline(args.workload, axes, args.cpu_seconds))
        emit()
        solo.cancel()


if __name__ == "__main__":
    main()
