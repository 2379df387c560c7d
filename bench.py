#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s of SDF evaluation on the 1024^3-request grid (1025^3 points after the
reference's odd-resolution rule) for the 10-primitive smooth-union tree (BASELINE.json `metric`,
SURVEY.md §8(d) "cfg 2 / north-star" recipe), coordinates resident in HBM, field left in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one evaluation of the whole grid. With N > 1 the flat point index is cut into N
contiguous slabs (= slabs along x), one per rank/GPU, no data-path collective (the path is
pointwise) -> strong scaling of the same grid. The RCCL all-gather that reassembles the field is
timed separately after the timed region and reported under "allgather".

The JSON line carries `roofline` (algorithmic 16 B/point over the live HIP-event kernel time, against
the 8 TB/s HBM peak) and `cpu_baseline` (the NumPy oracle timed on this host on a bounded x-slab
sample of the same grid).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_POINT = 16            # 3 x fp32 coordinate loads + 1 x fp32 store


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
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host_path extra (it launches the "
                    "same kernel on a small array, which skews per-kernel averages under rocprofv3)")
    ap.add_argument("--no-rows", action="store_true", help="do not pass the row-length layout hint (flat 128-point bricks)")
    return ap.parse_args()


def build_workload(name, ns, scenes):
    if name == "cfg1":
        return ns.Sphere(0.5), (2, 2, 2), "cfg1: Sphere(0.5)"
    if name == "cfg2":
        return scenes.cfg2_tree(ns), (2, 2, 2), "cfg2: 10-primitive left-deep SMOOTH_UNION2(0.1) chain, rng 1234"
    if name == "cfg3":
        return scenes.cfg3_chain(ns), (4, 4, 4), "cfg3: Box + elongation/twist/bend/infinite_repetition"
    if name == "cfg4":
        return scenes.cfg4_scene2d(ns), (10, 10), "cfg4: 2-D n-ary UNION of 50 onion/rounded primitives, rng 7"
    return scenes.cfg5_tree(ns), (3, 3, 3), "cfg5: 20-primitive 3-level tree, rng 2049"


def cpu_baseline(scenes, workload, axes, budget_s):
    """Oracle (float64 NumPy restatement of the reference, same operation order and temporaries)
    on whole x-planes of the same grid, default NumPy/BLAS threading."""
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count() or 1
    tree, _size, _desc = build_workload(workload, ns, scenes)
    ny, nz = axes[1].size, axes[2].size
    plane = np.empty((3, ny * nz))
    plane[1] = np.repeat(axes[1].astype(np.float64), nz)
    plane[2] = np.tile(axes[2].astype(np.float64), ny)
    done_pts, spent, k = 0, 0.0, 0
    order = np.linspace(0, axes[0].size - 1, min(axes[0].size, 64)).astype(int)   # planes spread over the grid
    while k < len(order):
        plane[0] = float(axes[0][order[k]])
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            sdf_oracle.evaluate(tree, plane)
        dt = time.perf_counter() - t0
        spent += dt
        done_pts += plane.shape[1]
        k += 1
        if spent + dt > budget_s:
            break
    return {"value": done_pts / spent / 1e6, "unit": "Mpoints/s", "cores": int(threads), "kind": "port",
            "sample": "%d x-planes of %dx%d points of the same grid (%.1f s)" % (k, ny, nz, spent),
            "host_cpus": os.cpu_count()}


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
    # SDFK_BENCH_REHEARSE=1: every rank on device 0 with gloo (reductions on the CPU) — to walk the N > 1 control
    # flow on a one-GPU box; RCCL refuses several ranks on one device. Never set by the driver.
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
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    _engine.lib()  # fail loudly if the HIP extension is missing

    tree, size, desc = build_workload(args.workload, ns, scenes)
    axes64, res = grid_axes(size, (args.grid,) * len(size))
    axes = [a.astype(np.float32) for a in axes64]           # fp32-rounded float64 linspace ("identical grids")
    n_total = int(axes[0].size) * int(axes[1].size) * int(axes[2].size)
    # contiguous slabs of the flat index, whole grid rows each (row = the last axis longer than 1);
    # remainder to the last rank
    row_len = int(axes[2].size) if axes[2].size > 1 else int(axes[1].size)
    per = (n_total // row_len // world) * row_len
    start = rank * per
    count = per if rank < world - 1 else n_total - start

    stride = (count + 255) // 256 * 256                      # 16-byte aligned rows -> dwordx4 loads
    co = torch.empty((3, stride), dtype=torch.float32, device=dev)
    out = torch.empty((stride,), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, start, count, stream=stream)

    low = lower_geometry(tree)
    prog = _engine.Program.from_lowered(low)
    mode = {"interpret": _engine.MODE_INTERPRET, "nocull": _engine.MODE_NOCULL}.get(args.mode, _engine.MODE_SPECIALIZED)
    culled = mode == _engine.MODE_SPECIALIZED and len(low.cull_sites) > 0

    def step():
        prog.eval_device(co.data_ptr(), count, stride, out.data_ptr(), stream=stream, mode=mode,
                         row_len=None if args.no_rows else row_len)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev0, ev1 = _engine.Event(), _engine.Event()
    fence()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_ms(ev1) / args.steps            # HIP events on the launch stream

    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(t[0]), float(t[1])

    # ---- extras outside the timed region ----
    probe_gbps = None
    if rank == 0:
        n4 = count // 4 * 4
        lib = _engine.lib()
        for _ in range(2):
            _engine.check(lib.sdfk_stream_probe(co.data_ptr(), n4, stride, out.data_ptr(), stream), "probe")
        p0, p1 = _engine.Event(), _engine.Event()
        p0.record(stream)
        for _ in range(5):
            _engine.check(lib.sdfk_stream_probe(co.data_ptr(), n4, stride, out.data_ptr(), stream), "probe")
        p1.record(stream)
        probe_gbps = BYTES_PER_POINT * n4 / (p0.elapsed_ms(p1) / 5 * 1e-3) / 1e9
        step()                                                # restore `out`
        torch.cuda.synchronize()

    # the same evaluation straight from the per-axis tables (no coordinate array: 4 B/point) -- a separate line
    grid_path = None
    if rank == 0:
        for _ in range(2):
            prog.eval_grid(axes, start, count, out.data_ptr(), stream=stream, mode=mode)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        for _ in range(5):
            prog.eval_grid(axes, start, count, out.data_ptr(), stream=stream, mode=mode)
        torch.cuda.synchronize()
        gms = (time.perf_counter() - g0) / 5 * 1e3
        grid_path = {"ms": gms, "mpoints_per_s": count / gms / 1e3, "bytes_per_point": 4,
                     "note": "sdfk_eval_grid: coordinates expanded in-kernel from three axis tables; includes the "
                             "per-call table upload and stream sync"}

    # end to end through host memory (PCIe inclusive): create()-style call on a plain (3, M) float32 host array of
    # whole x-planes of the same grid, bounded size — a separate line, never `value`
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path:
        planes = max(1, min(int(axes[0].size), int(2.5e7 // (axes[1].size * axes[2].size))))
        m = planes * int(axes[1].size) * int(axes[2].size)
        hco = np.empty((3, m), dtype=np.float32)
        hco[0] = np.repeat(axes[0][:planes], axes[1].size * axes[2].size)
        hco[1] = np.tile(np.repeat(axes[1], axes[2].size), planes)
        hco[2] = np.tile(axes[2], planes * axes[1].size)
        prog.eval_host(hco, device=local_rank, mode=mode)          # warm-up (allocations)
        h0 = time.perf_counter()
        prog.eval_host(hco, device=local_rank, mode=mode)
        hms = (time.perf_counter() - h0) * 1e3
        host_path = {"ms": hms, "points": m, "mpoints_per_s": m / hms / 1e3, "bytes_over_pcie_per_point": 16,
                     "note": "sdfk_eval_host: pageable host (3, M) float32 in, (M,) float32 out, synchronous staging"}
        del hco

    # the consumers of the field on the field this run has just produced, resident in HBM: separate
    # numbers, never `value`; any failure is reported, not raised (the headline line must survive)
    next_rows = None
    if rank == 0 and world == 1 and not args.no_next_rows:
        try:
            import ctypes
            lib, vp = _engine.lib(), ctypes.c_void_p
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
            import aegolius_amd.cores as ns_cores
            from aegolius_amd import _vector
            tiny = np.zeros((3, 8))
            vf = ns_cores.RadialCylindricalVectorField()
            vf.rotate_phi(np.zeros(8))
            vf.revolution_x(tiny)
            vf.normalize()
            vprog = _vector.program_array(_vector.lower_only(vf.vf, tiny, ())[0])

            def chain():
                _engine.check(lib.sdfk_vec_eval_device(vprog, len(vprog), vp(co.data_ptr()), count, stride, vp(out.data_ptr()), 1,
                                                       stride, 0, vp(vec.data_ptr()), stride, vp(stream)), "sdfk_vec_eval_device")
            chain()                                               # builds the kernel of this chain shape: not timed
            torch.cuda.synchronize()
            sel_ms, grad_ms, chain_ms = best_ms(select), best_ms(gradient), best_ms(chain)
            sel_bytes = 4.0 * count + 8.0 * selected.value
            next_rows = {
                "interior_selection": {"ms": sel_ms, "selected": selected.value, "bytes": sel_bytes,
                                       "frac_of_hbm_peak": sel_bytes / (sel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "gradient_direction": {"ms": grad_ms, "bytes": (12.0 if flat else 16.0) * count,
                                       "frac_of_hbm_peak": (12.0 if flat else 16.0) * count / (grad_ms * 1e-3) / 1e9
                                       / HBM_PEAK_GBPS},
                "vector_chain": {"ms": chain_ms, "bytes": 28.0 * count,
                                 "frac_of_hbm_peak": 28.0 * count / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "note": "sdfk_field_select (+ _finish), sdfk_field_gradient and a 4-instruction vector-field chain "
                        "(sdfk_vec_eval_device) on the resident coordinates / field of this run"}
            del scratch, index, vec
        except Exception as exc:  # noqa: BLE001
            next_rows = {"error": repr(exc)}

    allgather = None
    if world > 1 and not args.no_allgather and not rehearse:
        pad = (n_total - (world - 1) * per)                   # largest slab
        send = out[:pad].contiguous()
        full = torch.empty((world * pad,), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(full, send)               # warm-up (RCCL over xGMI)
        fence()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(full, send)
        fence()
        gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        allgather = {"ms": float(gt[0]) * 1e3, "bytes_per_rank": int(pad * 4),
                     "mpoints_per_s_with_gather": n_total / (elapsed / args.steps + float(gt[0])) / 1e6}
        del full

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed / 1e6
        achieved = BYTES_PER_POINT * count / (kernel_ms_max * 1e-3) / 1e9
        traffic = valu_busy = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("points_per_launch") == count and rec.get("workload") == args.workload:
                    traffic = rec.get("hbm_bytes_per_launch")
                    valu_busy = rec.get("valu_active_frac")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "Mpoints/sec SDF eval, 1024^3 grid, 10-prim smooth-union tree",
            "value": value, "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "grid": "%dx%dx%d (request %d per axis), size %s" % (
                axes[0].size, axes[1].size, axes[2].size, args.grid, tuple(size)),
                       "points": n_total, "points_per_gpu": count, "sharding": "contiguous x-slabs, no collective",
                       "kernel": (("sdfk_spec_t (hiprtc, topology-specialised, exact culling on 128-point bricks)" if args.no_rows else
                                   "sdfk_spec_r (hiprtc, topology-specialised, exact culling on 32x16-point row blocks)")
                                  if culled else
                                  "sdfk_spec_v4 (hiprtc, topology-specialised)") if mode != _engine.MODE_INTERPRET
                       else "sdfk_interp_kernel", "instructions": int(low.code.shape[0]),
                       "cull_sites": int(len(low.cull_sites)) if culled else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms": kernel_ms_max, "bytes_per_point": BYTES_PER_POINT,
                         "stream_probe_gbps": probe_gbps, "valu_active_frac": valu_busy},
        }
        if host_path:
            line["host_path"] = host_path
        if grid_path:
            line["grid_path"] = grid_path
        if next_rows:
            line["next_rows"] = next_rows
        if allgather:
            line["allgather"] = allgather
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(scenes, args.workload, axes, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
