"""Child process of tests/test_gpu_rccl.py (also runnable by hand on a GPU box): a process group of ONE rank on the
"nccl" backend (= RCCL on ROCm) and bench.py's reassembly legs on DEVICE tensors — communicator creation,
`all_gather_into_tensor`, the grouped send / receive batch, the communication stream and its events — with a real
evaluation (the north-star tree on a small grid) as the chunk evaluator. Prints one JSON line.

    python tests/rccl_world1_child.py [grid_request] [port]

Started fresh by the test, so that nothing has touched the GPU before the process group exists."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    request = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    port = sys.argv[2] if len(sys.argv) > 2 else "29611"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = port
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", device_id=dev)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": time.perf_counter() - t0}
    try:
        import bench
        import aegolius_amd.cores as ns
        from aegolius_amd import _engine, distributed as sdist
        from aegolius_amd._lower import lower_geometry
        from aegolius_amd.cores.helper_functions import grid_axes
        tree, size, _desc = bench.build_workload("cfg2", ns)
        axes = [a.astype(np.float32) for a in grid_axes(size, (request,) * 3)[0]]
        prog = _engine.Program.from_lowered(lower_geometry(tree))
        run = bench.Run(torch, dist, _engine, prog, axes, 1, 0, dev, dev, _engine.MODE_SPECIALIZED, True)
        e, k, _med, _mn = run.timed(3, 1)                          # the barrier of Run.fence is an RCCL collective here
        want = run.out[:run.count].clone()

        def evaluate_chunk(cstart, ccount, out_view):
            prog.eval_device(run.co.data_ptr() + 4 * cstart, ccount, run.stride, out_view.data_ptr(), stream=run.stream,
                             mode=_engine.MODE_SPECIALIZED, row_len=run.row_len, flat=False,
                             plane_rows=int(axes[1].size), first_row_in_plane=(cstart // run.row_len) % int(axes[1].size))
        legs = bench.reassembly_legs(torch, dist, sdist, run.out[:run.count], run.n_total, run.start, run.count, run.row_len,
                                     evaluate_chunk, run.fence, dev, e / 3, chunks=4, chunk_rows=32,
                                     exercise_transport=True)
        res["legs"] = legs
        res["field_unchanged"] = bool(torch.equal(run.out[:run.count], want))
        # the one-byte-per-point exchange `signed` needs between slabs (distributed._TorchComm): RCCL on device tensors
        comm = sdist._TorchComm()
        mask = torch.arange(run.n_total, device=dev, dtype=torch.int64).remainder(251).to(torch.uint8)
        comm.allgather_bytes(mask.data_ptr(), 0, run.n_total, run.n_total)
        res["allgather_bytes_ok"] = bool(torch.equal(comm._keep, mask))
        res["allreduce_min"] = comm.allreduce_min(-0.25)
        torch.cuda.synchronize()
        dist.barrier()
        res["ok"] = True
    except Exception as exc:  # noqa: BLE001
        import traceback
        res["ok"] = False
        res["error"] = repr(exc)
        res["trace"] = traceback.format_exc()
    finally:
        try:
            dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001
            res["destroy_error"] = repr(exc)
    print(json.dumps(res), flush=True)
    return 0 if res.get("ok") else 1


if __name__ == "__main__":
    sys.exit(main())
