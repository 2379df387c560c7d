"""The RCCL transport of the multi-GPU path on the one GPU a test box has: a process group of ONE
rank on the "nccl" backend, bench.py's reassembly legs on device tensors. What the gloo tests of
tests/test_distributed_cpu.py cannot reach — communicator creation, `all_gather_into_tensor` and a grouped
send / receive batch on device memory, the communication stream and its events — runs here; the peer-to-peer links
themselves need the driver's 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_reassembly_legs_over_rccl_with_one_rank(engine):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ONE fresh child process: nothing in it has touched the GPU before init_process_group("nccl")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py"), "128", str(_free_port())],
                          env=env, capture_output=True, text=True, timeout=420)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert lines, "no result line; rc %s\n%s" % (proc.returncode, proc.stderr[-3000:])
    res = json.loads(lines[-1])
    assert res.get("ok"), res.get("trace") or res
    assert proc.returncode == 0
    assert res["backend"] == "nccl" and res["world"] == 1
    legs = res["legs"]
    assert legs["after_compute"]["own_slab_intact"] is True
    for leg in ("overlapped_direct", "overlapped_collective"):
        assert "error" not in legs[leg], legs[leg]
        assert legs[leg]["own_slab_intact"] is True
        assert legs[leg]["evaluate_and_gather_ms"] > 0
    assert res["field_unchanged"] is True
    assert res["allgather_bytes_ok"] is True and res["allreduce_min"] == -0.25
    assert "destroy_error" not in res
