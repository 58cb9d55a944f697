"""The real multi-GPU path: engine + compact all-gather over RCCL with one process per GPU. Runs only where the box has
two or more GPUs (the builder's box has one: skipped there; the driver's 8-GPU node exercises it). Ranks are FRESH child processes
(python -m torch.distributed.run): this pytest process may have initialised the GPU and must not exec another program."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_engine_and_compact_allgather_over_rccl():
    import torch

    ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if ndev < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % ndev)
    world = 2 if ndev < 4 else 4  # at most 4 ranks on the card's process budget
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["world_size"] == world and out["backend"] == "nccl" and out["all_ranks_ok"] and out["mismatches_rank0"] == []
    assert out["order"][:3] == [["gather", 0], ["gather", 1], ["consume", 0]]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearsal():
    """The same worker with TWO ranks on ONE GPU (backend gloo on device tensors: RCCL refuses two ranks per device): world > 1 logic of
    the engine + double-buffered compact gather — two different shards, every rank expands both and compares with direct-mode digests —
    on the hardware the builder has. Not a measurement (gloo stages through the host)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BLSW_TEST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["world_size"] == 2 and out["backend"] == "gloo" and out["all_ranks_ok"] and out["mismatches_rank0"] == []


@pytest.mark.gpu
def test_bench_two_ranks_gloo_child():
    """`python bench.py --gpus 2 --steps 4` as the driver would start it for N = 2 (bench.py spawns its own child torch.distributed.run),
    with BLSW_TEST_BACKEND=gloo so that both ranks may share the one GPU of the builder's box: the line must carry the world size, the
    backend, per-rank rates, generation-only next to the all-gather leg, and right witnesses. Not a measurement."""
    env = dict(os.environ, BLSW_TEST_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "128", "--allgather-steps", "4",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["witness_ok"] and out["config"]["results_ok"] and out["config"]["result_shards_gathered"]
    assert out["per_rank"]["backend"] == "gloo" and len(out["per_rank"]["seconds"]) == 2
    assert out["per_rank"]["instances_per_s_min"] <= out["per_rank"]["instances_per_s_max"]
    assert out["value_generation_only"] == out["value"] and out["value"] >= 2 * out["per_rank"]["instances_per_s_min"] * 0.999
    ag = out["allgather"]
    assert ag["backend"] == "gloo" and ag["world"] == 2 and ag["ranks"] == 2 and "error" not in ag and out["value_with_allgather"] > 0
