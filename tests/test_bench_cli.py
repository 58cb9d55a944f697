"""bench.py host logic without a GPU: --gpus N outside a launcher starts N ranks as a CHILD torch.distributed.run (never an
exec of a process that touched the GPU) and relays rank 0's JSON line; launch groups are balanced; the CPU share honours the
cgroup quota."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_spawn_ranks_builds_a_child_launcher(bench, monkeypatch, capsys):
    seen = {}

    class P:
        returncode = 0
        stdout = "NCCL banner\n" + json.dumps({"metric": "m", "n_gpus": 4}) + "\n"

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return P()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    args = bench.parse_args()
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(args)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert json.loads(capsys.readouterr().out.strip())["n_gpus"] == 4  # only the JSON line reaches stdout


def test_world_size_must_match_gpus(bench, monkeypatch):
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    with pytest.raises(SystemExit, match="WORLD_SIZE"):
        bench.main()


def test_group_balancing_and_cpu_share(bench):
    assert bench.balanced_coalesce(20, 16) == 10 and bench.balanced_coalesce(1024, 16) == 16 and bench.balanced_coalesce(5, 16) == 5
    assert bench.balanced_coalesce(33, 16) == 11
    assert 1 <= bench.host_cores() <= (os.cpu_count() or 1)
