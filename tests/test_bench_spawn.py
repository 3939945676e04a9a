"""bench.py --gpus N must really run N ranks (round-1 verdict: `--gpus` was parsed and ignored).

CPU-only: `--dry-run` skips the GPU work, so what is exercised is the launcher half of the
contract — N fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a process
group that sees all of them, ONE JSON line from rank 0 whose n_gpus is the group's size, and the
exit code of a failing rank reaching the caller."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_gpus_flag_spawns_that_many_ranks(n):
    p = run(["--gpus", str(n), "--steps", "7", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly one record, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["spawned"] is True and rec["steps"] == 7
    assert rec["ranks"] == [[r, r] for r in range(n)]  # LOCAL_RANK = RANK on one node


def test_single_gpu_does_not_spawn():
    p = run(["--gpus", "1", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["spawned"] is False


def test_external_launcher_is_respected():
    """Under torch.distributed.run the ranks already exist: bench.py must not spawn again."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env,
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    rec = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["spawned"] is False
    assert not [ln for ln in outs[1].splitlines() if ln.startswith("{")]  # only rank 0 prints a record


def test_failing_rank_fails_the_run():
    p = run(["--gpus", "2", "--dry-run", "--spawn-timeout", "60"], {"INTERPN_BENCH_DRY_FAIL_RANK": "1"})
    assert p.returncode != 0


def test_record_names_the_devices_of_all_ranks():
    """Every rank contributes (rank, device index, PCI address) to one all-gather; rank 0 writes them as config.devices
    (round-5 review: the record must prove that N ranks sat on N distinct GPUs)."""
    p = run(["--gpus", "8", "--steps", "3", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    devs = rec["config"]["devices"]
    assert [d["rank"] for d in devs] == list(range(8))
    assert [d["cuda_device"] for d in devs] == list(range(8))
    assert len({d["pci"] for d in devs}) == 8 and all(d["pci"] for d in devs)


def test_ranks_sharing_a_device_fail_the_run_unless_asked_for():
    same = {"INTERPN_BENCH_DRY_SAME_PCI": "1"}
    p = run(["--gpus", "2", "--dry-run", "--spawn-timeout", "60"], same)
    assert p.returncode != 0
    assert "distinct device" in (p.stderr + p.stdout)
    p = run(["--gpus", "2", "--dry-run", "--same-device"], same)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert len({d["pci"] for d in rec["config"]["devices"]}) == 1
