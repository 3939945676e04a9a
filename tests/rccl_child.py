"""Child process of tests/test_rccl_gpu.py: the multi-GPU call sequence of interpn_amd/sharded.py
executed over the nccl (= RCCL) backend on device tensors, at whatever world size the launcher
set (1 on the one-GPU test box: same calls, same code path as on an 8-GPU node).

Started as a FRESH process (`python -m tests.rccl_child`), so that the process group, the RCCL
communicator and the HIP context are created the way a `torch.distributed.run` rank creates them.
Prints one JSON line with what was executed; any mismatch against the oracle exits non-zero."""

from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main() -> int:
    import torch
    import torch.distributed as dist

    from interpn_amd.sharded import ShardedInterpolator, broadcast_grid
    from oracle import pyoracle  # the checker (tests/ may use it)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # INTERPN_TEST_BACKEND=gloo + INTERPN_TEST_SAME_DEVICE=1: several ranks sharing the one GPU of the
    # test box (RCCL refuses two ranks on one device): the same sharded call sequence with real
    # handles, the grid staged through the host for the broadcast.
    backend = os.environ.get("INTERPN_TEST_BACKEND", "nccl")
    if os.environ.get("INTERPN_TEST_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where broadcast tensors live
    executed = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": []}
    try:
        n, P = 24, 200_003
        rng = np.random.default_rng(5)
        g = np.linspace(-1.0, 1.0, n)
        step = g[1] - g[0]
        axes_host = []
        for _ in range(3):
            j = (rng.random(n) - 0.5) * 0.5 * step
            j[0] = j[-1] = 0.0
            axes_host.append(g + j)
        vals_host = rng.uniform(-1.0, 1.0, n**3)
        obs_full = [np.random.default_rng(20 + d).uniform(-1.05, 1.05, P) for d in range(3)]

        # Rank 0 owns the grid; every rank receives vals AND the rectilinear axes as CUDA tensors
        # through RCCL broadcasts (one per array), then builds its interpolator on the broadcast
        # buffer itself (INTERPN_HIP_MEM_DEVICE: the handle borrows it, no second copy).
        vals = torch.zeros(n**3, dtype=torch.float64, device=cdev)
        axes = [torch.zeros(n, dtype=torch.float64, device=cdev) for _ in range(3)]
        if rank == 0:
            vals.copy_(torch.from_numpy(vals_host))
            for a, h in zip(axes, axes_host):
                a.copy_(torch.from_numpy(h))
        broadcast_grid(vals, axes)
        torch.cuda.synchronize()
        if backend != "nccl":
            vals = vals.to(dev)  # the handle is built on a device buffer either way
        assert np.array_equal(vals.cpu().numpy(), vals_host)
        grids = [a.cpu().numpy() for a in axes]
        assert all(np.array_equal(a, b) for a, b in zip(grids, axes_host))
        executed["broadcast"] = {"vals": int(vals.numel()), "axes": [int(a.numel()) for a in axes]}

        for kind in ("regular", "rectilinear"):
            kw = dict(dims=[n] * 3, starts=np.full(3, -1.0), steps=np.full(3, step)) if kind == "regular" else dict(grids=grids)
            sh = ShardedInterpolator("linear", kind, vals=vals, device=local_rank, **kw)
            lo, hi = sh.bounds(P)
            obs_dev = [torch.from_numpy(o[lo:hi].copy()).to(dev) for o in obs_full]
            out = sh.eval_shard(obs_dev, global_offset=lo)
            sh.finish()  # CUDA-tensor MIN all-reduce of (failed, first bad index)
            full = sh.concat_on_host(out, P, dst=0)
            if rank == 0:
                want = np.zeros(P)
                if kind == "regular":
                    pyoracle.linear_regular([n] * 3, np.full(3, -1.0), np.full(3, step), vals_host, obs_full, want)
                else:
                    pyoracle.linear_rectilinear(axes_host, vals_host, obs_full, want)
                assert full is not None and np.array_equal(full, want), f"{kind}: sharded result differs from the oracle"
            else:
                assert full is None
            case = {"kind": kind, "points": P, "kernel": sh._interp.kernel_name(), "bitwise_equal": True}

            if kind == "regular":
                # the reference's abort-at-first-bad-point contract across shards: the error every
                # rank raises carries the smallest GLOBAL index (here inside the last rank's shard)
                bad_at = P - 1234
                obs_bad = [o.copy() for o in obs_full]
                obs_bad[1][bad_at] = np.nan
                obs_bad[0][bad_at + 100] = np.inf
                obs_dev = [torch.from_numpy(o[lo:hi].copy()).to(dev) for o in obs_bad]
                sh.eval_shard(obs_dev, global_offset=lo)
                try:
                    sh.finish()
                    raise SystemExit("expected 'Unrepresentable coordinate value' on every rank")
                except AssertionError as e:
                    assert str(e) == "Unrepresentable coordinate value", str(e)
                    assert e.first_bad_index == bad_at, (e.first_bad_index, bad_at)
                case["first_bad_index"] = bad_at
                # and the handle is usable again afterwards
                sh.eval_shard([torch.from_numpy(o[lo:hi].copy()).to(dev) for o in obs_full], global_offset=lo)
                sh.finish()
            executed["cases"].append(case)
            sh.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(executed), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
