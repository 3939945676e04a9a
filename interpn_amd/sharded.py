"""Multi-GPU evaluation: one process per GPU, observation points sharded, grid replicated.

The hot path has no cross-point dependence (src/multilinear/regular.rs:276-280: `out[i]`
depends only on the grid and `obs[.][i]`), so the batch shards trivially:

  * rank r of R evaluates the contiguous index range `shard_bounds(P, R, r)` of every
    `obs[d]` and of `out` (equal sizes +-1);
  * the read-only grid (`vals`, and the axes of a rectilinear grid) is replicated by ONE
    broadcast from rank 0 at set-up (RCCL over xGMI when the process group is "nccl"); there is
    no collective in the evaluation loop;
  * the reference's "abort at the first failing point" contract is kept by taking the minimum
    over ranks of (shard offset + local first failing index) — one 8-byte all-reduce at status
    time, off the data path;
  * results stay sharded on the devices; `concat_on_host` assembles them on rank 0 for callers
    that want the reference's single `out` array.

torch.distributed is plumbing here (process group, broadcast); the evaluation itself goes
through libinterpn_hip.so.  `evaluator_factory` exists so that the sharding logic can be
exercised on CPU with the gloo backend (tests/test_sharded_gloo.py) — the product default is the
HIP `Interpolator`.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(npoints: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, equal (+-1) split: the first `npoints % world` ranks get one extra point."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("invalid rank/world")
    base, extra = divmod(int(npoints), world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def device_for_shard(shard: int, ndevices: int) -> int:
    """Device of shard / handle `shard` for a ONE-process caller that spreads its handles over the
    GPUs it can see: round-robin, so that R handles on D devices put ceil(R / D) or floor(R / D) on
    each and shard r and r + D share one (R <= D: one handle per device).  `ndevices` < 1 (no GPU
    visible, e.g. the CPU test tier) counts as 1."""
    if shard < 0:
        raise ValueError("invalid shard")
    return int(shard) % max(1, int(ndevices))


def replicate_across(first, count: int, ndevices: int | None = None):
    """`count` interpolators for a one-process multi-GPU evaluation: `first` itself, then clones
    made device to device (`interpn_hip_replicate`: xGMI between GPUs of a node) on the devices
    `device_for_shard` names, counted from `first`'s own device.  `ndevices` defaults to what the
    C ABI reports (`interpn_hip_device_count`)."""
    if count < 1:
        raise ValueError("count must be >= 1")
    if ndevices is None:
        from . import _lib

        ndevices = _lib.load().interpn_hip_device_count()
    d0 = first.device()
    return [first] + [first.replicate((d0 + device_for_shard(r, ndevices)) % max(1, ndevices)) for r in range(1, count)]


def broadcast_grid(vals, grids=None, src: int = 0):
    """Replicate the grid from `src` to every rank of the default process group (in place).

    `vals` (and each axis in `grids`) must be a torch tensor already allocated with the right
    shape/dtype on every rank (on the rank's GPU for the nccl/RCCL backend).  Returns them."""
    import torch.distributed as dist

    # Whenever a process group exists the collective is issued, also with a single rank: the RCCL
    # call sequence a 1-GPU box can execute is then the one an 8-GPU node executes.
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":  # RCCL moves device memory only: say so instead of failing inside the collective
            for t in [vals] + list(grids or []):
                if not getattr(t, "is_cuda", False):
                    raise ValueError("broadcast_grid under the nccl (RCCL) backend needs CUDA tensors on the rank's GPU; "
                                     "got a host tensor (allocate with device='cuda', or use the gloo backend)")
        dist.broadcast(vals, src=src)
        for g in grids or []:
            dist.broadcast(g, src=src)
    return vals, grids


def _collective_device(dist, interp=None):
    """Where tensors handed to the default group's collectives must live: the rank's GPU for
    nccl (= RCCL on ROCm) — the device the rank's interpolator lives on, not torch's current
    device (a caller that passed device=local_rank without torch.cuda.set_device would otherwise
    put every rank's tensor on cuda:0) —, the host otherwise."""
    import torch

    if dist.get_backend() == "nccl":
        idx = None
        dev_fn = getattr(interp, "device", None)
        if callable(dev_fn):
            try:
                idx = int(dev_fn())
            except Exception:
                idx = None
        return torch.device("cuda", idx if idx is not None and idx >= 0 else torch.cuda.current_device())
    return torch.device("cpu")


_HOST_GROUP = None  # one gloo group beside an nccl default group, shared by every ShardedInterpolator


def _host_side_group(dist):
    """The group host-memory objects travel through: the default group unless it is nccl (RCCL moves
    device memory only), in which case ONE gloo group over the same ranks, created on first use —
    `new_group` is a collective, so every rank creates it at the same point: its first
    ShardedInterpolator."""
    global _HOST_GROUP
    if dist.get_backend() != "nccl":
        return None
    world = dist.group.WORLD
    if _HOST_GROUP is None or _HOST_GROUP[0] is not world:  # (a process may destroy its group and initialise another)
        _HOST_GROUP = (world, dist.new_group(backend="gloo"))
    return _HOST_GROUP[1]


class ShardedInterpolator:
    """One rank's view of a sharded evaluation."""

    def __init__(self, method, kind, *, dims=None, starts=None, steps=None, grids=None, vals=None,
                 linearize_extrapolation=False, device=-1, dtype=np.float64, evaluator_factory=None):
        import torch.distributed as dist

        self._grouped = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self._grouped else 0
        self.world = dist.get_world_size() if self._grouped else 1
        # Host-side assembly of `out` moves numpy shards: RCCL moves device memory only, so under
        # the nccl backend a gloo group over the same ranks carries them (created here because
        # new_group is itself a collective every rank must join).
        self._host_group = _host_side_group(dist) if self._grouped else None
        self.method, self.kind = method, kind
        if evaluator_factory is None:
            from .handle import Interpolator

            if kind == "regular":
                self._interp = Interpolator.regular(method, dims, starts, steps, vals, linearize_extrapolation,
                                                    device, dtype)
            else:
                self._interp = Interpolator.rectilinear(method, grids, vals, linearize_extrapolation, device, dtype)
        else:
            self._interp = evaluator_factory(method=method, kind=kind, dims=dims, starts=starts, steps=steps,
                                             grids=grids, vals=vals,
                                             linearize_extrapolation=linearize_extrapolation)
        self._offset = 0

    def bounds(self, npoints: int) -> tuple[int, int]:
        return shard_bounds(npoints, self.world, self.rank)

    def eval_shard(self, obs_shard, out_shard=None, global_offset: int = 0):
        """Evaluate this rank's shard (device tensors -> device tensor, asynchronous).
        `global_offset` is the global index of the shard's first point (for error reporting)."""
        self._offset = int(global_offset)
        return self._interp.eval_tensors(obs_shard, out_shard)

    def finish(self) -> None:
        """Synchronise and raise AssertionError("Unrepresentable coordinate value") on EVERY rank
        if any rank saw a failing point; `.first_bad_index` is the smallest global index."""
        import torch
        import torch.distributed as dist

        sentinel = np.iinfo(np.int64).max
        local = sentinel
        msg = "Unrepresentable coordinate value"
        failure = None  # anything that is not the reference's per-point error (HIP fault, OOM, ...)
        try:
            self._interp.finish()
        except AssertionError as e:
            local = self._offset + int(getattr(e, "first_bad_index", 0))
            msg = str(e)
        except Exception as e:  # noqa: BLE001 - every rank must still reach the collective below
            failure = e
        if self._grouped:
            # One MIN all-reduce of (-failed, first bad index): a rank that failed outright still
            # takes part, so the others never hang in the collective; -1 wins the MIN.  Issued
            # with a single rank too (same call sequence at every world size).
            t = torch.tensor([-1 if failure is not None else 0, local], dtype=torch.int64,
                             device=_collective_device(dist, self._interp))
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            any_failed, local = int(t[0].item()) < 0, int(t[1].item())
            if failure is None and any_failed:
                failure = RuntimeError("another rank failed during the sharded evaluation")
        if failure is not None:
            raise failure
        if local != sentinel:
            err = AssertionError(msg)
            err.first_bad_index = local
            raise err

    def concat_on_host(self, out_shard, npoints: int, dst: int = 0):
        """Assemble the full `out` on rank `dst` (numpy array there, None elsewhere).  Host-side
        convenience, never inside a timed loop."""
        import torch
        import torch.distributed as dist

        local = out_shard.detach().cpu() if hasattr(out_shard, "detach") else torch.from_numpy(np.asarray(out_shard))
        if not self._grouped:
            return local.numpy()
        # Shards differ in size by one point and live on the host: ship them as objects over gloo
        # (RCCL cannot move host memory, gloo's tensor gather wants equal sizes).
        parts = [None] * self.world if self.rank == dst else None
        dist.gather_object(local.numpy(), parts, dst=dst, group=self._host_group)
        if self.rank != dst:
            return None
        full = np.concatenate(parts)
        assert full.size == npoints
        return full

    def close(self):
        if hasattr(self._interp, "close"):
            self._interp.close()
