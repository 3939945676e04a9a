"""Multi-GPU sharding logic on CPU: world_size 2 and 3 over the gloo backend.

No GPU here, so the per-rank evaluator is the oracle (allowed in tests/: it is the checker
standing in for the device, the thing under test is the sharding/broadcast/status plumbing of
interpn_amd/sharded.py).  The sharded result must equal the single-process result bit for bit."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from interpn_amd.sharded import ShardedInterpolator, broadcast_grid, shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 1000, 100_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class _OracleEvaluator:
    """Test double for the device interpolator: same eval_tensors/finish surface, CPU oracle inside."""

    def __init__(self, method, kind, dims, starts, steps, grids, vals, linearize_extrapolation):
        self.method, self.dims, self.starts, self.steps = method, dims, starts, steps
        self.vals = vals.numpy().copy()
        self._err = None

    def eval_tensors(self, obs, out=None):
        from oracle import pyoracle

        o = [t.numpy() for t in obs]
        res = np.zeros(o[0].size)
        try:
            pyoracle.linear_regular(self.dims, self.starts, self.steps, self.vals, o, res)
        except pyoracle.OracleError as e:
            self._err = e
        return torch.from_numpy(res)

    def finish(self):
        if self._err is not None:
            e, self._err = self._err, None
            err = AssertionError(str(e))
            err.first_bad_index = e.first_bad
            raise err


def _worker(rank, world, port, tmpdir, inject_nan):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, P = 12, 10_007
        dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, 2.0 / (n - 1))
        # rank 0 owns the grid; the others receive it by broadcast
        vals = torch.zeros(n**3, dtype=torch.float64)
        if rank == 0:
            vals.copy_(torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, n**3)))
        broadcast_grid(vals)
        obs_full = [np.random.default_rng(10 + d).uniform(-1.1, 1.1, P) for d in range(3)]
        if inject_nan:
            obs_full[1][7000] = np.nan  # lands in the last rank's shard
            obs_full[0][9000] = np.inf
        sh = ShardedInterpolator("linear", "regular", dims=dims, starts=starts, steps=steps, vals=vals,
                                 evaluator_factory=lambda **kw: _OracleEvaluator(**kw))
        lo, hi = sh.bounds(P)
        out = sh.eval_shard([torch.from_numpy(o[lo:hi].copy()) for o in obs_full], global_offset=lo)
        if inject_nan:
            try:
                sh.finish()
                raise SystemExit("expected an error on every rank")
            except AssertionError as e:
                assert str(e) == "Unrepresentable coordinate value"
                assert e.first_bad_index == 7000, e.first_bad_index
        else:
            sh.finish()
            full = sh.concat_on_host(out, P, dst=0)
            if rank == 0:
                np.save(os.path.join(tmpdir, "sharded.npy"), full)
                np.save(os.path.join(tmpdir, "vals.npy"), vals.numpy())
                np.save(os.path.join(tmpdir, "obs.npy"), np.stack(obs_full))
            else:
                assert full is None
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(tmp_path, oracle, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npy")
    vals = np.load(tmp_path / "vals.npy")
    obs = np.load(tmp_path / "obs.npy")
    n = 12
    want = np.zeros(obs.shape[1])
    oracle.linear_regular([n] * 3, np.full(3, -1.0), np.full(3, 2.0 / (n - 1)), vals, list(obs), want)
    assert np.array_equal(got, want)


def test_first_bad_index_is_global_minimum(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)


def test_device_assignment_with_a_faked_device_count():
    """Where the single-process multi-GPU callers (and the `-m gpu` tests) put handle i: round-robin over
    the visible devices, counted from the first handle's device — checked here without a GPU: a fake
    interpolator records the devices `replicate_across` asks `interpn_hip_replicate` for."""
    from interpn_amd.sharded import device_for_shard, replicate_across
    from tests.test_rccl_gpu import rccl_world_size

    assert [device_for_shard(i, 1) for i in range(4)] == [0, 0, 0, 0]
    assert [device_for_shard(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert [device_for_shard(i, 0) for i in range(3)] == [0, 0, 0]  # no GPU visible counts as one
    with pytest.raises(ValueError):
        device_for_shard(-1, 8)

    class Fake:
        def __init__(self, dev):
            self.dev = dev

        def device(self):
            return self.dev

        def replicate(self, device):
            return Fake(device)

    assert [h.device() for h in replicate_across(Fake(0), 8, ndevices=8)] == list(range(8))
    assert [h.device() for h in replicate_across(Fake(0), 8, ndevices=1)] == [0] * 8
    assert [h.device() for h in replicate_across(Fake(0), 3, ndevices=2)] == [0, 1, 0]
    assert [h.device() for h in replicate_across(Fake(5), 4, ndevices=8)] == [5, 6, 7, 0]
    with pytest.raises(ValueError):
        replicate_across(Fake(0), 0, ndevices=8)
    assert [rccl_world_size(n) for n in (0, 1, 2, 8, 16)] == [1, 1, 2, 8, 8]
