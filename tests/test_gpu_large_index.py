"""64-bit paths at full size (round-3 review, item 6): a batch of more than 2^32 points through the
bricked 3-D multilinear kernel, and a grid of more than 2^32 elements through the runtime-N kernel
(`k_generic`, the only one that indexes the grid with 64 bits).  Both need most of a 288 GB MI355X;
they skip themselves on a device with less free memory."""

import numpy as np
import pytest

from tests.helpers import synthetic_case

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch

    free, _total = torch.cuda.mem_get_info(0)
    return free


def test_more_than_2_to_32_points_through_eval_device(oracle):
    """2^32 + 1000 points of 3-D multilinear on a 64^3 grid through `interpn_hip_eval_device`
    (137 GB of coordinates and results): 1e5 sampled indices plus the last 1000 bit-equal to the
    oracle; a NaN at index 2^32 + 7 comes back as exactly that index (slot arithmetic in size_t,
    linear_brick.h; the first-failing-index word is 64-bit)."""
    import torch

    import interpn_amd

    npts = (1 << 32) + 1000
    need = 4 * 8 * npts + (4 << 30)
    if _free_bytes() < need:
        pytest.skip(f"needs {need >> 30} GiB of free device memory")
    dev = torch.device("cuda:0")
    case = synthetic_case("linear", "regular", 3, [64, 64, 64], 16, 4242, np.float64, specials=False)
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    lo = [float(case.starts[d]) - 0.02 for d in range(3)]
    hi = [float(case.starts[d] + case.steps[d] * (case.dims[d] - 1)) + 0.02 for d in range(3)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    obs = []
    for d in range(3):
        t = torch.empty(npts, dtype=torch.float64, device=dev)
        for a in range(0, npts, 1 << 28):  # filled in pieces: keeps every torch kernel below 2^31 elements
            b = min(npts, a + (1 << 28))
            t[a:b].uniform_(lo[d], hi[d], generator=gen)
        obs.append(t)
    out = torch.full((npts,), -9.0, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(5)
    idx = np.unique(np.concatenate([rng.integers(0, npts, 100_000), np.arange(npts - 1000, npts),
                                    np.arange((1 << 32) - 500, (1 << 32) + 500), np.arange(0, 1000)]))
    tidx = torch.from_numpy(idx).to(dev)
    sub = [np.ascontiguousarray(o[tidx].cpu().numpy()) for o in obs]
    want = np.zeros(idx.size)
    oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, sub, want)
    # the sweep kernel (what a batch of this size takes by itself: 5.6e6 rounds of 768 points,
    # linear_sweep.h) and the brick kernel
    for sweep, kernel in ((-1, "k_linear_sweep"), (0, "k_linear_brick")):
        it.set_option("sweep", sweep)
        out.fill_(-9.0)
        it.eval_tensors(obs, out)
        it.finish()
        assert kernel in it.kernel_name()
        got = out[tidx].cpu().numpy()
        assert np.array_equal(got, want), kernel
    bad = (1 << 32) + 7
    obs[1][bad] = float("nan")
    obs[2][npts - 3] = float("inf")  # a later failure must not win
    for sweep in (0, -1):
        it.set_option("sweep", sweep)
        it.eval_tensors(obs, out)
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            it.finish()
        assert ei.value.first_bad_index == bad
    it.close()


def test_grid_of_more_than_2_to_32_elements_through_the_runtime_n_kernel():
    """A 1024 x 2048 x 2049 f32 grid (4.297e9 elements, 17.2 GB, filled on the device): more
    elements than 32-bit offsets reach, so the handle evaluates with `k_generic` (64-bit strides).
    No host copy of such a grid fits a test, so the check is exactness at nodes: with start 0 and
    a power-of-two step every node coordinate is exact, floc is the node index, t = 0, and
    multilinear interpolation returns the node's value bit for bit
    (src/multilinear/regular.rs:414-425, :378-402) — for nodes all over the grid, the last
    element (flat index 2^32 + 2 097 151) included; plus one cell centre against the mean of its
    eight corners."""
    import torch

    import interpn_amd

    dims = [1024, 2048, 2049]
    nvals = dims[0] * dims[1] * dims[2]
    assert nvals > (1 << 32)
    if _free_bytes() < 4 * nvals + (6 << 30):
        pytest.skip("needs 24 GiB of free device memory")
    dev = torch.device("cuda:0")

    def node_value(flat):  # exact in f32: a 20-bit integer (the product stays far below 2^63)
        return ((flat * 40503) >> 3) & 0xFFFFF

    vals = torch.empty(nvals, dtype=torch.float32, device=dev)
    for a in range(0, nvals, 1 << 27):
        b = min(nvals, a + (1 << 27))
        flat = torch.arange(a, b, dtype=torch.int64, device=dev)
        vals[a:b] = (((flat * 40503) >> 3) & 0xFFFFF).to(torch.float32)
        del flat
    step = np.float32(2.0**-6)
    it = interpn_amd.Interpolator.regular("linear", dims, np.zeros(3, dtype=np.float32), np.full(3, step, dtype=np.float32), vals,
                                          False, 0, np.float32)
    rng = np.random.default_rng(11)
    n = 200_000
    ijk = np.stack([rng.integers(0, dims[d], n) for d in range(3)], axis=1)
    ijk[:8] = [[dims[0] - 1, dims[1] - 1, dims[2] - 1], [0, 0, 0], [dims[0] - 1, 0, 0], [0, dims[1] - 1, dims[2] - 1],
               [1023, 2047, 0], [1023, 0, 2048], [512, 1024, 1024], [1023, 2047, 2047]]
    flat = (ijk[:, 0].astype(np.int64) * dims[1] + ijk[:, 1]) * dims[2] + ijk[:, 2]
    assert flat[0] == nvals - 1 and flat[0] >= (1 << 32)
    obs = [torch.from_numpy((ijk[:, d].astype(np.float32) * step)).to(dev) for d in range(3)]
    out = it.eval_tensors(obs)
    it.finish()
    assert "k_generic" in it.kernel_name(), it.kernel_name()
    want = node_value(flat).astype(np.float32)
    assert np.array_equal(out.cpu().numpy(), want)
    # a cell centre in the far corner: t = 1/2 in every dimension -> the mean of the 8 corners (exact in f32 for 20-bit integers)
    c = np.array([dims[0] - 2, dims[1] - 2, dims[2] - 2])
    centre = [torch.tensor([(c[d] + 0.5) * float(step)], dtype=torch.float32, device=dev) for d in range(3)]
    got = float(it.eval_tensors(centre).cpu()[0])
    it.finish()
    corners = [node_value(((c[0] + a) * dims[1] + (c[1] + b)) * dims[2] + (c[2] + e)) for a in (0, 1) for b in (0, 1) for e in (0, 1)]
    assert got == float(np.float32(sum(corners) / 8.0))
    it.close()
