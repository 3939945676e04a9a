#!/usr/bin/env python3
"""Differential fuzz: random (method, kind, N, axis sizes, dtype, linearize, layout knobs, batch
size, special coordinates) through the C ABI against the CPU oracle, bit for bit.
    python tools/fuzz_parity.py [seconds=120] [seed=0]
Prints one line per failure and a summary; exit code 1 if anything differed."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle
from tests.helpers import run_hip_raw, run_oracle, synthetic_case

KNOBS = ("INTERPN_HIP_BRICKS", "INTERPN_HIP_PPL", "INTERPN_HIP_FORCE_GENERIC", "INTERPN_HIP_GENERIC_RUNTIME",
         "INTERPN_HIP_HOST_CHUNK", "INTERPN_HIP_ITERS_PER_BLOCK", "INTERPN_HIP_AXIS_REGS", "INTERPN_HIP_GENERIC_VEC", "INTERPN_HIP_PERSISTENT",
         "INTERPN_HIP_BINNED", "INTERPN_HIP_DEAL", "INTERPN_HIP_COLUMN", "INTERPN_HIP_COLUMN_THREADS", "INTERPN_HIP_COLUMN_PART",
         "INTERPN_HIP_COLUMN_GROUPS", "INTERPN_HIP_COLUMN_CPP", "INTERPN_HIP_COLUMN_COEF", "INTERPN_HIP_COLUMN_PAD", "INTERPN_HIP_COLUMN_TAIL", "INTERPN_HIP_COLUMN_KEYS", "INTERPN_HIP_SCATTER_STAGED", "INTERPN_HIP_BIN_SCRAMBLE", "INTERPN_HIP_AXIS_RECORDS", "INTERPN_HIP_BIN_SLICE_LOG2", "INTERPN_HIP_SWEEP", "INTERPN_HIP_SWEEP_PERIOD", "INTERPN_HIP_CUBIC_RECORDS", "INTERPN_HIP_SWEEP_PROBE", "INTERPN_HIP_GATED_ITERS", "INTERPN_HIP_SWEEP_LAYOUT")
LAYOUTS_LIN = [None, "off", "11", "12", "22", "c4"]
LAYOUTS_CUB = [None, "off", "44", "24", "22", "14", "11"]


def run_device(case, rng, dtype, fma=None):
    """The persistent-handle device entry point on torch tensors whose first element sits at a
    random element offset (exercises the aligned-vector and the scalar stream paths)."""
    import torch

    import interpn_amd

    if case.kind == "regular":
        it = interpn_amd.Interpolator.regular(case.method, case.dims, case.starts, case.steps, case.vals,
                                              linearize_extrapolation=case.linearize, fma=fma)
    else:
        it = interpn_amd.Interpolator.rectilinear(case.method, case.grids, case.vals,
                                                  linearize_extrapolation=case.linearize, fma=fma)
    try:
        nobs = case.obs[0].size
        obs_t = []
        # (the sweep kernel takes 16-byte aligned streams only: mostly even element offsets when it is forced)
        sweep = os.environ.get("INTERPN_HIP_SWEEP") in ("1", "2")
        for o in case.obs:
            off = int(rng.choice([0, 2, 0, 2, 1])) if sweep else int(rng.integers(0, 4))
            t = torch.empty(nobs + off, dtype=torch.float64 if dtype == np.float64 else torch.float32, device="cuda")
            t[off:].copy_(torch.from_numpy(np.ascontiguousarray(o)))
            obs_t.append(t[off:])
        off = int(rng.choice([0, 2, 0, 2, 3])) if sweep else int(rng.integers(0, 4))
        out_full = torch.full((nobs + off,), -777.0, dtype=obs_t[0].dtype, device="cuda")
        out_t = out_full[off:]
        it.eval_tensors(obs_t, out_t)
        err, first_bad = None, None
        try:
            it.finish()
        except AssertionError as e:
            err, first_bad = str(e), getattr(e, "first_bad_index", None)
        return out_t.cpu().numpy(), err, first_bad
    finally:
        it.close()


def run(budget: float, seed: int, max_cases: int = 0):
    """Returns (cases, failures)."""
    from interpn_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_cases = n_fail = 0
    saved = {k: os.environ.get(k) for k in KNOBS}
    prev_fma = lib.interpn_hip_set_fma(1)
    try:
        while time.time() < t_end and (max_cases == 0 or n_cases < max_cases):
            method = rng.choice(["linear", "cubic", "nearest"], p=[0.45, 0.4, 0.15])
            kind = rng.choice(["regular", "rectilinear"])
            fp = 4 if method == "cubic" else 2
            nmax = 6 if method == "nearest" else 8
            N = int(rng.integers(1, nmax + 1))
            # keep the grid and the oracle's work bounded
            cap = 2_000_000
            axis = []
            # long axes for N <= 2 (LDS-staged up to 60 KiB of axis image, L1/L2 search beyond)
            axis_cap = 70 if N > 2 or rng.random() < 0.5 else (8000 if N == 1 else 3000)
            for d in range(N):
                hi = max(fp, int(round(cap ** (1.0 / N))))
                axis.append(int(rng.integers(fp, min(hi, axis_cap) + 1)))
            while np.prod(axis) > cap:
                axis[int(np.argmax(axis))] = max(fp, axis[int(np.argmax(axis))] // 2)
            work_per_pt = fp ** N
            nobs = int(min(200_000, max(1, 3_000_000 // work_per_pt)) * rng.uniform(0.05, 1.0)) + int(rng.integers(0, 5))
            dtype = np.float64 if rng.random() < 0.6 else np.float32
            linearize = bool(rng.integers(0, 2))
            extrap = float(rng.choice([0.0, 0.05, 0.3, 2.0]))
            env = {}
            if method == "linear":
                lay = LAYOUTS_LIN[int(rng.integers(0, len(LAYOUTS_LIN)))]
            elif method == "cubic":
                lay = LAYOUTS_CUB[int(rng.integers(0, len(LAYOUTS_CUB)))]
            else:
                lay = None
            if lay: env["INTERPN_HIP_BRICKS"] = lay
            if rng.random() < 0.15: env["INTERPN_HIP_PPL"] = "1"
            if rng.random() < 0.1: env["INTERPN_HIP_FORCE_GENERIC"] = "1"
            if rng.random() < 0.1: env["INTERPN_HIP_GENERIC_RUNTIME"] = "1"
            if rng.random() < 0.2: env["INTERPN_HIP_HOST_CHUNK"] = str(int(rng.integers(1, max(2, nobs))))
            if rng.random() < 0.2: env["INTERPN_HIP_ITERS_PER_BLOCK"] = str(int(rng.choice([1, 2, 3, 8, 64])))
            if rng.random() < 0.4: env["INTERPN_HIP_AXIS_REGS"] = str(int(rng.integers(0, 3)))
            # both forms of the recursive-arm kernel wherever the row-vector form is compiled
            # (k_generic.hip::generic_vec_ok; elsewhere the option falls back to the one-tree form)
            if rng.random() < 0.5: env["INTERPN_HIP_GENERIC_VEC"] = str(int(rng.integers(0, 2)))
            if rng.random() < 0.1: env["INTERPN_HIP_PERSISTENT"] = "1"
            # binned evaluation of the tiled multicubic kernels (device entry point, N = 2..4), dealt or not
            if method == "cubic" and rng.random() < 0.5: env["INTERPN_HIP_BINNED"] = "1"
            if rng.random() < 0.3: env["INTERPN_HIP_DEAL"] = "0"
            # round 3: the LDS-column evaluation of sorted 4-D multicubic points on regular grids (forced where it
            # applies: fully overlapped tiles), its workgroup size / part size, deliberately mis-binned points, short
            # slices; per-bucket search records on or off
            if method == "cubic" and N == 4 and rng.random() < 0.7:  # regular and (round 4) rectilinear grids
                env["INTERPN_HIP_BINNED"] = "1"
                env["INTERPN_HIP_BRICKS"] = "11"
                env["INTERPN_HIP_COLUMN"] = str(int(rng.choice([-1, 1, 1, 1, 0])))
                env["INTERPN_HIP_COLUMN_THREADS"] = str(int(rng.choice([256, 384, 768, 768])))
                env["INTERPN_HIP_COLUMN_GROUPS"] = str(int(rng.integers(1, 3)))
                if rng.random() < 0.6: env["INTERPN_HIP_COLUMN_CPP"] = str(int(rng.choice([1, 2, 3, 5])))  # several K-range phases on small grids
                if rng.random() < 0.5: env["INTERPN_HIP_COLUMN_PART"] = str(int(rng.choice([1, 64, 700, 2048, 12288])))
                if rng.random() < 0.4: env["INTERPN_HIP_BIN_SCRAMBLE"] = "1"
                if rng.random() < 0.4: env["INTERPN_HIP_SCATTER_STAGED"] = "0"
                if rng.random() < 0.3: env["INTERPN_HIP_BIN_SLICE_LOG2"] = "16"
                # round 4, second session: dim 0 from per-part Hermite coefficients (default) or from the values, LDS tiles
                # padded / bare, the bins' tail cut
                if rng.random() < 0.25: env["INTERPN_HIP_COLUMN_COEF"] = "0"
                if rng.random() < 0.5: env["INTERPN_HIP_COLUMN_PAD"] = str(int(rng.integers(0, 2)))
                if rng.random() < 0.4: env["INTERPN_HIP_COLUMN_TAIL"] = str(int(rng.choice([0, 0x22, 0x54, 0x1f])))
                if rng.random() < 0.4: env["INTERPN_HIP_COLUMN_KEYS"] = "0"
            # round 5: the sweep evaluation of 3-D f64 multilinear batches (device entry point), forced on batches of
            # any size, with the measured period, a fixed one, or no clock
            if method == "linear" and N == 3 and rng.random() < 0.7:  # f64 and (second session) f32
                env["INTERPN_HIP_SWEEP"] = "1"
                env["INTERPN_HIP_SWEEP_PERIOD"] = str(int(rng.choice([0, 0, 1, 300, 2500])))
                env.pop("INTERPN_HIP_FORCE_GENERIC", None)
                if rng.random() < 0.4: env["INTERPN_HIP_SWEEP_LAYOUT"] = str(rng.choice(["11", "12"]))  # (round 6: either table, whatever the grid's size)
            if kind == "rectilinear" and rng.random() < 0.3: env["INTERPN_HIP_AXIS_RECORDS"] = "0"
            # round 5, last session: rectilinear multicubic with / without the per-cell records of the axes (or a bound they
            # exceed), and 2-D / 3-D multicubic through the sweep kernel
            if method == "cubic" and kind == "rectilinear" and rng.random() < 0.4: env["INTERPN_HIP_CUBIC_RECORDS"] = str(int(rng.choice([0, 0, 1, 64])))
            if method == "cubic" and N in (2, 3) and rng.random() < 0.3:
                env["INTERPN_HIP_SWEEP"] = "1"
                env["INTERPN_HIP_SWEEP_PERIOD"] = str(int(rng.choice([0, 1, 1500])))
                env["INTERPN_HIP_BRICKS"] = "11"
                env.pop("INTERPN_HIP_FORCE_GENERIC", None)
                env.pop("INTERPN_HIP_BINNED", None)
            # round 6: the device-side sample in front of the sweep kernels and the gated pair of launches behind it, on batches
            # of any size from 16384 points (option sweep = 2), unordered or clustered in one cell (the sample's two verdicts)
            clustered = False
            if ((method == "linear" and N in (2, 3)) or (method == "nearest" and N in (2, 3) and kind == "regular") or (method == "cubic" and N == 2)) \
                    and nobs >= 16384 and rng.random() < 0.5:
                env["INTERPN_HIP_SWEEP"] = "2"
                env["INTERPN_HIP_SWEEP_PROBE"] = str(int(rng.choice([1, 2])))
                env["INTERPN_HIP_GATED_ITERS"] = str(int(rng.choice([0, 1, 3, 16, 64])))
                env.pop("INTERPN_HIP_FORCE_GENERIC", None)
                env.pop("INTERPN_HIP_BINNED", None)
                if method == "cubic": env["INTERPN_HIP_BRICKS"] = "11"
                clustered = bool(rng.random() < 0.5)
            # per-bucket records for 1-D multilinear-rectilinear, also on axes short enough for LDS
            if method == "linear" and kind == "rectilinear" and N == 1 and rng.random() < 0.5: env["INTERPN_HIP_BRICKS"] = "on"
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            case = synthetic_case(method, kind, N, axis, nobs, int(rng.integers(0, 2**31)), dtype, linearize=linearize,
                                  extrap=extrap, specials=bool(rng.integers(0, 2)))
            if method == "cubic" and rng.random() < 0.3:
                # grid values the rectilinear node's short divisions must hand back: a lattice (differences that are exactly +0),
                # blocks of tiny / huge magnitude, zeros of both signs
                v = case.vals.astype(np.float64)
                mode = int(rng.integers(0, 3))
                if mode == 0:
                    v = np.round(v * 4) / 4 + 0.0
                elif mode == 1:
                    k = int(rng.integers(0, v.size))
                    v[k:k + v.size // 7 + 1] *= (1e-300 if dtype == np.float64 else 1e-36) if rng.random() < 0.5 else (1e200 if dtype == np.float64 else 1e30)
                else:
                    k = int(rng.integers(0, v.size))
                    z = np.zeros(min(v.size - k, v.size // 5 + 1))
                    z[rng.random(z.size) < 0.5] = -0.0
                    v[k:k + z.size] = z
                with np.errstate(all="ignore"):
                    case.vals = v.astype(dtype)
            if clustered:  # every point inside one cell (+ a few strays): neighbours share table lines, the one-pass kernel takes the batch
                for d in range(N):
                    g = case.grids[d].astype(np.float64)
                    c = int(rng.integers(0, g.size - 1))
                    keep = case.obs[d][::997].copy()
                    case.obs[d][:] = rng.uniform(g[c], g[c + 1], nobs).astype(dtype)
                    case.obs[d][::997] = keep
            if kind == "rectilinear" and rng.random() < 0.3:
                # inject NaN / inf / huge coordinates: rectilinear never errors, results must still match
                for _ in range(3):
                    case.obs[int(rng.integers(0, N))][int(rng.integers(0, nobs))] = rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30])
            fma = bool(rng.random() < 0.7)  # the reference's `fma` cargo feature, both flavours
            lib.interpn_hip_set_fma(1 if fma else 0)
            inject = kind == "regular" and rng.random() < 0.25
            if inject:
                # regular grids abort at the first unrepresentable coordinate: same error, same
                # prefix written, rest of `out` untouched (host entry points)
                for _ in range(int(rng.integers(1, 4))):
                    case.obs[int(rng.integers(0, N))][int(rng.integers(0, nobs))] = rng.choice([np.nan, np.inf, -np.inf])
            want = np.full(nobs, -777.0, dtype=dtype)
            got = np.full(nobs, -777.0, dtype=dtype)
            err_o = err_g = None
            try:
                run_oracle(pyoracle, case, fma, out=want)
            except AssertionError as e:
                err_o = str(e)
            device_path = rng.random() < 0.3 or "INTERPN_HIP_COLUMN" in env or "INTERPN_HIP_SWEEP" in env
            # per-handle flavour (round 3): on the device path half of the cases pass the flavour to the handle while the
            # process default says the opposite
            fma_arg = None
            if device_path and rng.random() < 0.5:
                fma_arg = fma
                lib.interpn_hip_set_fma(0 if fma else 1)
            try:
                if device_path:
                    got, err_g, first_bad = run_device(case, rng, dtype, fma_arg)
                else:
                    run_hip_raw(case, out=got)
            except AssertionError as e:
                err_g = str(e)
            except Exception as e:  # noqa: BLE001
                err_g = "EXC " + repr(e)
            if device_path and err_g is not None and err_o == err_g:
                # device entry point: out[first_bad..] is unspecified; the index and the prefix are not
                k = int(np.flatnonzero(want == dtype(-777.0))[0]) if np.any(want == dtype(-777.0)) else nobs
                same = first_bad == k and np.array_equal(got[:k], want[:k], equal_nan=True)
            else:
                same = err_o == err_g and np.array_equal(got, want, equal_nan=True)
            if inject and err_o is None:
                same = False  # the injection must have been seen
            n_cases += 1
            if not same:
                n_fail += 1
                nbad = int(np.sum(~((got == want) | (np.isnan(got) & np.isnan(want)))))
                if os.environ.get("FUZZ_DUMP"):
                    ib = np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))[:4]
                    print("   DUMP vals abs min/max", float(np.nanmin(np.abs(case.vals))), float(np.nanmax(np.abs(case.vals))), "zeros", int((case.vals == 0).sum()),
                          "bad:", [(int(i), float(got[i]), float(want[i]), [float(o[i]) for o in case.obs]) for i in ib], flush=True)
                print(f"FAIL method={method} kind={kind} N={N} axis={axis} nobs={nobs} dtype={np.dtype(dtype).name} "
                      f"linearize={linearize} fma={fma} device_path={device_path} extrap={extrap} env={env} nbad={nbad} err_oracle={err_o!r} err_hip={err_g!r}", flush=True)
    finally:
        lib.interpn_hip_set_fma(prev_fma)
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return n_cases, n_fail


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_cases, n_fail = run(budget, seed)
    print(f"fuzz: {n_cases} cases, {n_fail} failures, seed {seed}, {budget:.0f} s")
    sys.exit(1 if n_fail else 0)
