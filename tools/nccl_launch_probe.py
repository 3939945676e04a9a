"""Why does the K-step region of bench.py run slower once an nccl process group exists?
Times, per step, the host side of `eval_tensors` (launch) and `finish` (status round trip) and the
HIP-event kernel time, before and after `init_process_group("nccl")`, and after a barrier.

    python tools/nccl_launch_probe.py [gloo|nccl]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import interpn_amd  # noqa: E402


def loop(tag, it, obs, out, steps=40):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    tl, tf = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        a = time.perf_counter()
        it.eval_tensors(obs, out)
        b = time.perf_counter()
        ev[k][1].record()
        c = time.perf_counter()
        it.finish()
        d = time.perf_counter()
        tl.append(b - a)
        tf.append(d - c)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    km = [x.elapsed_time(y) for x, y in ev]
    print(f"{tag:32s} step {el / steps * 1e3:.4f} ms  event-kernel {np.mean(km):.4f} (min {np.min(km):.4f} max {np.max(km):.4f})  "
          f"launch-call {np.mean(tl) * 1e6:.1f} us  finish-call {np.mean(tf) * 1e3:.4f} ms  affinity {len(os.sched_getaffinity(0))} cpus",
          flush=True)


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    n, P = 64, 100_000_000
    g = np.linspace(-1, 1, n)
    rng = np.random.default_rng(1)
    vals = rng.uniform(-1, 1, n**3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
    out = torch.empty(P, dtype=torch.float64, device=dev)
    it = interpn_amd.Interpolator.regular("linear", [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals, False, 0, np.float64)
    for _ in range(100):
        it.eval_tensors(obs, out)
    it.finish()
    loop("no process group", it, obs, out)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    loop(f"{backend} group initialised", it, obs, out)
    t = torch.zeros(4, device=dev if backend == "nccl" else "cpu")
    dist.broadcast(t, src=0)
    torch.cuda.synchronize()
    loop("after one broadcast", it, obs, out)
    dist.barrier()
    torch.cuda.synchronize()
    loop("after barrier", it, obs, out)
    loop("again", it, obs, out)
    for env in ("TORCH_NCCL_BLOCKING_WAIT", "NCCL_LAUNCH_MODE"):
        print(env, os.environ.get(env))
    dist.destroy_process_group()
    loop("group destroyed", it, obs, out)
    it.close()


if __name__ == "__main__":
    main()
