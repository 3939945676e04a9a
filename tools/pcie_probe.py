#!/usr/bin/env python3
"""PCIe probe: pageable / pinned H2D and D2H rates, alone and concurrent (threads)."""
import threading, time
import numpy as np, torch

N = 20_000_000
dev = torch.device("cuda:0")
host = [torch.from_numpy(np.random.default_rng(i).uniform(-1, 1, N)) for i in range(4)]
pinned = [h.clone().pin_memory() for h in host]
devb = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]

def h2d(i, src):
    with torch.cuda.stream(streams[i]):
        devb[i].copy_(src[i], non_blocking=True)
        streams[i].synchronize()

def d2h(i, dst):
    with torch.cuda.stream(streams[i]):
        dst[i].copy_(devb[i], non_blocking=True)
        streams[i].synchronize()

def timed(label, jobs, nbytes):
    best = 1e9
    for _ in range(4):
        ths = [threading.Thread(target=f, args=a) for f, a in jobs]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{label:50s} {nbytes/best/1e9:7.1f} GB/s", flush=True)

B = N * 8
timed("H2D pageable x1", [(h2d, (0, host))], B)
timed("H2D pageable x2 threads", [(h2d, (0, host)), (h2d, (1, host))], 2 * B)
timed("H2D pageable x3 threads", [(h2d, (i, host)) for i in range(3)], 3 * B)
timed("H2D pinned x1", [(h2d, (0, pinned))], B)
timed("H2D pinned x3 threads", [(h2d, (i, pinned)) for i in range(3)], 3 * B)
timed("D2H pageable x1", [(d2h, (0, host))], B)
timed("D2H pinned x1", [(d2h, (0, pinned))], B)
timed("H2D x3 + D2H x1 pageable (4 threads)", [(h2d, (i, host)) for i in range(3)] + [(d2h, (3, host))], 4 * B)
timed("H2D x3 + D2H x1 pinned (4 threads)", [(h2d, (i, pinned)) for i in range(3)] + [(d2h, (3, pinned))], 4 * B)
t0 = time.perf_counter(); x = host[0].clone().pin_memory(); print(f"pin_memory of 160 MB (alloc+copy): {(time.perf_counter()-t0)*1e3:.1f} ms")
import ctypes
rt = ctypes.CDLL(None)
