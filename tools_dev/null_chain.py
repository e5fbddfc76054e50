#!/usr/bin/env python3
"""What a dependent kernel costs on this box whatever it does: N tiny elementwise kernels back to back on one stream (enqueued far
ahead of the GPU, so the stream is GPU-bound), wall time per kernel.  The floor under the 8-launch chain of a small-corpus search."""
import time, torch
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
for n in (2000, 20000):
    for _ in range(200): x.add_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} one-wave kernels on one stream: {(t2 - t0) / n * 1e6:.2f} us per kernel (host enqueue alone {(t1 - t0) / n * 1e6:.2f} us)", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(8): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize()
    print(f"hipGraph of 8 such kernels, 500 replays: {(time.perf_counter() - t0) / 500 * 1e6:.2f} us per replay = {(time.perf_counter() - t0) / 4000 * 1e6:.2f} us per kernel", flush=True)
