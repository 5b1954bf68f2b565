#!/usr/bin/env python3
"""What a HIP graph saves per kernel boundary on this box: N back-to-back launches of a tiny kernel and of a ~100 us kernel, eager (one stream) against a
captured graph replay; microseconds per launch."""
import time

import torch

dev = "cuda:0"
small = torch.zeros(4096, device=dev)
big = torch.zeros(64 << 20, device=dev)      # 256 MB: add_ ~ 100 us


def run(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n


def graphed(fn, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / (5 * n)


for name, t, n in (("tiny kernel (4096 floats)", small, 400), ("~100 us kernel (256 MB add_)", big, 60)):
    f = lambda: t.add_(1.0)  # noqa: E731
    run(f, 20)
    print(f"{name}: eager {run(f, n):.2f} us per launch, graph replay {graphed(f, n):.2f} us per launch", flush=True)
