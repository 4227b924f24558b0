#!/usr/bin/env python
"""What does a fork cost at hipGraphLaunch?  DESIGN.md 3.6 measured 11.2 ms of host time per launch for a captured training step whose
backward forks EVERY layer's weight gradient onto a side stream (22 forks, 22 joins) against 0.3 ms for the linear chain.  This probe
captures N small kernels as a linear chain and with K fork/join pairs (K in 0, 1, 2, 4, 8, 22; the side branch carries N/10 of the kernels)
and prints the host time of graph.replay() with the GPU idle and the GPU time of a replay."""
import json
import sys
import time

import torch

dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 270
x = torch.zeros(1 << 16, device=dev)
y = torch.zeros(1 << 16, device=dev)
main = torch.cuda.Stream()
side = torch.cuda.Stream()


def chain(K):
    per = max(1, N // (K + 1)) if K else N
    done = 0
    for seg in range(K + 1):
        if seg > 0:
            side.wait_stream(torch.cuda.current_stream())          # fork: the side branch starts behind what main has issued so far
            with torch.cuda.stream(side):
                for _ in range(max(1, N // 10 // max(K, 1))):
                    y.add_(1.0)
        for _ in range(per if seg < K else N - done):
            x.add_(1.0)
        done += per
    if K:
        torch.cuda.current_stream().wait_stream(side)               # one join at the end (earlier side work chains on the side stream)


for K in (0, 1, 2, 4, 8, 22):
    with torch.cuda.stream(main):
        chain(K)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            chain(K)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        host = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            host.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        e1.synchronize()
        print(json.dumps({'kernels': N, 'forks': K, 'host_ms_per_launch_median': round(sorted(host)[5], 3), 'host_ms_min': round(min(host), 3), 'gpu_ms_per_replay': round(e0.elapsed_time(e1) / 5, 3)}))
