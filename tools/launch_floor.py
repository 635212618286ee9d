#!/usr/bin/env python3
"""What a chain of N dependent, near-empty kernels costs on this box, eager and as one HIP graph -- the floor under the one-utterance forward
(c1: 85 dependent launches; VERDICT r05 item 8: "find out why the captured graph (1.10 ms) loses to eager (1.07 ms)").
  python tools/launch_floor.py [N = 85]"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 85
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)


def chain():
    for _ in range(N):
        x.add_(1.0)          # one 64-thread kernel, dependent on the previous one through x


def timed(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, 1e3 * (time.perf_counter() - t0) / reps


with torch.no_grad():
    g_ms, g_wall = None, None
    e_ms, e_wall = timed(chain)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        chain()
    g_ms, g_wall = timed(graph.replay)
    # one chain at a time, the host waiting for each (the latency a single request sees)
    lat = []
    for fn in (chain, graph.replay):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
            torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t0) / 100)
print("%d dependent 64-thread kernels: back to back, GPU time per chain  eager %.3f ms (%.1f us per launch; host %.3f ms)  |  one HIP graph %.3f ms (%.1f us per node; host %.3f ms)"
      % (N, e_ms, 1e3 * e_ms / N, e_wall, g_ms, 1e3 * g_ms / N, g_wall))
print("one chain at a time, host waits for each: eager %.3f ms, graph %.3f ms" % (lat[0], lat[1]))
