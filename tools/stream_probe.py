#!/usr/bin/env python3
"""Does the stream a step is enqueued on change its GPU time?  (tools only)

torch's default stream is HIP's legacy NULL stream (handle 0).  This probe times (a) a chain of N trivial dependent kernels and (b) the
bench workload's train step, on the NULL stream and on an ordinary stream created by torch (torch.cuda.Stream()), in one process.
    python tools/stream_probe.py [--layers 2] [--bf16-grads]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cu_net_amd  # noqa: E402
from cu_net_amd.trainer import FusedTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--layers', type=int, default=2)
ap.add_argument('--bf16-grads', action='store_true')
ap.add_argument('--steps', type=int, default=40)
a = ap.parse_args()
dev = torch.device('cuda', 0)


def chain(stream, n=400):
    z = torch.zeros(64, device=dev)
    with torch.cuda.stream(stream):
        for _ in range(20):
            z.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            z.add_(1.0)
        e1.record()
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


null = torch.cuda.default_stream(dev)
own = torch.cuda.Stream(device=dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
for name, s in (('NULL stream', null), ('torch.cuda.Stream()', own), ('torch.cuda.Stream(priority=-1)', hi), ('NULL stream again', null)):
    print(f'chain of 400 trivial dependent kernels on {name:32s}: {chain(s):6.2f} us per kernel', flush=True)

torch.manual_seed(2)
K = 68
net = cu_net_amd.create_cu_net(4, 32, 128, K, a.layers, 1, a.layers).to(dev).train()
tr = FusedTrainer(net, bf16=a.bf16_grads, bf16_grads=a.bf16_grads)
x, t = bench.synthetic_batch(24, K, 256, seed=1000, device=dev)


def steps(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(5):
            tr.step(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = tr.step(x, t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return 24 * n / dt, 1e3 * dt / n, float(loss)


for rep in range(2):
    for name, s in (('NULL stream', null), ('torch.cuda.Stream()', own), ('torch.cuda.Stream(priority=-1)', hi)):
        v, ms, loss = steps(s, a.steps)
        print(f'CU-Net-{a.layers} {"bf16_grads" if a.bf16_grads else "fp32"} train step on {name:32s}: {v:8.1f} img/s {ms:7.3f} ms  (loss {loss:.5f})', flush=True)
