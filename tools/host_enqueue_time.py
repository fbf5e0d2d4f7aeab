"""Host-side enqueue time of one training step against its GPU time (tools only).

    python tools/host_enqueue_time.py [--layers L] [--class-num K] [--bf16-grads] [--steps N]

The host enqueues a step (4 C-ABI calls, a few hundred kernel launches, the side-stream hand-overs) without waiting for the
GPU; when enqueueing a step takes nearly as long as executing it, the short launches at the bottom of every U are host-bound
and the remedy is a captured graph, not a faster kernel.  Two measurements: steps enqueued one at a time from an idle GPU (the
host's own cost), and back to back (where the runtime's bound on outstanding launches throttles a host that runs ahead, so the
figure converges to the GPU's step time and says nothing about the host).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--class-num', type=int, default=68)
    ap.add_argument('--bs', type=int, default=24)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--bf16-grads', action='store_true')
    a = ap.parse_args()
    import cu_net_amd
    from cu_net_amd.trainer import FusedTrainer
    from bench import synthetic_batch
    dev = torch.device('cuda:0')
    net = cu_net_amd.create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=a.class_num, layer_num=a.layers, order=1,
                                   loss_num=a.layers).to(dev).train()
    tr = FusedTrainer(net, lr=2.5e-4, alpha=0.99, eps=1e-8, bf16=a.bf16_grads, bf16_grads=a.bf16_grads)
    x, t = synthetic_batch(a.bs, a.class_num, 256, seed=1000, device=dev)
    for _ in range(5):
        tr.step(x, t)
    torch.cuda.synchronize()
    # (a) one step at a time from an idle GPU: what enqueueing a step costs the host when nothing throttles it
    iso = []
    for _ in range(a.steps):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        tr.step(x, t)
        iso.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    iso.sort()
    # (b) back to back: the host runs ahead until the runtime's bound on outstanding launches stops it
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h0 = time.perf_counter()
        tr.step(x, t)
        host += time.perf_counter() - h0
    enq_done = time.perf_counter()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(json.dumps({'layers': a.layers, 'bf16_grads': a.bf16_grads, 'steps': a.steps,
                      'host_enqueue_ms_isolated_step_median': round(1e3 * iso[len(iso) // 2], 3), 'host_enqueue_ms_isolated_step_min': round(1e3 * iso[0], 3),
                      'host_enqueue_ms_per_step_back_to_back': round(1e3 * host / a.steps, 3), 'wall_ms_per_step': round(1e3 * wall / a.steps, 3),
                      'host_lead_at_end_ms': round(1e3 * (wall - (enq_done - t0)), 3)}))


if __name__ == '__main__':
    main()
