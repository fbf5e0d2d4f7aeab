"""One training step captured into a HIP graph (torch.cuda.CUDAGraph around FusedTrainer.step) against the same step launched
eagerly (tools only).  The capture goes through the C-ABI unchanged: the library's launches, memsets and its fork / join of the
internal side stream land in the graph.

    python tools/graph_probe.py [--layers L] [--class-num K] [--bf16-grads] [--steps N]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--class-num', type=int, default=68)
    ap.add_argument('--bs', type=int, default=24)
    ap.add_argument('--steps', type=int, default=40)
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
    eager = timed(lambda: tr.step(x, t), a.steps)
    out = {'layers': a.layers, 'bf16_grads': a.bf16_grads, 'eager_ms_per_step': round(eager, 3)}
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = tr.step(x, t)
        for _ in range(3):
            g.replay()
        out['graph_ms_per_step'] = round(timed(g.replay, a.steps), 3)
        out['loss_after_replays'] = float(loss)
        out['eager_again_ms_per_step'] = round(timed(lambda: tr.step(x, t), a.steps), 3)
    except Exception as e:      # capture is an experiment here: report, do not fail
        out['graph_error'] = repr(e)[:400]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
