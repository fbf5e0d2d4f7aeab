#!/usr/bin/env python3
"""bench.py -- images/sec of one CU-Net train step on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is the reference's training iteration (cu-net.py:171-183): forward of
create_cu_net(4, 32, 128, K, L, order, loss_num), sum of per-head pixelwise MSE, backward, RMSprop
update -- plus, for N > 1, the per-bucket RCCL gradient all-reduce overlapped with backward.
Workload at N=1 is BASELINE.json configs[1]: L=2, order=1, loss_num=2, bs=24, 256x256, 68 landmarks,
fp32, synthetic MPII/300-W-shaped input (U[0,1) images, one 7x7 Gaussian blob per landmark), random
reference-scheme init.  Weak scaling: every rank processes its own 24 images.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel class (chosen from a profiled
warm-up step): achieved = algorithmic FLOPs of its launches / their HIP-event time measured inside
the timed region on the launch stream.  `cpu_baseline` times the CPU oracle (oracle/cunet_ref.py, a
restatement of the reference pinned bit-exact to it) on this host for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0
# algorithmic train-step GFLOP per image, conv MACs x2 only, no credit for recompute
TRAIN_GFLOP_PER_IMG = {(2, 68): 16.252, (8, 68): 65.731, (8, 16): 64.422, (16, 16): 129.086}
FWD_GFLOP_PER_IMG = {(2, 68): 5.623, (8, 68): 22.116, (8, 16): 21.680, (16, 16): 43.234}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def synthetic_batch(n, class_num, hw, seed, device):
    """x ~ U[0,1); target = one 7x7 blob exp(-(dx^2+dy^2)/9) per landmark (pylib/HumanPts.py:49-76, sigma=1)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, hw, hw, generator=g)
    res = hw // 4
    cx = torch.randint(3, res - 3, (n, class_num), generator=g)
    cy = torch.randint(3, res - 3, (n, class_num), generator=g)
    ax = torch.arange(7, dtype=torch.float32) - 3
    blob = torch.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / 9.0)
    t = torch.zeros(n, class_num, res, res)
    for a in range(n):
        for k in range(class_num):
            yy, xx = int(cy[a, k]), int(cx[a, k])
            t[a, k, yy - 3:yy + 4, xx - 3:xx + 4] = blob
    return x.to(device), t.to(device)


def cpu_baseline(layers, class_num, steps):
    """CPU oracle (kind 'port': restatement of the reference, bit-exact to it on the golden vectors)
    timed on this host: CU-Net-L order 1, bs=4, 256x256 (BASELINE.json configs[0]), full train step."""
    from oracle import cunet_ref as O
    cores = torch.get_num_threads()
    spec = O.Spec(4, 32, 128, class_num, layers, 1, layers)
    st = O.init_state(spec, seed=2)
    x, t = O.synthetic_batch(4, class_num, 256, seed=0)
    opt = {}
    O.train_step(spec, st, x, t, opt)            # warm-up (first step is several times slower)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(spec, st, x, t, opt)
    dt = time.perf_counter() - t0
    return {'value': round(4 * steps / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'CU-Net-{layers} order 1 K={class_num}, bs=4, 256x256 fp32, {steps} full train steps '
                      f'(fwd+MSE+bwd+RMSprop) after 1 warm-up, torch CPU {torch.__version__}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--class-num', type=int, default=68)
    ap.add_argument('--bs', type=int, default=24, help='per-GPU batch')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-out', default='')
    ap.add_argument('--forward-only', action='store_true', help='inference: eval-mode forward (running statistics), no loss / backward')
    ap.add_argument('--bf16', action='store_true', help='bf16 activation storage + bf16 MFMA forward (train step: gradients, weights, optimiser stay fp32; '
                    'with --forward-only: bf16-storage inference)')
    ap.add_argument('--bf16-grads', action='store_true', help='with --bf16: the gradient tensors of backward (dY, dz, dX) stored as bf16 too')
    ap.add_argument('--bits-w', type=int, default=0, help='>0: quantised-weight train step (QuanOp, utils/quantize.py), e.g. 1 = BASELINE config 5')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pg = None
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run even one rank goes through RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD

    import cu_net_amd
    from cu_net_amd.trainer import FusedTrainer

    L, K, bs = args.layers, args.class_num, args.bs
    torch.manual_seed(2)
    net = cu_net_amd.create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=K,
                                   layer_num=L, order=1, loss_num=L).to(dev)
    net.train()
    quan = None
    if args.bits_w > 0:
        from cu_net_amd.quant import QuanOp
        quan = QuanOp(net, bits_w=args.bits_w, bits_i=8, bits_g=8)
    tr = FusedTrainer(net, lr=2.5e-4, alpha=0.99, eps=1e-8, process_group=pg, quan_op=quan, bf16=args.bf16 or args.bf16_grads, bf16_grads=args.bf16_grads)
    tr.broadcast_parameters(0)
    x, t = synthetic_batch(bs, K, 256, seed=1000 + rank, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.forward_only:
        net.eval()

        def one_step():
            with torch.no_grad():
                outs = net.forward_bf16(x) if args.bf16 else net(x)      # the public path: loss_num NCHW heat maps
            return outs[-1][0, 0, 0, 0]
    else:
        def one_step():
            return tr.step(x, t)
    args.bf16 = args.bf16 or args.bf16_grads
    plan = net._get_plan(bs, 256, 256, not args.forward_only, bf16=args.bf16)
    # ---- warm-up; one of the warm-up steps is profiled per kernel class to pick the dominant one
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1:
            plan.handle.profile_reset()
            plan.handle.profile_begin(1)
        loss = one_step()
    torch.cuda.synchronize(dev)
    prof_all = plan.handle.profile_collect()
    plan.handle.profile_begin(0)
    have_classes = any(v[0] for v in prof_all.values())       # the bf16 path is not instrumented per class
    dominant = max(prof_all.items(), key=lambda kv: kv[1][1])[0]
    if rank == 0:
        tot = sum(v[1] for v in prof_all.values())
        log(f'per-class profile of one warm-up step (sum of kernel times {tot:.3f} ms):')
        lines = []
        for name, (cnt, ms, fl, by) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
            if cnt:
                lines.append(f'  {name:22s} launches={cnt:4d} ms={ms:8.3f} ({100 * ms / tot:5.1f}%) '
                             f'TFLOP/s={fl / ms / 1e9 if ms else 0:7.2f} GB/s={by / ms / 1e6 if ms else 0:8.1f}')
        log('\n'.join(lines))
        if args.profile_out:
            with open(args.profile_out, 'w') as f:
                f.write('\n'.join(lines) + '\n')

    # ---- timed region: exactly K steps, events only around the dominant class
    plan.handle.profile_reset()
    plan.handle.profile_begin(2, plan.handle.profile_class_index(dominant))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    prof = plan.handle.profile_collect()
    plan.handle.profile_begin(0)
    if pg is not None:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss)

    if rank == 0:
        imgs = world * bs * args.steps
        value = imgs / dt
        cnt, ms, fl, by = prof[dominant]
        mfma_bound = fl > 0
        roof = None
        if not have_classes or ms <= 0:
            pass
        elif mfma_bound:
            achieved = fl / (ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': dominant, 'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': None,
                    'launches': cnt, 'avg_launch_us': round(1e3 * ms / max(cnt, 1), 2)}
        else:
            achieved = by / (ms * 1e-3) / 1e9
            roof = {'bound': 'hbm', 'kernel': dominant, 'achieved': round(achieved, 1), 'peak': PEAK_HBM_GBS,
                    'unit': 'GB/s', 'frac': round(achieved / PEAK_HBM_GBS, 4), 'traffic': None,
                    'launches': cnt, 'avg_launch_us': round(1e3 * ms / max(cnt, 1), 2)}
        if not have_classes and args.forward_only:      # bf16 inference: whole-forward algorithmic bytes (bf16 forward is HBM-bound) against HBM peak
            fb = {(2, 68): 55.2e6, (8, 68): 221.7e6, (8, 16): 218.3e6, (16, 16): 437.0e6}.get((L, K))
            ach = (fb * value / 1e9) if fb else 0.0
            roof = {'bound': 'hbm', 'kernel': 'whole forward (conv inputs + outputs once, bf16)', 'achieved': round(ach, 1),
                    'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': None}
        # HBM bytes per launch of that kernel class from the committed PMC passes (rocprofv3 --pmc cannot run inside
        # this process): profiles/r01_traffic.json, produced by tools/profile_round.sh + tools/pmc_traffic.py on
        # this workload.  null when the file has no entry for the class or the workload is not the profiled one.
        try:
            if (L, K, bs) == (2, 68, 24) and have_classes:
                tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_traffic.json')))
                ent = tj['classes'].get(dominant)
                if ent:
                    roof['traffic'] = ent['hbm_bytes_per_launch']
                    roof['traffic_unit'] = 'bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_traffic.json)'
                    roof['algorithmic_bytes_per_launch'] = round(by / max(cnt, 1))
        except Exception:
            pass
        out = {
            'metric': ('images/sec inference forward' if args.forward_only else 'images/sec train step')
                      + ', 256x256x3 -> 64x64xK heatmaps, CU-Net-%d' % L,
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': f'CU-Net layer_num={L} order=1 loss_num={L}, bs={bs}/GPU, 256x256, {K} landmarks, '
                                   + (f'QuanOp bits_w={args.bits_w} bits_g=8 (quantise -> step -> restore -> grad rewrite), ' if args.bits_w > 0 else '')
                                   + (('bf16-storage' if args.bf16 else 'fp32') + ' eval-mode forward only' if args.forward_only else
                                      ('bf16 activation + gradient tensors' if args.bf16_grads else 'bf16-activation' if args.bf16 else 'fp32') + ' train step (fwd + MSE + bwd + RMSprop')
                                   + ('' if args.forward_only else (' + RCCL bucketed grad all-reduce)' if world > 1 else ')')),
                       'global_batch': world * bs, 'parallelism': f'dp{world}'},
            'roofline': roof,
            'final_loss': final_loss,
        }
        g = (FWD_GFLOP_PER_IMG if args.forward_only else TRAIN_GFLOP_PER_IMG).get((L, K))
        if g:
            out['step_tflops'] = round(g * value / 1e3, 2)
            if args.bf16:
                out['step_frac_of_bf16_mfma_peak'] = round(g * value / 1e3 / (2500.0 * world), 4)      # dense bf16 MFMA ~2.5 PFLOP/s
            else:
                out['step_frac_of_f32_mfma_peak'] = round(g * value / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 4)
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            out['cpu_baseline'] = cpu_baseline(L, K, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
