#!/usr/bin/env python3
"""bench.py -- images/sec of one CU-Net train step on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher: bench.py starts the N ranks itself through torch.distributed.run)

A "step" is the reference's training iteration (cu-net.py:171-183): forward of
create_cu_net(4, 32, 128, K, L, order, loss_num), sum of per-head pixelwise MSE, backward, RMSprop
update -- plus, for N > 1, the per-bucket RCCL gradient all-reduce overlapped with backward.
Workload at N=1 is BASELINE.json configs[1]: L=2, order=1, loss_num=2, bs=24, 256x256, 68 landmarks,
fp32, synthetic MPII/300-W-shaped input (U[0,1) images, one 7x7 Gaussian blob per landmark), random
reference-scheme init.  Weak scaling: every rank processes its own 24 images.

Rank 0 prints ONE JSON line.  `value` / `ms_per_step` are the wall clock of exactly K steps between barriers
(the driver's contract); `ms_per_step_median` is the median of the K per-step times taken from events on the
stream.  `roofline` describes the dominant kernel class of the caller's stream (the critical path; chosen from a
profiled warm-up step; a larger side-stream class, if any, is named in `roofline.largest_side_stream_class`):
achieved = algorithmic FLOPs of its launches / their HIP-event time measured inside the timed region on the
launch stream.  `cpu_baseline` times the CPU oracle (oracle/cunet_ref.py, a restatement of the reference pinned
bit-exact to it) on this host for a bounded sample.  At N=1 the same run also times the other single-GPU
headline configurations of BASELINE.json -- config 3 (CU-Net-8, bf16 storage), config 5 (CU-Net-16, binary weights; MFMA and
AND-popcount forward) -- and the forward-only (inference) rates, attached as `also: [...]` (BASELINE's metric names "CU-Net-2 and
CU-Net-8"); at N > 1 the `also` entries are BASELINE config 4 (CU-Net-8, K = 16, 24 images per rank, RCCL all-reduce).
It measures the SHIPPED library only (CUNET_LIB_PATH is refused; the loaded path is printed and reported as `library_path`).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (which launches ran on the library's internal low-priority side stream -- weight gradients, heat-map heads of a training pass,
# skip adapters without a pair kernel -- is recorded per launch by the library: cunet_profile_get_stream)
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
# algorithmic GFLOP per image, conv MACs x2 only, no credit for recompute
TRAIN_GFLOP_PER_IMG = {(2, 68): 16.252, (8, 68): 65.731, (8, 16): 64.422, (16, 16): 129.086}
FWD_GFLOP_PER_IMG = {(2, 68): 5.623, (8, 68): 22.116, (8, 16): 21.680, (16, 16): 43.234}
# algorithmic forward bytes per image at perfect per-node fusion, fp32 (bf16: half); train step = 3x
FWD_MB_PER_IMG_F32 = {(2, 68): 110.4, (8, 68): 443.5, (8, 16): 436.7, (16, 16): 873.9}


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


def host_cpu():
    """(model string, physical cores) of this host from /proc/cpuinfo; physical = distinct (package, core id) pairs."""
    model, cores, phys, core = 'unknown', set(), None, None
    try:
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name' and model == 'unknown':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif k == '' and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    return model, n


def cpu_baseline(layers, class_num, steps):
    """CPU oracle (kind 'port': restatement of the reference, bit-exact to it on the golden vectors) timed on this host:
    CU-Net-L order 1, bs=4, 256x256 (BASELINE.json configs[0]), full train step.  A short sample (1 warm-up + 3 timed steps) at
    torch.set_num_threads(n) for n = the host's physical cores, 32 and 8 finds the fastest thread count; BASELINE.md section 3's
    protocol -- 5 warm-up steps, `steps` (default 20) timed steps, MEDIAN -- then runs at that count: `value` / `cores`."""
    from oracle import cunet_ref as O
    model, phys = host_cpu()
    usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    spec = O.Spec(4, 32, 128, class_num, layers, 1, layers)
    x, t = O.synthetic_batch(4, class_num, 256, seed=0)
    saved = torch.get_num_threads()

    def run(n, warm, timed):
        torch.set_num_threads(n)
        st = O.init_state(spec, seed=2)
        opt = {}
        for _ in range(warm):                        # (the first step is several times slower)
            O.train_step(spec, st, x, t, opt)
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            O.train_step(spec, st, x, t, opt)
            ts.append(time.perf_counter() - t0)
        return round(4.0 / statistics.median(ts), 3)
    try:
        counts = sorted({max(1, min(phys, usable)), min(32, usable), min(8, usable)}, reverse=True)
        short = {n: run(n, 1, 3) for n in counts}
        best_n = max(short, key=lambda n: short[n])
        value = run(best_n, 5, steps)
    finally:
        torch.set_num_threads(saved)
    return {'value': value, 'unit': 'images/sec', 'cores': best_n, 'kind': 'port',
            'protocol': f'5 warm-up + {steps} timed steps, median (BASELINE.md section 3)',
            'cpu_model': model, 'physical_cores': phys,
            'short_samples': {str(n): v for n, v in short.items()},      # threads -> img/s, 1 warm-up + 3 timed steps, median
            'sample': f'CU-Net-{layers} o1 K={class_num} bs=4 256x256 fp32 train step (fwd+MSE+bwd+RMSprop), oracle/cunet_ref.py, torch CPU {torch.__version__}'}


def load_mfma_busy(cls, workload_key):
    """MFMA pipe utilisation of a kernel class, alone on the GPU, from the newest committed SQ-counter pass of THIS workload
    (profiles/rNN_*mfma_busy.json, written by tools/pmc_summary.py --classes from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
    ...` run with the side stream off): busy SIMD-cycles / (launch duration x 2.4 GHz x 1024 SIMDs)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_busy.json')), reverse=True):
        try:
            tj = json.load(open(path))
        except Exception:
            continue
        if tj.get('workload') != workload_key:
            continue
        ent = tj.get('classes', {}).get(cls)
        if ent:
            return ent, f'{os.path.basename(path)}@{tj.get("commit", "?")}'
    return None, None


def load_traffic(cls, workload_key):
    """HBM bytes per launch of a kernel class from the newest committed PMC passes of THIS workload
    (rocprofv3 --pmc cannot run inside this process): profiles/rNN_traffic.json, produced by tools/profile_r04.sh +
    tools/pmc_traffic.py.  Returns (bytes or None, source string)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        try:
            tj = json.load(open(path))
        except Exception:
            continue
        if tj.get('workload', '2,68,24,f32') != workload_key:
            continue
        ent = tj.get('classes', {}).get(cls)
        if ent:
            return ent['hbm_bytes_per_launch'], f'{os.path.basename(path)}@{tj.get("commit", "r01")}'
    return None, None


def coll_device(pg, dev):
    """Device of the small tensors handed to collectives: the GPU under RCCL, the host under gloo (the de-risking backend)."""
    import torch.distributed as dist
    return torch.device('cpu') if dist.get_backend(pg) == 'gloo' else dev


def measure(dev, pg, rank, world, L, K, bs, steps, warmup, mode='fp32', bits_w=0, forward_only=False, popcount=False,
            profile_out='', seed=2, serial=False, force_class=None):
    """Time `steps` steps of one workload.  mode: 'fp32' | 'bf16' (activations) | 'bf16_grads' (+ gradient tensors).
    serial: build the plan with CUNET_NO_SIDE_STREAM (weight gradients on the caller's stream, so every kernel runs ALONE on
    the GPU): the per-launch durations of a class are then the kernel's own, not its share of a GPU it co-occupies."""
    import cu_net_amd
    from cu_net_amd.trainer import FusedTrainer
    bf16 = mode != 'fp32'
    torch.manual_seed(seed)
    net = cu_net_amd.create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=K,
                                   layer_num=L, order=1, loss_num=L).to(dev)
    net.train()
    quan = None
    if bits_w > 0:
        from cu_net_amd.quant import QuanOp
        quan = QuanOp(net, bits_w=bits_w, bits_i=8, bits_g=8)
    kw = {}
    if popcount:
        kw['popcount'] = True
    tr = FusedTrainer(net, lr=2.5e-4, alpha=0.99, eps=1e-8, process_group=pg, quan_op=quan, bf16=bf16,
                      bf16_grads=mode == 'bf16_grads', **kw)
    tr.broadcast_parameters(0)
    x, t = synthetic_batch(bs, K, 256, seed=1000 + rank, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    if forward_only:
        net.eval()

        def one_step():
            with torch.no_grad():
                outs = net.forward_bf16(x) if bf16 else net(x)      # the public path: loss_num NCHW heat maps
            return outs[-1][0, 0, 0, 0]
    else:
        def one_step():
            return tr.step(x, t)
    if serial:
        os.environ['CUNET_NO_SIDE_STREAM'] = '1'       # read when the plan handle is created
    try:
        plan = net._get_plan(bs, 256, 256, not forward_only, bf16=bf16)
    finally:
        if serial:
            del os.environ['CUNET_NO_SIDE_STREAM']
    # ---- warm-up; the last warm-up step is profiled per kernel class to pick the dominant one
    for i in range(max(warmup, 1)):
        if i == max(warmup, 1) - 1:
            plan.handle.profile_reset()
            plan.handle.profile_begin(1)
        loss = one_step()
    torch.cuda.synchronize(dev)
    prof_all = plan.handle.profile_collect()
    prof_main = plan.handle.profile_collect(0)      # launches on the caller's stream only
    prof_side = plan.handle.profile_collect(1)      # launches on the library's side stream only
    plan.handle.profile_begin(0)
    have_classes = any(v[0] for v in prof_all.values())
    # The roofline class is the largest one ON THE CALLER'S STREAM -- the step's critical path -- counted over the launches that
    # really ran there (the library tags every record with its stream).  On the low-priority side stream a launch's duration
    # includes the time it is switched out for the caller's kernels; when a class's inflated sum there is the largest of the step
    # it is reported next to the roofline as `largest_side_stream_class`.
    which = -1 if serial else 0
    on_path = prof_all if serial else prof_main
    dominant = force_class or max(on_path.items(), key=lambda kv: kv[1][1])[0]
    side_top = max(prof_side.items(), key=lambda kv: kv[1][1])
    side_note = None
    if not serial and not force_class and side_top[1][1] > on_path[dominant][1]:
        side_note = {'kernel': side_top[0], 'launches': side_top[1][0], 'sum_ms_in_profiled_step': round(side_top[1][1], 3)}
    if rank == 0 and have_classes and not serial:
        tot = sum(v[1] for v in prof_all.values())
        lines = [f'per-class profile of one warm-up step, CU-Net-{L} K={K} {mode} bits_w={bits_w} (sum of kernel times {tot:.3f} ms):']
        for name, (cnt, ms, fl, by) in sorted(prof_all.items(), key=lambda kv: -kv[1][1]):
            if cnt:
                lines.append(f'  {name:22s} launches={cnt:4d} ms={ms:8.3f} ({100 * ms / tot:5.1f}%) '
                             f'TFLOP/s={fl / ms / 1e9 if ms else 0:7.2f} GB/s={by / ms / 1e6 if ms else 0:8.1f}')
        log('\n'.join(lines))
        if profile_out:
            with open(profile_out, 'a') as f:
                f.write('\n'.join(lines) + '\n')

    # ---- timed region: exactly `steps` steps; HIP events only around the dominant class, one event per step
    # A timed HIP event is a marker packet on the stream and the kernel behind it waits for it to retire: ~6 us per event when the
    # class runs on the caller's stream (the 1x1 data gradient: 47 launches x 2 events = 0.5 ms of a 7.2 ms CU-Net-2 step, ~1.2 ms
    # of a CU-Net-8 step).  The dominant class is therefore bracketed in TWO steps of the timed region only (the first and the
    # middle one): its average launch duration is still measured live, inside the timed region, on the stream the kernel runs on,
    # over ~100 - 500 launches, and `value` is not taxed by the measurement.
    EVENT_STRIDE = max(4, steps // 2)
    plan.handle.profile_reset()
    cls_index = plan.handle.profile_class_index(dominant)
    use_events = not os.environ.get('CUNET_BENCH_NO_CLASS_EVENTS')      # (tools only: how much do the per-launch events cost?)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        if use_events:
            plan.handle.profile_begin(2 if i % EVENT_STRIDE == 0 else 0, cls_index)
        loss = one_step()
        evs[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    prof = plan.handle.profile_collect(which)
    plan.handle.profile_begin(0)
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    if pg is not None:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_device(pg, dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    res = {'dt': dt, 'final_loss': float(loss), 'per_step_ms': per_step}
    imgs = world * bs * steps
    value = imgs / dt
    res['value'] = value
    res['ms_per_step'] = 1e3 * dt / steps
    res['ms_per_step_median'] = statistics.median(per_step)
    cnt, ms, fl, by = prof[dominant]
    peak_mfma = PEAK_BF16_MFMA_TFLOPS if dominant.endswith('_bf16') else PEAK_F32_MFMA_TFLOPS      # classes that run on bf16 MFMA carry the suffix
    # fp32 convolution classes on the split contraction (F32_SPLIT): six bf16 products per fp32 product -- the matrix-pipe ceiling of the
    # class's fp32-equivalent flops is the bf16 peak / 6 (416.7 TFLOP/s), its ridge 52 flop/B: the 1x1 data gradient (24 flop/B at 192
    # input channels) is bound by HBM there, not by the pipe
    on_split = F32_SPLIT and not bf16 and not dominant.endswith('_bf16') and dominant.startswith('conv')
    if on_split:
        peak_mfma = PEAK_BF16_MFMA_TFLOPS / 6
    roof = None
    if have_classes and ms > 0:
        # the roof that bounds the class: arithmetic intensity of its algorithmic work against the ridge of ITS MFMA peak
        # (fp32 1x1 weight gradient: 46 flop/B > 19.7 -> MFMA; the same contraction on bf16 MFMA: 91 flop/B < 312 -> HBM)
        ai = fl / by if by > 0 else float('inf')
        ridge = peak_mfma * 1e12 / (PEAK_HBM_GBS * 1e9)
        if fl > 0 and ai >= ridge:
            achieved = fl / (ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': dominant, 'achieved': round(achieved, 3), 'peak': peak_mfma,
                    'unit': 'TFLOP/s', 'frac': round(achieved / peak_mfma, 4), 'traffic': None,
                    'launches': cnt, 'avg_launch_us': round(1e3 * ms / max(cnt, 1), 2),
                    'achieved_algorithmic_GBs': round(by / (ms * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        else:
            achieved = by / (ms * 1e-3) / 1e9
            roof = {'bound': 'hbm', 'kernel': dominant, 'achieved': round(achieved, 1), 'peak': PEAK_HBM_GBS,
                    'unit': 'GB/s', 'frac': round(achieved / PEAK_HBM_GBS, 4), 'traffic': None,
                    'launches': cnt, 'avg_launch_us': round(1e3 * ms / max(cnt, 1), 2)}
            if fl > 0:
                roof['achieved_TFLOPs'] = round(fl / (ms * 1e-3) / 1e12, 2)
                roof['arithmetic_intensity_flop_per_byte'] = round(ai, 1)
        if on_split:
            roof['matrix_pipe'] = 'bf16 MFMA x6 (f32_split): ceiling %.1f TFLOP/s, ridge %.0f flop/B' % (peak_mfma, ridge)
            if fl > 0:
                roof['frac_of_f32_mfma_peak'] = round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)      # (the denominator of rounds 1-3)
        wkey = f'{L},{K},{bs},{"f32" if not bf16 else mode}'
        tb, src = load_traffic(dominant, wkey)
        if tb is not None:
            roof['traffic'] = tb
            roof['traffic_source'] = src
            roof['algorithmic_bytes_per_launch'] = round(by / max(cnt, 1))
        mb, msrc = load_mfma_busy(dominant, wkey)
        if mb is not None:       # north_star: "MFMA utilisation against gfx950 peak" (counter-based; the live figure above is flop-based)
            roof['mfma_busy'] = mb['mfma_busy']
            roof['mfma_busy_detail'] = {k: v for k, v in mb.items() if k != 'mfma_busy'}
            roof['mfma_busy_source'] = msrc
        if side_note:
            roof['largest_side_stream_class'] = side_note
    res['roofline'] = roof
    g = (FWD_GFLOP_PER_IMG if forward_only else TRAIN_GFLOP_PER_IMG).get((L, K))
    if g:
        res['step_tflops'] = round(g * value / 1e3, 2)
        if bf16:
            res['step_frac_of_bf16_mfma_peak'] = round(g * value / 1e3 / (PEAK_BF16_MFMA_TFLOPS * world), 4)
        else:
            res['step_frac_of_f32_mfma_peak'] = round(g * value / 1e3 / (PEAK_F32_MFMA_TFLOPS * world), 4)
    fb = FWD_MB_PER_IMG_F32.get((L, K))
    if fb:      # whole-step algorithmic bytes (B_train = 3 x B_fwd; bf16 storage halves them) against the HBM peak
        mb = fb * (0.5 if bf16 else 1.0) * (1.0 if forward_only else 3.0)
        res['step_algorithmic_GBs'] = round(mb * value / 1e3, 1)
        res['step_frac_of_hbm_peak'] = round(mb * value / 1e3 / (PEAK_HBM_GBS * world), 4)
    del tr, net, plan
    torch.cuda.empty_cache()
    return res


F32_SPLIT = 1      # planner option f32_split as this process has it (main() follows --planner-opt and the fp32-pipe `also` line)
# (what the fields of the line mean -- the split contraction, `roofline.alone`, `traffic` / `mfma_busy` and their `*_source` tags
# "<file under profiles/>@<commit>", `largest_side_stream_class`, the `also` entries -- is written down once; the line
# itself stays under ~6 KB so that the driver's record of it is complete)
SPLIT_NOTE = 'split-bf16: fp32 operands cut into 3 bf16 pieces, 6 bf16-MFMA products, fp32 accumulate (f32_split=1; DESIGN.md 4a)'


def workload_name(L, K, bs, mode, bits_w, forward_only, world, popcount=False):
    return (f'CU-Net L={L} order=1 loss_num={L} bs={bs}/GPU 256x256 K={K} '
            + (f'QuanOp bits_w={bits_w} ' if bits_w > 0 else '')
            + ('AND-popcount fwd ' if popcount else '')
            + (('bf16-storage' if mode != 'fp32' else 'fp32') + ' eval forward' if forward_only else
               {'fp32': 'fp32', 'bf16': 'bf16-act', 'bf16_grads': 'bf16 act+grad tensors'}[mode]
               + ' train step (fwd+MSE+bwd+RMSprop' + ('+RCCL grad all-reduce)' if world > 1 else ')')))


def compact_roofline(roof):
    """The `also` entries carry the contract's roofline fields only (the headline entry keeps the detail)."""
    if not roof:
        return roof
    keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches', 'avg_launch_us', 'mfma_busy')
    return {k: roof[k] for k in keep if k in roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--class-num', type=int, default=68)
    ap.add_argument('--bs', type=int, default=24, help='per-GPU batch')
    ap.add_argument('--cpu-steps', type=int, default=20, help='timed steps of the CPU baseline at its best thread count (5 warm-up steps in front)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='skip the extra CU-Net-8 bf16 / CU-Net-16 binary-weight lines')
    ap.add_argument('--also-steps', type=int, default=20)
    ap.add_argument('--no-alone', action='store_true', help='skip the extra (untimed) pass that times the dominant class with the side stream off')
    ap.add_argument('--profile-out', default='')
    ap.add_argument('--forward-only', action='store_true', help='inference: eval-mode forward (running statistics), no loss / backward')
    ap.add_argument('--bf16', action='store_true', help='bf16 activation storage + bf16 MFMA forward (train step: gradients, weights, optimiser stay fp32; '
                    'with --forward-only: bf16-storage inference)')
    ap.add_argument('--bf16-grads', action='store_true', help='with --bf16: the gradient tensors of backward (dY, dz, dX) stored as bf16 too')
    ap.add_argument('--bits-w', type=int, default=0, help='>0: quantised-weight train step (QuanOp, utils/quantize.py), e.g. 1 = BASELINE config 5')
    ap.add_argument('--popcount', action='store_true', help='with --bits-w 1/2: forward convs of the quantised layers on the AND-popcount kernel')
    ap.add_argument('--planner-opt', action='append', default=[], metavar='NAME=VALUE',
                    help='cunet_set_planner_option(NAME, VALUE) before any plan is created (kernel-selection sweeps); repeatable')
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the N ranks ourselves (one process per GPU, RCCL), exactly as the driver's
    #      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` would
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log('bench.py: --gpus %d without WORLD_SIZE: re-launching as  %s' % (args.gpus, ' '.join(cmd)))
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        raise SystemExit(subprocess.call(cmd, env=env))
    if os.environ.get('CUNET_LIB_PATH'):
        raise SystemExit('bench.py measures the shipped library (cu_net_amd/libcunet_hip.so): unset CUNET_LIB_PATH '
                         '(the -DCUNET_TUNING build is for tools/ only)')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         '(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N) or run without a launcher')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # CUNET_BENCH_BACKEND=gloo (tests/test_gpu_dp.py::test_bench_two_ranks_end_to_end only): the N-rank code path of this file --
    # self-launch, rank-0-only printing, the `also` entries of N > 1, the max-over-ranks clock -- on a box with FEWER GPUs than
    # ranks: ranks share the GPUs round-robin and the gradient buckets travel over gloo staged through the host (parallel.py).
    # The line it prints says so (`backend`) and is not a measurement; the driver's runs use RCCL.
    backend = os.environ.get('CUNET_BENCH_BACKEND', 'nccl')
    if backend not in ('nccl', 'gloo'):
        raise SystemExit('CUNET_BENCH_BACKEND must be nccl (RCCL, default) or gloo')
    ndev = torch.cuda.device_count()
    if backend == 'nccl' and local_rank >= ndev:
        raise SystemExit(f'bench.py: rank {local_rank} has no GPU of its own ({ndev} visible): RCCL needs one GPU per rank')
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    pg = None
    ranks_seen = 1
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run even one rank goes through RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        pg = dist.group.WORLD
        one = torch.ones(1, device=coll_device(pg, dev))
        dist.all_reduce(one)                    # what the collective library itself saw: every rank contributes 1
        ranks_seen = int(one.item())

    import cu_net_amd
    split = 1                                   # planner option f32_split (default on): see SPLIT_NOTE
    for po in args.planner_opt:
        name, _, val = po.partition('=')
        cu_net_amd._lib.check(cu_net_amd._lib.lib().cunet_set_planner_option(name.encode(), int(val)), 'cunet_set_planner_option')
        if name == 'f32_split':
            split = int(val)
    global F32_SPLIT
    F32_SPLIT = split
    if rank == 0:
        log('library: %s  (%s)' % (cu_net_amd._lib.LIB_PATH, cu_net_amd._lib.lib().cunet_version().decode()))
    L, K, bs = args.layers, args.class_num, args.bs
    mode = 'bf16_grads' if args.bf16_grads else ('bf16' if args.bf16 else 'fp32')
    is_default = (L, K, bs, mode, args.bits_w, args.forward_only, args.popcount) == (2, 68, 24, 'fp32', 0, False, False)
    r = measure(dev, pg, rank, world, L, K, bs, args.steps, args.warmup, mode, args.bits_w, args.forward_only,
                args.popcount, args.profile_out)
    alone = None
    if world == 1 and is_default and r['roofline'] and not args.no_alone and 'CUNET_NO_SIDE_STREAM' not in os.environ:
        # the same class timed with nothing else on the GPU (outside the timed region; a separate plan without the side stream)
        try:
            a = measure(dev, None, 0, 1, L, K, bs, 8, 3, mode, args.bits_w, False, args.popcount, serial=True,
                        force_class=r['roofline']['kernel'])
            ar = a['roofline']
            alone = {'avg_launch_us': ar['avg_launch_us'], 'achieved': ar['achieved'], 'frac': ar['frac'], 'launches': ar['launches'],
                     'ms_per_step_serial': round(a['ms_per_step_median'], 3)}
        except Exception as ex:
            alone = {'error': repr(ex)}
    if rank == 0:
        if alone is not None and r['roofline']:
            r['roofline']['alone'] = alone
        out = {
            'metric': ('images/sec inference forward' if args.forward_only else 'images/sec train step')
                      + ', 256x256x3 -> 64x64xK heatmaps, CU-Net-%d' % L,
            'value': round(r['value'], 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(r['ms_per_step'], 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if mode != 'fp32' else 'f32', 'data': 'synthetic',
            'config': {'workload': workload_name(L, K, bs, mode, args.bits_w, args.forward_only, world, args.popcount),
                       'global_batch': world * bs, 'parallelism': f'dp{world}'},
            'ms_per_step_median': round(r['ms_per_step_median'], 3),
            'value_at_median': round(bs * world / r['ms_per_step_median'] * 1e3, 2),
            'ranks_seen_by_rccl': ranks_seen,
            'backend': 'rccl' if backend == 'nccl' else 'gloo (host-staged: not a measurement)',
            'library_path': os.path.relpath(cu_net_amd._lib.LIB_PATH, ROOT),
            'roofline': r['roofline'],
            'final_loss': r['final_loss'],
        }
        if mode == 'fp32':
            out['config']['contraction'] = SPLIT_NOTE if split else 'fp32 MFMA (f32_split=0)'
        if not args.forward_only:
            # (FusedTrainer.step: the MSE and its gradient are computed in the heads' epilogues from the staged target, so the timed step never
            # converts the heat maps to NCHW -- they stay NHWC in the workspace; tr.last_outputs() makes the NCHW copies on request)
            out['config']['heat_maps'] = 'NHWC in the workspace, loss fused into the head epilogues'
        for k in ('step_tflops', 'step_frac_of_f32_mfma_peak', 'step_frac_of_bf16_mfma_peak', 'step_algorithmic_GBs', 'step_frac_of_hbm_peak'):
            if k in r:
                out[k] = r[k]
    # ---- the other headline configurations of BASELINE.json, attached as `also`:
    #      N = 1: configs[2] (CU-Net-8, bf16 storage), configs[4] (CU-Net-16, binary weights; MFMA and AND-popcount forward) and
    #             the forward-only (inference) rates of the bench workload (SURVEY 8d "also forward-only img/s");
    #      N > 1: configs[3] (CU-Net-8, K = 16, 24 images per rank, bucketed RCCL all-reduce) -- every rank takes part.
    if is_default and not args.no_also:
        also = []
        if world == 1:
            extra = [(8, 68, 'bf16_grads', 0, False, False), (16, 16, 'fp32', 1, False, False), (16, 16, 'fp32', 1, True, False),
                     (2, 68, 'fp32', 0, False, True), (2, 68, 'bf16', 0, False, True)]
        else:
            extra = [(8, 16, 'bf16_grads', 0, False, False), (8, 16, 'fp32', 0, False, False)]
        if world == 1 and split:
            # the bench workload on the fp32 matrix pipe (every earlier round's kernel selection), so that both numbers are on record
            extra.append((2, 68, 'fp32', 0, False, False, 0))
        for spec in extra:
            (l2, k2, m2, bw, pc, fwd), sp2 = spec[:6], (spec[6] if len(spec) > 6 else split)
            name2 = workload_name(l2, k2, bs, m2, bw, fwd, world, pc) + ('' if sp2 == split else ' [f32_split=0: fp32 MFMA]')
            try:
                if sp2 != split:
                    cu_net_amd._lib.set_planner_option('f32_split', sp2)
                    F32_SPLIT = sp2
                try:
                    # (a forward-only step is 1 - 2 ms: five times the steps, so that one host hiccup does not halve the figure)
                    nst = args.also_steps * (5 if fwd else 1)
                    e = measure(dev, pg, rank, world, l2, k2, bs, nst, max(3, args.warmup), m2, bw, fwd, pc, args.profile_out)
                finally:
                    if sp2 != split:
                        cu_net_amd._lib.set_planner_option('f32_split', split)
                        F32_SPLIT = split
                ent = {'workload': name2, 'value': round(e['value'], 2), 'unit': 'images/sec', 'steps': nst,
                       'ms_per_step': round(e['ms_per_step'], 3), 'dtype': 'bf16' if m2 != 'fp32' else 'f32',
                       'roofline': compact_roofline(e['roofline']), 'final_loss': round(e['final_loss'], 6)}
                for k in ('step_tflops', 'step_frac_of_hbm_peak'):
                    if k in e:
                        ent[k] = e[k]
                also.append(ent)
            except Exception as ex:      # an extra line must never take the headline measurement down
                if world > 1:
                    raise                # (a rank that drops out of a collective would hang the others)
                also.append({'workload': name2, 'error': repr(ex)})
        if rank == 0:
            out['also'] = also
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            out['cpu_baseline'] = cpu_baseline(L, K, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1 or pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
