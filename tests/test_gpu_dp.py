"""GPU (-m gpu): the data-parallel step (cu-net.py:59 torch.nn.DataParallel -> one process per GPU + bucketed
all-reduce) with TWO ranks launched by torch.distributed.run.

With >= 2 GPUs each rank owns one and the buckets go over RCCL; on a one-GPU box both ranks share cuda:0 and the
buckets go over gloo (host-staged) -- everything else (per-rank BatchNorm statistics, cunet_backward_ex bucket
callbacks in backward's completion order, side-stream joins, 1/world folded into the fused RMSprop) is the same code.

Checks: (1) the one-time broadcast made rank 1 start from rank 0's parameters; (2) both ranks end with IDENTICAL
gradient arenas and parameters; (3) the all-reduced arena equals g_A + g_B where g_A, g_B are the HIP path's own
single-process gradients of the two shards (1e-5: the weight-gradient atomics are order-dependent in the last bits);
(4) the parameters equal RMSprop(p0, (g_A + g_B) / 2); (5) each rank's loss equals the oracle's loss on its shard, and
the reduced arena equals the oracle's summed shard gradients at the whole-network sanity bound; (6) buckets were
reduced in the order cunet_bucket_order promises."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(neck_size=2, growth_rate=16, init_chan_num=32, class_num=6, layer_num=3, order=1, loss_num=3)
GLOBAL_BATCH, HW = 4, 128
# BASELINE config 4's network (CU-Net-8, K = 16, production widths), two images per rank
CFG4 = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=8, order=1, loss_num=8)
CASES = {'toy': (CFG, 4, 128), 'config4': (CFG4, 4, 256)}


def make_inputs(case='toy'):
    cfg, gb, hw = CASES[case]
    g = torch.Generator().manual_seed(62)
    x = torch.rand(gb, 3, hw, hw, generator=g)
    t = torch.rand(gb, cfg['class_num'], hw // 4, hw // 4, generator=g) * 0.3
    return x, t


def _launch_two_ranks(tmp_path, case):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['CUNET_DP_CASE'] = case
    port = 29700 + (os.getpid() + len(case)) % 1500
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', '_dp_worker.py'), str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [torch.load(os.path.join(str(tmp_path), f'rank{i}.pt')) for i in range(2)]


def test_two_rank_step_on_the_config4_network(tmp_path):
    """The same two-rank step on BASELINE config 4's network -- CU-Net-8, K = 16, production widths, 256 x 256, two images per
    rank (RCCL when two devices are visible, host-staged gloo on a one-GPU box): nine gradient buckets of ~4 MB in backward's
    completion order, replicas identical afterwards, the reduced arena equal to the sum of the HIP path's own single-process
    shard gradients, parameters equal to RMSprop on their mean."""
    import cu_net_amd
    from cu_net_amd.parallel import shard_batch
    from cu_net_amd.trainer import FusedTrainer
    from oracle import cunet_ref as O
    ranks = _launch_two_ranks(tmp_path, 'config4')
    cfg, gb, hw = CASES['config4']
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=61)
    x, t = make_inputs('config4')
    assert torch.equal(ranks[1]['p0'], ranks[0]['p0'])
    assert torch.equal(ranks[0]['grads'], ranks[1]['grads']) and torch.equal(ranks[0]['params'], ranks[1]['params'])
    assert ranks[0]['ranks_seen'] == 2
    for rk in ranks:
        assert rk['reduced'] == rk['order'] == list(range(cfg['layer_num'] - 1, -1, -1)) + [cfg['layer_num']]
    gs = []
    for i in range(2):
        lo, hi = shard_batch(gb, i, 2)
        net = cu_net_amd.create_cu_net(**cfg)
        net.load_state_dict(st)
        net = net.cuda().train()
        tr = FusedTrainer(net)
        loss = float(tr.step(x[lo:hi].cuda(), t[lo:hi].cuda()))
        torch.cuda.synchronize()
        assert abs(loss - ranks[i]['loss']) <= 1e-4 * abs(loss)
        gs.append(net._grad_arena.detach().cpu().clone())
    gsum = gs[0] + gs[1]
    # (two runs of the same shard differ in the order of the fp64 statistics atomics -- 1e-7 relative -- and eight U-Nets in train mode
    # amplify that by ~3.5x each, forward and again backward: relative L2 at the whole-network sanity level, not element-wise)
    rel = float((ranks[0]['grads'] - gsum).double().norm() / gsum.double().norm())
    assert rel <= 5e-2, rel
    g = ranks[0]['grads'] * 0.5
    v = 0.01 * g * g
    expect = ranks[0]['p0'] - 2.5e-4 * g / (v.sqrt() + 1e-8)
    assert float((ranks[0]['params'] - expect).abs().max()) <= 2e-6


def test_two_rank_step_equals_mean_of_shard_gradients(tmp_path):
    import cu_net_amd
    from cu_net_amd.parallel import shard_batch
    from cu_net_amd.trainer import FusedTrainer
    from oracle import cunet_ref as O
    ranks = _launch_two_ranks(tmp_path, 'toy')
    spec = O.Spec(**CFG)
    st = O.init_state(spec, seed=61)
    x, t = make_inputs()
    # (1) broadcast
    ref_net = cu_net_amd.create_cu_net(**CFG)
    ref_net.load_state_dict(st)
    assert torch.equal(ranks[1]['p0'], ranks[0]['p0']) and torch.equal(ranks[0]['p0'], ref_net._param_arena)
    # (2) replicas agree
    assert torch.equal(ranks[0]['grads'], ranks[1]['grads'])
    assert torch.equal(ranks[0]['params'], ranks[1]['params'])
    assert ranks[0]['ranks_seen'] == 2
    # (6) bucket order
    for rk in ranks:
        assert rk['reduced'] == rk['order'] == list(range(CFG['layer_num'] - 1, -1, -1)) + [CFG['layer_num']]
    # single-process shard gradients on the HIP path
    gs, losses = [], []
    for i in range(2):
        lo, hi = shard_batch(GLOBAL_BATCH, i, 2)
        net = cu_net_amd.create_cu_net(**CFG)
        net.load_state_dict(st)
        net = net.cuda().train()
        tr = FusedTrainer(net)
        losses.append(float(tr.step(x[lo:hi].cuda(), t[lo:hi].cuda())))
        torch.cuda.synchronize()
        gs.append(net._grad_arena.detach().cpu().clone())
        if i == 0:                                                   # rank 0's BatchNorm running statistics are ITS shard's
            assert torch.allclose(ranks[0]['buffers'], net._buffer_arena.cpu(), rtol=1e-5, atol=1e-6)
    gsum = gs[0] + gs[1]
    scale = float(gsum.abs().max())
    # (3)
    assert float((ranks[0]['grads'] - gsum).abs().max()) <= 1e-5 * scale, float((ranks[0]['grads'] - gsum).abs().max()) / scale
    # (4) RMSprop on the averaged gradient (first step: v = (1 - alpha) g^2)
    g = ranks[0]['grads'] * 0.5
    v = 0.01 * g * g
    expect = ranks[0]['p0'] - 2.5e-4 * g / (v.sqrt() + 1e-8)
    assert float((ranks[0]['params'] - expect).abs().max()) <= 2e-6
    # (5) the oracle on each shard
    off = {name: (o, n, shape) for name, kind, shape, o, n in ref_net._entries if kind == 0}
    osum = torch.zeros_like(gsum)
    for i in range(2):
        lo, hi = shard_batch(GLOBAL_BATCH, i, 2)
        ol, _, og = O.train_step(spec, {k: v.clone() for k, v in st.items()}, x[lo:hi], t[lo:hi], apply_update=False)
        assert abs(ranks[i]['loss'] - float(ol)) <= 2e-3 * abs(float(ol)), (i, ranks[i]['loss'], float(ol))
        assert abs(losses[i] - ranks[i]['loss']) <= 1e-6 * abs(losses[i])
        for k, gr in og.items():
            if gr is not None:
                o, n, shape = off[k]
                osum[o:o + n] += gr.reshape(-1)
    rel2 = float((ranks[0]['grads'] - osum).double().norm() / osum.double().norm())
    assert rel2 <= 0.2, rel2


def test_failing_bucket_callback_aborts_the_step():
    """A collective that cannot be issued (an exception inside the bucket callback, which C calls) must surface as that
    exception and must NOT reach the optimiser (ctypes would print and swallow it)."""
    import cu_net_amd
    from cu_net_amd.trainer import FusedTrainer
    from oracle import cunet_ref as O
    spec = O.Spec(**CFG)
    net = cu_net_amd.create_cu_net(**CFG)
    net.load_state_dict(O.init_state(spec, seed=3))
    net = net.cuda().train()
    x, t = make_inputs()
    plan = net._get_plan(2, HW, HW, True)
    plan.forward(x[:2].cuda(), True, want_outputs=False)
    plan.loss_mse(t[:2].cuda())
    p0 = net._param_arena.clone()
    calls = []

    def boom(b):
        calls.append(b)
        if len(calls) == 2:
            raise RuntimeError('simulated RCCL failure')
    with pytest.raises(RuntimeError, match='simulated RCCL failure'):
        plan.backward(None, on_bucket=boom)
    torch.cuda.synchronize()
    assert len(calls) == 2                                  # backward stopped at the failing bucket
    assert torch.equal(net._param_arena, p0)
    with pytest.raises(cu_net_amd.CUNetError):              # the aborted backward consumed the forward
        plan.backward(None)
    # and the plan is usable again
    tr = FusedTrainer(net)
    assert torch.isfinite(tr.step(x[:2].cuda(), t[:2].cuda()))


def test_bench_two_ranks_end_to_end():
    """De-risks the driver's first `bench.py --gpus N` run with N > 1 (cu-net.py:59's replacement under the bench contract): the whole
    N-rank path of bench.py -- its self-launch through torch.distributed.run, one process per rank, the barrier + max-over-ranks
    clock, rank-0-only printing, the BASELINE config-4 `also` entries of N > 1, a clean exit of both ranks -- executed with
    world size 2.  With two GPUs visible it runs over RCCL exactly as the driver will; on a one-GPU box the ranks share cuda:0 and
    CUNET_BENCH_BACKEND=gloo carries the gradient buckets through the host (the printed line says so; it is not a measurement)."""
    import json
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    rccl = torch.cuda.device_count() >= 2
    env['CUNET_BENCH_BACKEND'] = 'nccl' if rccl else 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--also-steps', '2',
           '--no-cpu-baseline', '--no-alone']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]               # rank 0 prints ONE JSON line, rank 1 none
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen_by_rccl'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['config']['global_batch'] == 48 and out['config']['parallelism'] == 'dp2' and out['scaling'] == 'weak'
    assert out['backend'].startswith('rccl' if rccl else 'gloo')
    assert out['value'] > 0 and abs(out['value'] - 48 * 3 / (out['ms_per_step'] * 3e-3)) <= 1e-2 * out['value']
    also = out['also']
    assert len(also) == 2 and all('error' not in e for e in also), also
    assert all('L=8' in e['workload'] and 'K=16' in e['workload'] and 'RCCL grad all-reduce' in e['workload'] and e['value'] > 0 for e in also)
    assert len(lines[0]) < 6144, len(lines[0])            # (the driver keeps a bounded tail of stdout: the whole line has to fit)
    assert 'cpu_baseline' not in out
    for key in ('roofline', 'final_loss', 'library_path'):
        assert key in out
