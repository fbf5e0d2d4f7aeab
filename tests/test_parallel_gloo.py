"""CPU, world_size 2 over gloo: the data-parallel plumbing (bucket layout from the C plan, per-bucket
all-reduce in backward's completion order, one-time state broadcast, 1/world scaling, batch sharding).
The kernels themselves need a GPU (tests/test_gpu_dp.py runs the same two-rank step on the HIP path); here
`test_two_rank_bucket_allreduce_gloo` moves seeded random arenas through the bucket plumbing and
`test_two_rank_step_equals_dataparallel_semantics` feeds it per-rank gradients computed by the CPU oracle."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cu_net_amd._lib import PlanHandle
from cu_net_amd.parallel import BucketAllReducer, broadcast_state, shard_batch

CFG = dict(neck_size=2, growth_rate=4, init_chan_num=8, class_num=3, layer_num=3, order=1, loss_num=3)


def _layout():
    plan = PlanHandle(**CFG, batch=2, height=64, width=64)
    return plan, plan.state_entries(), plan.buckets(), plan.param_numel, plan.bucket_order()


def test_bucket_layout_is_bucket_major():
    plan, ents, buckets, numel, order = _layout()
    L = CFG['layer_num']
    assert len(buckets) == L + 1
    red = BucketAllReducer(buckets)
    assert red.covers(numel)
    # every parameter lies inside exactly the bucket its U-Net index says
    for name, kind, shape, off, n in ents:
        if kind != 0:
            continue
        if name.startswith('features.'):
            b = L
        elif name.startswith('intermedia.adapters.'):
            b = int(name.split('.')[2]) + 1
        elif name.startswith('linears.'):
            b = int(name.split('.')[1])
        else:
            parts = name.split('.')
            b = int(parts[parts.index('layers') + 1] if 'layers' in parts else
                    parts[[i for i, p in enumerate(parts) if p.startswith('adapters_')][0] + 1])
        begin, count = buckets[b]
        assert begin <= off and off + n <= begin + count, (name, b)
    # backward finishes the last U-Net first and the stem last
    assert order == list(range(L - 1, -1, -1)) + [L]


def _free_port():
    """A port the kernel just handed out (bound to port 0 and released): no collision between concurrent test sessions."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _run_ranks(target, world, timeout):
    """Spawn `world` ranks of `target(rank, world, port, queue)`, collect one result per rank, and ALWAYS terminate and join the children:
    a rank that dies hard leaves the others in init_process_group / all_reduce for ever otherwise (round-5 advice)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(5 if len(res) < len(procs) else 30)
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
    return res



def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        plan, ents, buckets, numel, order = _layout()
        pg = dist.group.WORLD
        # one-time broadcast: rank 1 starts from different parameters and must end up with rank 0's
        torch.manual_seed(100 + rank)
        params = torch.randn(numel)
        bufs = torch.randn(64)
        broadcast_state([params, bufs], 0, pg)
        torch.manual_seed(100)
        assert torch.equal(params, torch.randn(numel))
        # per-rank gradients, reduced bucket by bucket in backward's completion order
        torch.manual_seed(7 + rank)
        grads = torch.randn(numel)
        mine = grads.clone()
        red = BucketAllReducer(buckets, pg, overlap=True)
        red.begin_step()
        for b in order:
            red.reduce_bucket(grads, b)
        red.finish(grads)
        assert red.reduced == order
        torch.manual_seed(7 + (1 - rank))
        other = torch.randn(numel)
        inside = torch.zeros(numel, dtype=torch.bool)
        for begin, count in buckets:
            inside[begin:begin + count] = True
        assert torch.allclose(grads[inside], (mine + other)[inside])
        assert torch.equal(grads[~inside], mine[~inside])           # alignment padding is never touched
        # averaging: what the fused RMSprop sees is grads * (1/world)
        assert abs(red.world - world) == 0
        q.put((rank, 'ok'))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_bucket_allreduce_gloo():
    res = _run_ranks(_worker, 2, 120)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def _dp_worker(rank, world, port, q):
    """Each rank: oracle gradient of ITS shard (own BatchNorm statistics, loss = mean over the local elements), laid
    out in the C plan's flat arena, summed bucket by bucket, scaled by 1/world as the fused RMSprop does.  Rank 0 also
    evaluates torch.nn.DataParallel's semantics directly (cu-net.py:59,171-182: replicas forward their chunks with
    per-replica BatchNorm, the gathered outputs enter ONE loss over the global batch) and compares."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import cunet_ref as O
        torch.set_num_threads(2)
        plan, ents, buckets, numel, order = _layout()
        spec = O.Spec(**CFG)
        st = O.init_state(spec, seed=5)
        gen = torch.Generator().manual_seed(6)
        gb = 4
        x = torch.rand(gb, 3, 128, 128, generator=gen)
        tgt = torch.rand(gb, CFG['class_num'], 32, 32, generator=gen)
        lo, hi = shard_batch(gb, rank, world)
        _, _, grads = O.train_step(spec, {k: v.clone() for k, v in st.items()}, x[lo:hi], tgt[lo:hi], apply_update=False)
        off = {name: (o, n, shape) for name, kind, shape, o, n in ents if kind == 0}
        arena = torch.zeros(numel)
        for k, g in grads.items():
            if g is not None:
                o, n, shape = off[k]
                arena[o:o + n] = g.reshape(-1)
        red = BucketAllReducer(buckets, dist.group.WORLD, overlap=True)
        red.begin_step()
        for b in order:
            red.reduce_bucket(arena, b)
        red.finish(arena)
        arena.mul_(1.0 / red.world)
        if rank == 0:
            state = {k: v.clone() for k, v in st.items()}
            names = O.param_names(spec)
            for n in names:
                state[n].requires_grad_(True)
            outs = []
            for r in range(world):                       # replicas: per-replica BatchNorm statistics
                a, b = shard_batch(gb, r, world)
                outs.append(O.forward(spec, state, x[a:b], True))
            gathered = [torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0]))]
            O.mse_loss(gathered, tgt).backward()          # one loss over the global batch
            worst = 0.0
            for n in names:
                if state[n].grad is None:
                    continue
                o, cnt, shape = off[n]
                ref = state[n].grad.reshape(-1)
                worst = max(worst, float((arena[o:o + cnt] - ref).abs().max() / (ref.abs().max() + 1e-12)))
            assert worst < 2e-5, worst
        q.put((rank, 'ok'))
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_dataparallel_semantics():
    res = _run_ranks(_dp_worker, 2, 300)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_shard_batch_matches_dataparallel_chunking():
    # torch.nn.DataParallel scatters with tensor.chunk(n): contiguous, ceil-sized chunks
    x = torch.arange(192)
    for world in (1, 2, 4, 8):
        chunks = x.chunk(world)
        for r in range(world):
            lo, hi = shard_batch(192, r, world)
            assert torch.equal(x[lo:hi], chunks[r])
    assert shard_batch(10, 3, 4) == (9, 10)


CFG4 = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=8, order=1, loss_num=8)      # BASELINE config 4's network


def _worker8(rank, world, port, q):
    """World size 8 over gloo with the REAL bucket table of BASELINE config 4 (CU-Net-8, K = 16: nine buckets, 32 MB of gradients,
    24 images per rank): state broadcast from rank 0, the nine all-reduces in backward's completion order with `overlap=True`,
    finish(), 1/world.  Every element of the arena is an exactly representable small integer pattern (rank + 1) * f(index), so the
    sum over 8 ranks is exact in fp32 whatever order gloo's ring adds in: the check is torch.equal, per bucket."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        plan = PlanHandle(**CFG4, batch=24, height=256, width=256)
        buckets, numel, order = plan.buckets(), plan.param_numel, plan.bucket_order()
        assert len(buckets) == 9 and order == [7, 6, 5, 4, 3, 2, 1, 0, 8]
        assert 31e6 < 4 * sum(c for _, c in buckets) < 33e6                      # 32.0 MB
        assert shard_batch(192, rank, world) == (24 * rank, 24 * rank + 24)      # cu-net.py:59: contiguous chunks of the global batch
        pg = dist.group.WORLD
        base = (torch.arange(numel, dtype=torch.float32) % 1021.0) - 510.0       # |value| <= 510; times 36 = sum of (rank + 1): < 2^24
        params = base * (1.0 if rank == 0 else -float(rank))
        broadcast_state([params], 0, pg)
        assert torch.equal(params, base)
        grads = base * float(rank + 1)
        red = BucketAllReducer(buckets, pg, overlap=True)
        red.begin_step()
        for b in order:
            red.reduce_bucket(grads, b)
        red.finish(grads)
        assert red.reduced == order and red.world == 8
        total = float(sum(range(1, world + 1)))
        inside = torch.zeros(numel, dtype=torch.bool)
        for b, (begin, count) in enumerate(buckets):
            inside[begin:begin + count] = True
            assert torch.equal(grads[begin:begin + count], base[begin:begin + count] * total), f'bucket {b}'
        assert torch.equal(grads[~inside], (base * float(rank + 1))[~inside])   # alignment padding between buckets: untouched
        q.put((rank, 'ok'))
    except Exception:   # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_eight_rank_config4_bucket_table_gloo():
    """The first time 8 ranks meet must not be the first hardware run (cu-net.py:59 on the 8 GPUs of a node)."""
    res = _run_ranks(_worker8, 8, 300)
    assert sorted(res) == [(r, 'ok') for r in range(8)], res
