"""One rank of the two-rank data-parallel step of tests/test_gpu_dp.py (launched by torch.distributed.run).

RCCL ("nccl") when every rank has its own GPU; on a one-GPU box both ranks share cuda:0 and the gradient buckets
travel over gloo staged through the host -- the step, the bucket callbacks and the stream joins are the same code."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    own_gpu = ndev >= world
    dev = torch.device('cuda', local if own_gpu else 0)
    torch.cuda.set_device(dev)
    if own_gpu:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import cu_net_amd
    from cu_net_amd.parallel import shard_batch
    from cu_net_amd.trainer import FusedTrainer
    for kv in filter(None, os.environ.get('CUNET_TEST_PLANNER_OPTS', '').split(',')):      # (the suite's kernel selection, see conftest.py)
        from cu_net_amd._lib import set_planner_option
        set_planner_option(kv.split('=')[0].strip(), int(kv.split('=')[1]))
    from oracle import cunet_ref as O
    from tests.test_gpu_dp import CASES, make_inputs
    case = os.environ.get('CUNET_DP_CASE', 'toy')
    CFG, GLOBAL_BATCH, HW = CASES[case]
    spec = O.Spec(**CFG)
    st = O.init_state(spec, seed=61 + 7 * rank)           # rank 1 starts from other parameters: the broadcast must fix that
    net = cu_net_amd.create_cu_net(**CFG)
    net.load_state_dict(st)
    net = net.to(dev).train()
    tr = FusedTrainer(net, process_group=dist.group.WORLD, overlap=True)
    tr.broadcast_parameters(0)
    p0 = net._param_arena.detach().cpu().clone()
    x, t = make_inputs(case)
    lo, hi = shard_batch(GLOBAL_BATCH, rank, world)
    loss = tr.step(x[lo:hi].to(dev), t[lo:hi].to(dev))
    torch.cuda.synchronize(dev)
    plan = net._get_plan(hi - lo, HW, HW, True)
    seen = torch.ones(1, device=dev if own_gpu else 'cpu')
    dist.all_reduce(seen)
    torch.save({'loss': float(loss), 'grads': net._grad_arena.detach().cpu(), 'params': net._param_arena.detach().cpu(),
                'p0': p0, 'reduced': list(tr.reducer.reduced), 'order': plan.handle.bucket_order(), 'backend': dist.get_backend(),
                'ranks_seen': int(seen.item()), 'buffers': net._buffer_arena.detach().cpu()},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
