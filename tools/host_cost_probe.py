"""Host-side enqueue cost of one fused train step (queue empty at the start): is the CPU ahead of the GPU?"""
import time, torch, cu_net_amd
from cu_net_amd.trainer import FusedTrainer
from oracle.cunet_ref import synthetic_batch
dev = torch.device('cuda', 0)
torch.manual_seed(2)
net = cu_net_amd.create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2).to(dev).train()
tr = FusedTrainer(net)
x, t = synthetic_batch(24, 68, 256, seed=1000)
x, t = x.to(dev), t.to(dev)
for _ in range(5): tr.step(x, t)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print('host enqueue ms/step: min %.3f median %.3f   step (enqueue + drain) ms: median %.3f' % (min(host), sorted(host)[5], sorted(total)[5]))
