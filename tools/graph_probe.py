"""Does capturing the fused train step in a HIP graph shorten it? (one GPU, no process group)"""
import time, torch, cu_net_amd
from cu_net_amd.trainer import FusedTrainer
from oracle.cunet_ref import synthetic_batch
dev = torch.device('cuda', 0)
torch.manual_seed(2)
net = cu_net_amd.create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2).to(dev).train()
tr = FusedTrainer(net)
x, t = synthetic_batch(24, 68, 256, seed=1000)
x, t = x.to(dev), t.to(dev)
for _ in range(5): loss = tr.step(x, t)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager  ms/step', timeit(lambda: tr.step(x, t)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.step(x, t)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = tr.step(x, t)
print('graph  ms/step', timeit(g.replay), 'loss', float(loss), flush=True)
