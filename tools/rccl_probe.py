"""RCCL micro-probe (run under torch.distributed.run): cost of a bucket-sized all_reduce alone and next to a matmul."""
import os, time, torch, torch.distributed as dist
rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
g = torch.zeros(1_000_000, device=dev)
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
side = torch.cuda.Stream()
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def ar(): dist.all_reduce(g)
def mm(): torch.mm(a, b)
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): dist.all_reduce(g)
    torch.mm(a, b); torch.mm(a, b)
    torch.cuda.current_stream().wait_stream(side)
def two_mm(): torch.mm(a, b); torch.mm(a, b)
if rank == 0:
    print(f'world {world}: all_reduce(4MB) {t(ar):.3f} ms   mm {t(mm):.3f} ms   2mm {t(two_mm):.3f} ms   2mm || all_reduce {t(both):.3f} ms', flush=True)
else:
    t(ar); t(mm); t(two_mm); t(both)
dist.destroy_process_group()
