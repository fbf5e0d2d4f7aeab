"""GPU diagnostic: per-tensor gradient error of the HIP path AND of torch-fp32-CPU, both measured
against an fp64 CPU execution of the same plan.  Tells precision problems from conditioning."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cu_net_amd
from cu_net_amd.trainer import FusedTrainer
from tests._golden import Golden
from tests._plan_interp import run_plan

tag = sys.argv[1] if len(sys.argv) > 1 else 'G9_L2_o1_c32'
g = Golden(tag)
x, target = g.t('x'), g.t('target')
net = cu_net_amd.create_cu_net(**g.cfg)
net.load_state_dict(g.group('state0'))
net = net.cuda().train()
tr = FusedTrainer(net)
n, _, h, w = x.shape
plan = net._get_plan(n, h, w, True)
desc = plan.handle.describe()

def ref(dtype):
    st = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in g.group('state0').items()}
    for k in st:
        if st[k].is_floating_point() and 'running' not in k:
            st[k].requires_grad_(True)
    outs, acts, grads, loss = run_plan(desc, st, x.to(dtype), True, True, target.to(dtype))
    pg = {k: v.grad for k, v in st.items() if v.is_floating_point() and v.grad is not None}
    return acts, grads, pg

a64, g64, p64 = ref(torch.float64)
a32, g32, p32 = ref(torch.float32)
tr.step(x.cuda(), target.cuda())
torch.cuda.synchronize()

def rel(a, b):
    return ((a.double() - b).abs().max() / (b.abs().max() + 1e-300)).item()

print(f'{"tensor":60s} {"hip_vs_f64":>11s} {"cpu32_vs_f64":>12s}')
for t in desc['tensors']:
    nm = t['name']
    print(f'act  {nm:55s} {rel(plan.debug_tensor(nm).cpu(), a64[nm]):11.2e} {rel(a32[nm], a64[nm]):12.2e}')
for t in reversed(desc['tensors']):
    nm = t['name']
    if nm in g64:
        print(f'grad {nm:55s} {rel(plan.debug_tensor(nm, grad=True).cpu(), g64[nm]):11.2e} {rel(g32[nm], g64[nm]):12.2e}')
off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
for k, v in p64.items():
    o, nmel, shape = off[k]
    print(f'dpar {k:55s} {rel(net._grad_arena[o:o + nmel].view(shape).cpu(), v):11.2e} {rel(p32[k], v):12.2e}')
