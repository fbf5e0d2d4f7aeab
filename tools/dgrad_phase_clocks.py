#!/usr/bin/env python3
"""Where a wave of the fp32 1x1 data gradient spends its cycles (tools only; the -DCUNET_TUNING library with CUNET_CONV_DBG=512).

    CUNET_LIB_PATH=.../libcunet_hip_tuning.so python tools/dgrad_phase_clocks.py [--serial]

Runs BASELINE config 2 (CU-Net-2, K = 68, bs 24) for a few steps with s_memtime stamps around the phases of conv_body's PF2 tile
loop and prints shader cycles per wave and tile: requests + MFMA issue, x pieces -> LDS tile (waits for the x loads), column pass (waits
for the MFMA chain), LDS fp64 atomics, dz out; plus block set-up and the whole kernel per wave.  --serial: weight gradients on the
caller's stream (every kernel alone on the GPU)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('CUNET_LIB_PATH', os.path.join(ROOT, 'cu_net_amd', 'libcunet_hip_tuning.so'))
os.environ['CUNET_CONV_DBG'] = '512'
if '--serial' in sys.argv:
    os.environ['CUNET_NO_SIDE_STREAM'] = '1'
import torch  # noqa: E402
import cu_net_amd  # noqa: E402
from cu_net_amd import _lib  # noqa: E402
from cu_net_amd.trainer import FusedTrainer  # noqa: E402
import bench  # noqa: E402

L = _lib.lib()
fn = L.cunet_tuning_conv_phase
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device('cuda', 0)
torch.manual_seed(2)
net = cu_net_amd.create_cu_net(4, 32, 128, 68, 2, 1, 2).to(dev).train()
tr = FusedTrainer(net)
x, t = bench.synthetic_batch(24, 68, 256, seed=1000, device=dev)
for _ in range(3):
    tr.step(x, t)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
assert fn(buf, 1) == 0
steps = 5
for _ in range(steps):
    tr.step(x, t)
torch.cuda.synchronize()
assert fn(buf, 1) == 0
v = [int(b) for b in buf]
tiles = max(v[0], 1)
names = ['requests + MFMA issue', 'x pieces -> LDS tile (x loads land)', 'column pass (MFMA chain lands)', 'LDS fp64 atomics', 'dz pieces out']
print(f'{steps} steps, {v[0]} wave-tiles of the 1x1 data gradient ({v[0] // steps} per step)')
tot = sum(v[1:6])
for n, c in zip(names, v[1:6]):
    print(f'  {n:40s} {c / tiles:9.0f} cycles per tile  {100.0 * c / tot:5.1f} %')
print(f'  {"sum of the phases":40s} {tot / tiles:9.0f} cycles per tile (64 MFMAs = 4096 matrix-pipe cycles)')
print(f'  block set-up {v[6]} / whole kernel {v[7]} wave-cycles summed over the first lanes = {100.0 * v[6] / max(v[7], 1):.1f} % of a wave\'s life; '
      f'tile loop {100.0 * tot / max(v[7], 1):.1f} %')
