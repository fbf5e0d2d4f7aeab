#!/usr/bin/env python3
"""Where a wave of a row-ring kernel spends its cycles (tools only; the -DCUNET_TUNING library, CUNET_CONV_DBG = 4096 / 8192).

    CUNET_LIB_PATH=.../libcunet_hip_tuning.so python tools/ring_phase_clocks.py [--fwd] [--serial]

--rows: dgrad1x1_rows_split2_kernel (the 1x1 data gradient at 64 x 64: counted wait, barrier, requests, cut, MFMAs, epilogue).
Default: dgrad3x3_ring_split_kernel (the 3x3 data gradient); --fwd: conv3x3_ring_split_kernel (the 3x3 forward: requests, MFMAs, partial
tiles to LDS + barrier, sum of the eight partial tiles + store + statistics, barrier, ring commit + barrier).

Runs BASELINE config 2 (CU-Net-2, K = 68, bs 24) for a few steps with s_memtime stamps around the phases of
dgrad3x3_ring_split_kernel's row loop and prints shader cycles per wave and 32-pixel tile: requests (next dY row, next tile's x pieces),
MFMAs (LDS fragment reads + the chain), partial hand-over + first barrier, ring commit (+ partial add), epilogue (first tap group's
waves; the second group's wait shows up in the second barrier), second barrier.  60 / 48 MFMAs of 32 cycles per wave and tile are
1920 / 1536 matrix-pipe cycles.  --serial: weight gradients on the caller's stream (every kernel alone on the GPU).
(Round 4 ended with this kernel at 39 us per launch against an HBM floor of ~23 us at 64 x 64 and no GPU budget left to run this.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('CUNET_LIB_PATH', os.path.join(ROOT, 'cu_net_amd', 'libcunet_hip_tuning.so'))
FWD = '--fwd' in sys.argv
ROWS = '--rows' in sys.argv      # dgrad1x1_rows_split2_kernel (round 5)
os.environ['CUNET_CONV_DBG'] = '16384' if ROWS else ('8192' if FWD else '4096')
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
if ROWS:
    names = ['counted wait (all but the previous dz stores)', 'barrier', 'x -> LDS tile, LDS-DMA + x requests', 'cut of the next tile (3 bf16 planes)',
             'MFMAs (plane reads + chain of 48)', 'epilogue + dz stores']
    print(f'{steps} steps, {v[0]} wave-tiles of the 1x1 row-tile data gradient ({v[0] // steps} per step)')
    tot = sum(v[1:7])
    for n, c in zip(names, v[1:7]):
        print(f'  {n:48s} {c / tiles:9.0f} cycles per wave and tile  {100.0 * c / max(tot, 1):5.1f} %')
    # (round 6: [7] runs from the wave's first instruction to the end of its tile loop and is stamped BEFORE the wave queues its eight
    # same-address atomics -- round 5 stamped it behind them, and ~10 k serialised atomics per launch inflated the "whole kernel" figure)
    print(f'  {"sum of the phases":48s} {tot / tiles:9.0f} cycles per wave and tile')
    print(f'  {"set-up (tables, weight slice cut, first tiles)":48s} {(v[7] - tot) / tiles:9.0f} cycles per wave and tile = {100.0 * (v[7] - tot) / max(v[7], 1):.1f} % of the wave\'s life ({v[7] / tiles:.0f})')
    sys.exit(0)
names = (['requests (row g + 3)', 'MFMAs (fragment reads + chain)', 'partial tiles to LDS + barrier 1', 'sum of partials + store + statistics', 'barrier 2',
          'ring commit + barrier 3'] if FWD else
         ['requests (dY row g + 3, next x pieces)', 'MFMAs (fragment reads + chain)', 'partial hand-over + barrier 1', 'ring commit (+ partial add)',
          'epilogue (first tap group)', 'barrier 2'])
print(f'{steps} steps, {v[0]} wave-tiles of the 3x3 {"forward" if FWD else "data gradient"} on the row ring ({v[0] // steps} per step)')
tot = sum(v[1:7])
for n, c in zip(names, v[1:7]):
    print(f'  {n:42s} {c / tiles:9.0f} cycles per wave and tile  {100.0 * c / max(tot, 1):5.1f} %')
print(f'  {"sum of the phases":42s} {tot / tiles:9.0f} cycles per wave and tile; whole kernel {v[7] / tiles:.0f} (set-up: weights, tables, three rows)')
