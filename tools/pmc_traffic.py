"""HBM traffic per launch and kernel class from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <dir of the --pmc FETCH_SIZE run> <dir of the --pmc WRITE_SIZE run> <out.json>

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B, so it is doubled.
WRITE_SIZE is taken as reported (uncalibrated).  Classes are the ones bench.py's `roofline.kernel` names.
"""
import glob
import json
import sys

import pandas as pd

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic_classes import classify      # noqa: E402


def load(d, counter):
    cc = pd.read_csv(glob.glob(d + '/*counter_collection.csv')[0])
    cc = cc[cc.Counter_Name == counter].copy()
    cc['name'] = cc['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    cc['cls'] = cc['name'].map(classify)
    cc = cc[cc.cls.notna()]
    return cc.groupby('cls').Counter_Value.agg(['mean', 'size'])


def step_total(d, counter):
    """Sum of the counter over every kernel of the run / number of steps (one repack_kernel launch per step)."""
    cc = pd.read_csv(glob.glob(d + '/*counter_collection.csv')[0])
    cc = cc[cc.Counter_Name == counter]
    steps = int(cc['Kernel_Name'].str.contains('repack_kernel').sum()) or 1
    return float(cc.Counter_Value.sum()) * 1024.0 / steps, steps


rd = load(sys.argv[1], 'FETCH_SIZE')
wr = load(sys.argv[2], 'WRITE_SIZE')
out = {}
for cls in rd.index:
    f = float(rd.loc[cls, 'mean']) * 1024.0 * 2.0
    w = float(wr.loc[cls, 'mean']) * 1024.0 if cls in wr.index else 0.0
    out[cls] = {'launches_sampled': int(rd.loc[cls, 'size']), 'fetch_bytes_per_launch': round(f),
                'write_bytes_per_launch': round(w), 'hbm_bytes_per_launch': round(f + w)}
meta = {'source': 'rocprofv3 --pmc FETCH_SIZE (KiB, x2 on gfx950) and --pmc WRITE_SIZE (KiB), separate passes, '
                  'bench.py --steps 3 --warmup 2 --no-also', 'workload': sys.argv[4] if len(sys.argv) > 4 else '2,68,24,f32',
        'commit': sys.argv[5] if len(sys.argv) > 5 else 'unknown', 'classes': out}
ft, fs = step_total(sys.argv[1], 'FETCH_SIZE')
wt, wsteps = step_total(sys.argv[2], 'WRITE_SIZE')
meta['step_total'] = {'fetch_bytes_per_step': round(2.0 * ft), 'write_bytes_per_step': round(wt), 'hbm_bytes_per_step': round(2.0 * ft + wt),
                      'steps_sampled': fs, 'note': 'every kernel of the step (FETCH_SIZE x2), incl. warm-up steps'}
print(f"whole step: fetch {2.0 * ft / 1e9:.2f} GB + write {wt / 1e9:.2f} GB = {(2.0 * ft + wt) / 1e9:.2f} GB over {fs} steps sampled")
json.dump(meta, open(sys.argv[3], 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch']):
    print(f"{k:22s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB  per launch ({v['launches_sampled']} launches)")
