"""MFMA pipe utilisation per kernel class from one rocprofv3 SQ-counter pass (side stream off: every kernel alone on the GPU).

usage: pmc_mfma_busy.py <dir of the --pmc SQ_... run> <out.json> <workload key> <commit>

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs)

SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles in which the matrix pipe is executing (checked on wgrad3_stem_kernel: 3 932 160
v_mfma_f32_32x32x2_f32 x 64 cycles = 251 658 240, the value the counter reports); 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
are the denominator the 157.3 TFLOP/s fp32 / 2.5 PFLOP/s bf16 peaks are quoted on (MI355X_MICROARCH.md), so a chip that runs
below 2.4 GHz under the profiler shows as idle pipe, as it does in the flop-based fraction.  Also reduced per class: the share
of wave cycles parked on s_waitcnt (SQ_WAIT_ANY) and waiting to issue (SQ_WAIT_INST_ANY).  bench.py copies the entry of its
roofline class into `roofline.mfma_busy`.
"""
import glob
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic_classes import classify      # noqa: E402

CLOCK_HZ, SIMDS = 2.4e9, 1024


def main():
    d, out_path, workload, commit = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else 'unknown')
    cc = pd.read_csv(glob.glob(d + '/*counter_collection.csv')[0])
    kt = pd.read_csv(glob.glob(d + '/*kernel_trace.csv')[0])
    kt['dur_ns'] = kt['End_Timestamp'] - kt['Start_Timestamp']
    piv = cc.pivot_table(index=['Dispatch_Id', 'Kernel_Name'], columns='Counter_Name', values='Counter_Value', aggfunc='sum').reset_index()
    piv = piv.merge(kt[['Dispatch_Id', 'dur_ns']], on='Dispatch_Id')
    piv['name'] = piv['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    piv['cls'] = piv['name'].map(classify)
    piv = piv[piv.cls.notna()]
    classes = {}
    for cls, g in piv.groupby('cls'):
        busy = float(g['SQ_VALU_MFMA_BUSY_CYCLES'].sum())
        dur_s = float(g['dur_ns'].sum()) * 1e-9
        wave = float(g['SQ_WAVE_CYCLES'].sum()) if 'SQ_WAVE_CYCLES' in g else 0.0
        ent = {'mfma_busy': round(busy / (dur_s * CLOCK_HZ * SIMDS), 4) if dur_s > 0 else None,
               'launches_sampled': int(len(g)), 'avg_launch_us': round(1e6 * dur_s / len(g), 2)}
        if wave > 0:
            if 'SQ_WAIT_ANY' in g:
                ent['wave_cycles_on_waitcnt'] = round(float(g['SQ_WAIT_ANY'].sum()) / wave, 4)
            if 'SQ_WAIT_INST_ANY' in g:
                ent['wave_cycles_waiting_to_issue'] = round(float(g['SQ_WAIT_INST_ANY'].sum()) / wave, 4)
        classes[cls] = ent
    meta = {'source': 'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES '
                      '... --kernel-trace, CUNET_NO_SIDE_STREAM=1, bench.py --steps 2 --warmup 1 --no-also',
            'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (sum of launch durations x 2.4e9 Hz x 1024 SIMDs), per class',
            'workload': workload, 'commit': commit, 'classes': classes}
    json.dump(meta, open(out_path, 'w'), indent=1)
    for k, v in sorted(classes.items(), key=lambda kv: -(kv[1]['mfma_busy'] or 0)):
        print(f"{k:26s} mfma_busy {100 * (v['mfma_busy'] or 0):5.1f} %  launches {v['launches_sampled']:4d}  avg {v['avg_launch_us']:8.2f} us  "
              f"waitcnt {100 * v.get('wave_cycles_on_waitcnt', 0):5.1f} %  issue-wait {100 * v.get('wave_cycles_waiting_to_issue', 0):5.1f} %")


if __name__ == '__main__':
    main()
