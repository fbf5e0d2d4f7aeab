"""Every kernel of the last complete train step on the GPU's time line (rocprofv3 --kernel-trace CSV): start offset from the first kernel of
the step, duration, the idle time in front of it on ITS stream, how many kernels of the other stream were running when it started, stream,
name, grid.  Usage: step_timeline.py <dir with *kernel_trace.csv> [step index from the end, default 2]"""
import glob
import sys
import pandas as pd
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
k = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.slice(0, 46)
st = k.index[k.name == 'repack_kernel'].tolist()
a, b = st[-back], st[-back + 1] if back > 1 else len(k)
step = k.iloc[a:b]
col = 'Stream_Id' if 'Stream_Id' in step.columns else 'Queue_Id'
t0 = step.Start_Timestamp.min()
print(f'step wall {(step.End_Timestamp.max() - t0) / 1e3:.1f} us, {len(step)} kernels; columns: start us, duration us, gap in front on its stream us, stream, name, grid, workgroup')
last_end = {}
ends = {q: [] for q in step[col].unique()}
for _, r in step.iterrows():
    q = r[col]
    gap = (r.Start_Timestamp - last_end[q]) / 1e3 if q in last_end else 0.0
    other = sum(1 for qq, lst in ends.items() if qq != q for (s, e) in lst if s <= r.Start_Timestamp < e)
    print(f'{(r.Start_Timestamp - t0) / 1e3:9.1f} {(r.End_Timestamp - r.Start_Timestamp) / 1e3:8.1f} {gap:7.1f} {"*" if other else " "} s{q} {r["name"]:46s} {r.Grid_Size_X:8d} {r.Workgroup_Size_X if "Workgroup_Size_X" in r else 0}')
    last_end[q] = max(last_end.get(q, 0), r.End_Timestamp)
    ends[q].append((r.Start_Timestamp, r.End_Timestamp))
