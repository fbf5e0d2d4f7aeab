"""Timeline view of a rocprofv3 kernel trace: per-step wall time, busy time (union), sum of kernel time
per stream, and the top kernels.  Usage: trace_timeline.py <kernel_trace.csv> [nsteps_to_skip]"""
import sys
import pandas as pd
k = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp')
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
# steps are delimited by the repack kernel (first kernel of every forward)
starts = k.index[k.name == 'repack_kernel'].tolist()
pos = {idx: i for i, idx in enumerate(k.index)}
rows = k.reset_index(drop=True)
st = [pos[i] for i in starts]
if len(st) < 3:
    raise SystemExit('need >= 3 steps')
a, b = st[-2], st[-1]          # last complete step
step = rows.iloc[a:b]
t0, t1 = step.Start_Timestamp.min(), step.End_Timestamp.max()
print(f'step wall {(t1 - t0) / 1e3:.1f} us, kernels {len(step)}')
# union busy
iv = sorted(zip(step.Start_Timestamp, step.End_Timestamp))
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'union busy {busy / 1e3:.1f} us, idle gaps {(t1 - t0 - busy) / 1e3:.1f} us, sum of kernel durations {(step.End_Timestamp - step.Start_Timestamp).sum() / 1e3:.1f} us')
for q, g in step.groupby('Stream_Id' if 'Stream_Id' in step.columns else 'Queue_Id'):
    print(f'  stream {q}: {len(g)} kernels, sum {((g.End_Timestamp - g.Start_Timestamp).sum()) / 1e3:.1f} us')
step = step.assign(dur=(step.End_Timestamp - step.Start_Timestamp) / 1e3)
g = step.groupby('name').agg(n=('dur', 'size'), tot=('dur', 'sum'), avg=('dur', 'mean')).sort_values('tot', ascending=False)
print(g.head(25).to_string())
