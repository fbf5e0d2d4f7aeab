"""The end of a train step on the GPU's time line (rocprofv3 --kernel-trace CSV): the last kernels of the last complete step with their start
offset from the end of the step, stream, duration -- where the caller's stream waits for the side stream's weight gradients, the stem's
backward, the bucket reduces and the optimiser.  Usage: step_tail.py <dir with *kernel_trace.csv> [how many kernels]"""
import glob
import sys
import pandas as pd
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
k = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.slice(0, 44)
st = k.index[k.name == 'repack_kernel'].tolist()
a, b = st[-2], st[-1]
step = k.iloc[a:b]
t1 = step.End_Timestamp.max()
t0 = step.Start_Timestamp.min()
print(f'step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels')
col = 'Stream_Id' if 'Stream_Id' in step.columns else 'Queue_Id'
tail = step.sort_values('End_Timestamp').tail(n).sort_values('Start_Timestamp')
for _, r in tail.iterrows():
    print(f'  start {-(t1 - r.Start_Timestamp) / 1e3:8.1f} us  end {-(t1 - r.End_Timestamp) / 1e3:8.1f} us  stream {r[col]}  {(r.End_Timestamp - r.Start_Timestamp) / 1e3:7.1f} us  {r["name"]}  grid {r.Grid_Size_X}')
for q, g in step.groupby(col):
    print(f'stream {q}: first start {(g.Start_Timestamp.min() - t0) / 1e3:.1f} us, last end {-(t1 - g.End_Timestamp.max()) / 1e3:.1f} us before the end of the step')
