"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launches / avg / min / total us."""
import sys
import pandas as pd
k = pd.read_csv(sys.argv[1])
k['dur'] = (k['End_Timestamp'] - k['Start_Timestamp']) / 1e3
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
g = k.groupby(['name', 'Grid_Size_X', 'Grid_Size_Y', 'Workgroup_Size_X']).agg(n=('dur', 'size'), avg=('dur', 'mean'), mn=('dur', 'min'), tot=('dur', 'sum')).reset_index()
g = g.sort_values('tot', ascending=False)
pd.set_option('display.width', 220)
print(g.head(int(sys.argv[2]) if len(sys.argv) > 2 else 45).to_string())
print('total kernel us', k['dur'].sum())
