import sys, pandas as pd
k = pd.read_csv(sys.argv[1]); pat = sys.argv[2]
k['dur'] = (k['End_Timestamp'] - k['Start_Timestamp']) / 1e3
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
k = k[k.name.str.contains(pat, regex=False)]
g = k.groupby(['name', 'Grid_Size_X', 'Grid_Size_Y', 'Workgroup_Size_X']).agg(n=('dur', 'size'), avg=('dur', 'mean'), mn=('dur', 'min')).reset_index()
print(g.sort_values('avg', ascending=False).head(int(sys.argv[3]) if len(sys.argv) > 3 else 6).to_string())
