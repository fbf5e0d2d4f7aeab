"""Per-kernel summary of a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv in a directory)."""
import sys, glob, pandas as pd
d = sys.argv[1]
cc = pd.read_csv(glob.glob(d + '/*counter_collection.csv')[0])
kt = pd.read_csv(glob.glob(d + '/*kernel_trace.csv')[0])
kt['dur'] = kt['End_Timestamp'] - kt['Start_Timestamp']
piv = cc.pivot_table(index=['Dispatch_Id', 'Kernel_Name', 'Grid_Size'], columns='Counter_Name', values='Counter_Value', aggfunc='sum').reset_index()
piv = piv.merge(kt[['Dispatch_Id', 'dur']], on='Dispatch_Id')
piv['name'] = piv['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
piv = piv[piv.name.str.contains(pat, regex=False)]
cols = [c for c in piv.columns if c.startswith('SQ_') or c.startswith('GRBM') or c.startswith('TCC') or c.startswith('TCP')]
g = piv.groupby(['name', 'Grid_Size']).agg(n=('dur', 'size'), us=('dur', lambda x: x.mean() / 1e3), **{c: (c, 'mean') for c in cols}).reset_index()
g = g.sort_values('us', ascending=False).head(int(sys.argv[3]) if len(sys.argv) > 3 else 8)
pd.set_option('display.width', 250); pd.set_option('display.max_columns', 40)
if 'SQ_WAVE_CYCLES' in g:
    for c in cols:
        if c != 'SQ_WAVE_CYCLES' and c.startswith('SQ_') and 'INSTS' not in c and 'BUSY_CYCLES' not in c:
            g[c + '%'] = (100 * g[c] / g['SQ_WAVE_CYCLES']).round(1)
print(g.to_string())
