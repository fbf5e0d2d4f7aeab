import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
k = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.slice(0, 48)
print(k.columns.tolist())
idx = k.index[k['name'].str.contains('copyBuffer')].tolist()
print(len(idx), 'copyBuffer launches; streams/queues:', k.loc[idx, ['Queue_Id', 'Stream_Id']].drop_duplicates().to_dict('records') if 'Stream_Id' in k else '')
# neighbours in the same queue
from collections import Counter
prev, nxt = Counter(), Counter()
for i in idx[len(idx)//2: len(idx)//2 + 60]:
    q = k.loc[i, 'Queue_Id']
    same = k[(k['Queue_Id'] == q)]
    pos = same.index.get_loc(i)
    p = same.iloc[pos - 1]['name'] if pos > 0 else '-'
    n = same.iloc[pos + 1]['name'] if pos + 1 < len(same) else '-'
    prev[p] += 1; nxt[n] += 1
print('previous kernel in the same queue:', prev.most_common(8))
print('next kernel in the same queue:', nxt.most_common(8))
print(k.groupby('Queue_Id').size())
