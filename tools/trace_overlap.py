"""How well do the two streams of a step overlap?  From a rocprofv3 kernel trace (CSV):
  * per stream of the LAST complete step: kernels, sum of durations, sum of gaps between consecutive kernels, span;
  * the caller's stream (the one that runs repack_kernel): gaps grouped by the kernel that FOLLOWS the gap, and how much of
    each main-stream kernel's duration had a side-stream kernel running next to it;
  * with a second trace taken with CUNET_NO_SIDE_STREAM=1: per (kernel, grid) the ratio of the overlapped to the alone duration.
Usage: trace_overlap.py <overlapped kernel_trace.csv> [<serial kernel_trace.csv>]"""
import sys

import pandas as pd


def load(path):
    k = pd.read_csv(path).sort_values('Start_Timestamp').reset_index(drop=True)
    k['name'] = k['Kernel_Name'].str.replace('cunet::', '').str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    k['dur'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
    return k


def last_step(k):
    st = k.index[k.name == 'repack_kernel'].tolist()
    if len(st) < 3:
        raise SystemExit('need >= 3 steps')
    return k.iloc[st[-2]:st[-1]].copy()


def main():
    k = load(sys.argv[1])
    step = last_step(k)
    qcol = 'Stream_Id' if 'Stream_Id' in step.columns and step['Stream_Id'].nunique() > 1 else 'Queue_Id'
    t0, t1 = step.Start_Timestamp.min(), step.End_Timestamp.max()
    print(f'step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels, streams by {qcol}')
    main_q = step[step.name == 'repack_kernel'][qcol].iloc[0]
    streams = {}
    for q, g in step.groupby(qcol):
        g = g.sort_values('Start_Timestamp')
        gaps = (g.Start_Timestamp.values[1:] - g.End_Timestamp.values[:-1]) / 1e3
        streams[q] = g
        print(f'  stream {q}{" (caller)" if q == main_q else ""}: {len(g)} kernels, sum of durations {g.dur.sum():.1f} us, '
              f'sum of positive gaps {gaps[gaps > 0].sum():.1f} us, span {(g.End_Timestamp.max() - g.Start_Timestamp.min()) / 1e3:.1f} us')
    m = streams[main_q].sort_values('Start_Timestamp').reset_index(drop=True)
    others = step[step[qcol] != main_q]
    gap = [0.0] + list((m.Start_Timestamp.values[1:] - m.End_Timestamp.values[:-1]) / 1e3)
    m['gap_before'] = gap
    # share of each main kernel's duration during which some other-stream kernel was running
    iv = sorted(zip(others.Start_Timestamp, others.End_Timestamp))
    merged = []
    for s, e in iv:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [a for a, _ in merged]
    cov = []
    for s, e in zip(m.Start_Timestamp, m.End_Timestamp):
        i = max(0, bisect.bisect_right(starts, s) - 1)
        c = 0
        while i < len(merged) and merged[i][0] < e:
            c += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        cov.append(c / max(e - s, 1))
    m['covered'] = cov
    m['grid'] = m.Grid_Size_X
    g = m.groupby(['name', 'grid']).agg(n=('dur', 'size'), dur=('dur', 'sum'), avg=('dur', 'mean'), gap=('gap_before', 'sum'),
                                        gap_avg=('gap_before', 'mean'), covered=('covered', 'mean')).sort_values('dur', ascending=False)
    print('\ncaller stream, by (kernel, grid): n, sum / avg duration, sum / avg gap in front of it, share of its duration with a side kernel running')
    print(g.head(60).to_string())
    print(f'\ncaller stream total: durations {m.dur.sum():.1f} us + gaps {m.gap_before[m.gap_before > 0].sum():.1f} us; '
          f'duration-weighted co-running share {float((m.covered * m.dur).sum() / m.dur.sum()):.2f}')
    small = m[m.dur < 20]
    print(f'kernels shorter than 20 us on the caller stream: {len(small)}, durations {small.dur.sum():.1f} us, gaps in front {small.gap_before.sum():.1f} us')
    if len(sys.argv) > 2:
        s = last_step(load(sys.argv[2]))
        a = s.groupby(['name', 'Grid_Size_X']).dur.mean()
        o = step.groupby(['name', 'Grid_Size_X']).agg(n=('dur', 'size'), avg=('dur', 'mean'), tot=('dur', 'sum'))
        o['alone'] = [a.get(ix, float('nan')) for ix in o.index]
        o['ratio'] = o.avg / o.alone
        o['excess_us'] = o.tot - o.alone * o.n
        print('\noverlapped vs alone, by (kernel, grid): n, avg overlapped, alone, ratio, excess per step')
        print(o.sort_values('excess_us', ascending=False).head(50).to_string())
        print(f'sum of kernel durations: overlapped {step.dur.sum():.1f} us, alone {s.dur.sum():.1f} us')


if __name__ == '__main__':
    main()
