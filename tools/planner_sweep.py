"""Sweep planner options around bench.py (tools only): python tools/planner_sweep.py name=v[,v...] -- <bench args>"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index('--')
opts = [a.split('=') for a in sys.argv[1:i]]
names = [o[0] for o in opts]
vals = [[int(v) for v in o[1].split(',')] for o in opts]
for combo in itertools.product(*vals):
    code = ('import sys, runpy; sys.path.insert(0, %r); from cu_net_amd._lib import set_planner_option as s; ' % ROOT
            + ''.join('s(%r, %d); ' % (n, v) for n, v in zip(names, combo))
            + 'sys.argv = [%r] + %r; runpy.run_path(%r, run_name="__main__")' % (os.path.join(ROOT, 'bench.py'), sys.argv[i + 1:], os.path.join(ROOT, 'bench.py')))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if line:
        d = json.loads(line[-1])
        print(dict(zip(names, combo)), d['value'], d['ms_per_step_median'], flush=True)
    else:
        print(dict(zip(names, combo)), 'FAILED', r.stderr[-400:], flush=True)
