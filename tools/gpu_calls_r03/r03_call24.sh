#!/bin/bash
# Round-3 GPU call 24: the default bench line at HEAD, then the profile recipe.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/r03c24_bench.json 2> $OUT/r03c24_bench.err
python -c "
import json
d=json.load(open('$OUT/r03c24_bench.json'))
r=d['roofline']
print('value', d['value'], d['ms_per_step_median'], r['kernel'], r['frac'], r.get('alone',{}).get('frac'), r.get('largest_side_stream_class'))
for a in d['also']: print(a['workload'][:100], a.get('value'), a.get('roofline',{}).get('kernel'), a.get('roofline',{}).get('frac'))
"
bash tools/profile_r03.sh $1 > $OUT/r03c24_profile.log 2>&1
tail -4 $OUT/r03c24_profile.log | cut -c1-200
