#!/bin/bash
# Round-3 GPU call 12: the whole GPU suite on the current tree, then the default bench line.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=10 > $OUT/r03c12_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/r03c12_pytest.txt )
tail -20 $OUT/r03c12_pytest.txt
timeout 600 python bench.py > $OUT/r03c12_bench.json 2> $OUT/r03c12_bench.err
python -c "
import json
d=json.load(open('$OUT/r03c12_bench.json'))
print('value', d['value'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline'].get('alone',{}).get('frac'))
for a in d['also']: print(a['workload'][:100], a.get('value'), a.get('ms_per_step_median'))
"
