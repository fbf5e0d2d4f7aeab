#!/usr/bin/env python3
"""bench.py on the -DCUNET_TUNING library (tools only).  bench.py itself refuses CUNET_LIB_PATH so that a published line can only
come from the shipped library; sweeps of the tuning build's CUNET_* knobs go through this wrapper, whose output is labelled
`library_path: cu_net_amd/libcunet_hip_tuning.so`.  Usage: CUNET_LIB_PATH=.../libcunet_hip_tuning.so CUNET_<KNOB>=v python tools/bench_tuning.py <bench.py flags>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get('CUNET_LIB_PATH'):
    os.environ['CUNET_LIB_PATH'] = os.path.join(ROOT, 'cu_net_amd', 'libcunet_hip_tuning.so')
import cu_net_amd._lib  # noqa: E402  (binds LIB_PATH from the environment)
del os.environ['CUNET_LIB_PATH']
import bench  # noqa: E402
bench.main()
