import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped automatically when no GPU is visible (so a plain `pytest tests/`
    works in the CPU container); `-m gpu` on a box without a GPU therefore skips, not fails."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def pytest_sessionstart(session):
    """CUNET_TEST_PLANNER_OPTS="name=value,..." applies cunet_set_planner_option before any plan exists: the whole GPU suite can be run
    on a non-default kernel selection (e.g. f32_split=1: the convolutions contract on the bf16 matrix pipe)."""
    opts = os.environ.get('CUNET_TEST_PLANNER_OPTS', '')
    if not opts:
        return
    from cu_net_amd._lib import set_planner_option
    for kv in opts.split(','):
        name, val = kv.split('=')
        set_planner_option(name.strip(), int(val))
