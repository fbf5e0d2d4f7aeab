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
    # Reference parity first: under `-x` one miss in a hand-calibrated kernel-level test must not hide the tests that compare
    # with the reference's own vectors (round 4 lost all of test_gpu_parity / test_gpu_quant behind one node test).
    order = ['test_gpu_parity', 'test_gpu_quant', 'test_gpu_configs', 'test_gpu_dp', 'test_gpu_augment', 'test_gpu_exact', 'test_gpu_nodes']
    rank = {name: i for i, name in enumerate(order)}
    items.sort(key=lambda it: rank.get(os.path.splitext(os.path.basename(str(it.fspath)))[0], len(order)))      # (stable: order inside a file is kept)
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


@pytest.fixture(autouse=True)
def _planner_options_are_restored():
    """Every test starts from -- and leaves behind -- the planner options in effect when it began (the library's defaults, or what
    CUNET_TEST_PLANNER_OPTS selected for the whole session): a test that switches `f32_split`, `wgrad3_min_rows`, ... cannot leak its
    selection into the tests behind it, whatever it restores by hand."""
    from cu_net_amd import _lib
    try:
        before = _lib.planner_options_snapshot()
    except Exception:      # library not built: the CPU tests that need it fail on their own
        yield
        return
    yield
    for k, v in before.items():
        if _lib.get_planner_option(k) != v:
            _lib.set_planner_option(k, v)
