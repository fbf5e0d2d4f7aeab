"""CPU: the C++ plan builder (cu_net_amd/csrc/plan.cpp, reached through the C ABI) wires the network
exactly like the reference: executing its node list with torch ops reproduces the golden outputs."""
import os

import pytest
import torch

from cu_net_amd._lib import CUNetError, PlanHandle
from oracle import cunet_ref as O
from tests._golden import TINY, Golden
from tests._plan_interp import run_plan


@pytest.mark.parametrize('tag', TINY)
def test_plan_reproduces_golden(tag):
    g = Golden(tag)
    x, target = g.t('x'), g.t('target')
    plan = PlanHandle(**g.cfg, batch=x.shape[0], height=x.shape[2], width=x.shape[3])
    st = g.group('state0')
    ents = plan.state_entries()
    assert [e[0] for e in ents] == list(st.keys())
    for name, kind, shape, off, numel in ents:
        assert tuple(st[name].shape) == shape
    for k in st:
        if st[k].is_floating_point() and 'running' not in k:
            st[k].requires_grad_(True)
    outs, acts, grads, loss = run_plan(plan.describe(), st, x, True, True, target)
    for a, b in zip(outs, g.list('out')):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.detach(), g.t('loss'), rtol=1e-5, atol=1e-6)
    gg = g.group('grad')
    for k, v in gg.items():
        err = (st[k].grad - v).abs().max().item()
        assert err <= 1e-4 * v.abs().max().item() + 1e-7, (k, err)
    assert plan.anchors() == O.Spec(**g.cfg).loss_anchors


def test_plan_layout_full_size():
    plan = PlanHandle(4, 32, 128, 68, 2, 1, 2, 24, 256, 256)
    assert plan.param_numel == 1936512
    ents = plan.state_entries()
    assert len(ents) == 396
    spec = O.Spec(4, 32, 128, 68, 2, 1, 2)
    assert [(e[0], e[2]) for e in ents] == [(n, tuple(s)) for n, s, _ in O.state_entries(spec)]
    d = plan.describe()
    # every concat of order 1 has at most 4 segments; r=64 adapters of U-Net 1 see 128+32+32 channels
    node = [n for n in d['nodes'] if n['name'] == 'hg.down_blocks.0.adapters_ahead.1.adapter_conv'][0]
    assert sum(d['tensors'][s['t']]['C'] for s in node['segs']) == 192
    up = [n for n in d['nodes'] if n['name'] == 'hg.up_blocks.3.layers.1.conv1'][0]
    assert [s['ups'] for s in up['segs']] == [1, 0, 0]


def test_plan_errors():
    with pytest.raises(CUNetError):
        PlanHandle(4, 32, 128, 16, 2, 2, 2, 1, 256, 256)     # order >= layer_num (cu_net.py:285-287)
    with pytest.raises(CUNetError):
        PlanHandle(4, 32, 128, 16, 2, 1, 3, 1, 256, 256)     # loss_num > layer_num (cu_net.py:274)
    with pytest.raises(CUNetError):
        PlanHandle(4, 32, 128, 16, 2, 1, 2, 1, 200, 256)     # H not a multiple of 64
    with pytest.raises(CUNetError):
        PlanHandle(4, 6, 128, 16, 2, 1, 2, 1, 256, 256)      # growth not a multiple of 4


@pytest.mark.parametrize('cfg,n,h,w', [
    (dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2), 24, 256, 256),
    (dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=8, order=1, loss_num=8), 24, 256, 256),
    (dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2), 5, 256, 128),
    (dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2), 1, 256, 256),
])
def test_weight_gradient_partial_tile_plan(cfg, n, h, w):
    """Host-side invariants of the partial-tile weight gradients (wgrad3 family, plan.cpp): every split count covers the node's rows,
    the bf16 split policy never needs more partials than the fp32 one (they share the region), 1x1 ranges are whole chunks of both
    kernels, the bf16 3x3 ring walks rows in pairs, the stem's workgroups cover every output row of an image and fit the LDS, and the
    partial regions of a gradient bucket do not overlap (the region itself is reused bucket after bucket)."""
    from cu_net_amd._lib import PlanHandle, set_planner_option
    set_planner_option('fuse_wgrad', 1)              # (the option that asks for the larger partial slices; default 0)
    try:
        ph = PlanHandle(cfg['neck_size'], cfg['growth_rate'], cfg['init_chan_num'], cfg['class_num'], cfg['layer_num'], cfg['order'],
                        cfg['loss_num'], n, h, w)
    finally:
        set_planner_option('fuse_wgrad', 0)
    d = ph.describe()
    T = d['tensors']
    per_bucket = {}
    seen = 0
    for nd in d['nodes']:
        if nd.get('wg3', 0) <= 0:
            continue
        seen += 1
        o = T[nd['out']]
        S, rows, S16, rows16 = nd['wg3'], nd['wg3_rows'], nd['wg3_bf16'], nd['wg3_rows_bf16']
        assert 1 <= S16 <= S
        if nd['op'] == 'stem_conv':
            wpi = nd['wg3_wpi']
            assert S == o['N'] * wpi and wpi * rows >= o['H'] and (wpi - 1) * rows < o['H']
            cp = w + 6
            cp += (17 - cp % 32 + 32) % 32
            rp = 3 * cp
            rp += (7 - rp % 32 + 32) % 32
            assert (((2 * rows + 6) * rp + 3) // 4 * 4 + 64 * 128) * 4 <= 160 * 1024      # wgrad3_stem_lds_bytes
            assert nd['wg3_numel'] == 128 * 147
        elif nd['taps'] == 9:
            nh = o['N'] * o['H']
            assert S * rows >= nh and (S - 1) * rows < nh
            assert S16 * rows16 >= nh and (S16 - 1) * rows16 < nh and rows16 % 2 == 0
            assert nd['wg3_numel'] == 32 * 128 * 9
        else:
            m = o['N'] * o['H'] * o['W']
            assert rows % 64 == 0 and rows16 % 64 == 0
            assert S * rows >= m and (S - 1) * rows < m
            assert S16 * rows16 >= m and (S16 - 1) * rows16 < m
            assert S <= 256 and S16 <= 128
            # the fused data + weight gradient (fp32) writes one partial tile per row block of its launch: the node's slice holds
            # at least min(32-row tiles, 256) of them
            assert nd['fuse_wgrad'] == (1 if nd.get('head', -1) < 0 else 0)
            if nd['fuse_wgrad']:
                assert nd['wg3_cap'] >= min((m + 31) // 32, 256)
        cap = nd['wg3_cap']
        assert cap >= S
        per_bucket.setdefault(nd['bucket'], []).append((nd['wg3_part'], nd['wg3_part'] + cap * nd['wg3_numel']))
    assert seen >= 20 * cfg['layer_num']
    L = cfg['layer_num']
    lo_hi = {}
    for b, spans in per_bucket.items():
        spans.sort()
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 <= b0, (b, a0, a1, b0, b1)
        lo_hi[b] = (spans[0][0], spans[-1][1])
    # two partial regions, alternating in the order backward visits the buckets (U-Nets L-1 .. 0, then the stem): neighbours in that
    # order never share floats, so the caller's stream (fused launches) only waits for the reduce two buckets back
    order = [b for b in list(range(L - 1, -1, -1)) + [L] if b in lo_hi]
    for b0, b1 in zip(order, order[1:]):
        (a0, a1), (c0, c1) = lo_hi[b0], lo_hi[b1]
        assert a1 <= c0 or c1 <= a0, (b0, b1, lo_hi[b0], lo_hi[b1])


def test_adapter_pairs_are_marked_on_the_down_blocks():
    """Node::pair (plan.cpp): the ahead adapter of every down block is followed by the skip adapter over the same concat
    (models/cu_net.py:139-142) and carries pair = 1 -- 4 per U-Net; planner option pair_adapters = 0 clears it."""
    from cu_net_amd._lib import set_planner_option
    try:
        for on in (1, 0):
            set_planner_option('pair_adapters', on)
            d = PlanHandle(4, 32, 128, 16, 3, 1, 3, 2, 256, 256).describe()
            nodes = d['nodes']
            marked = [i for i, nd in enumerate(nodes) if nd.get('pair')]
            assert len(marked) == (4 * 3 if on else 0)
            for i in marked:
                a, b = nodes[i], nodes[i + 1]
                assert '.down_blocks.' in a['name'] and '.adapters_ahead.' in a['name']
                assert b['name'] == a['name'].replace('.adapters_ahead.', '.adapters_skip.')
                assert a['segs'] == b['segs'] and a['taps'] == b['taps'] == 1 and a['bucket'] == b['bucket']
                assert d['tensors'][a['out']]['C'] == d['tensors'][b['out']]['C']
    finally:
        set_planner_option('pair_adapters', 1)


def test_planner_options_through_the_abi():
    """cunet_set_planner_option (include/cunet.h): unknown names and negative values are refused (dgrad_rows = -1, its "by f32_split"
    default, is the one negative value accepted), and the options of round 4 reach the plan: with the split contraction (f32_split = 1,
    the default) an fp32 weight-gradient launch is cut into at most 192 workgroups, on the fp32 matrix pipe into at most 256."""
    from cu_net_amd._lib import CUNetError, PlanHandle, set_planner_option
    with pytest.raises(CUNetError):
        set_planner_option('no_such_option', 1)
    for name in ('f32_split', 'stem_split', 'dgrad3_ring', 'dgrad3_nt', 'wgrad3_max_splits', 'heads_on_side'):
        with pytest.raises(CUNetError):
            set_planner_option(name, -1)
    set_planner_option('dgrad_rows', -1)
    with pytest.raises(CUNetError):
        set_planner_option('dgrad_rows', -2)
    smax = {}
    try:
        for split in (1, 0):
            set_planner_option('f32_split', split)
            plan = PlanHandle(4, 32, 128, 68, 2, 1, 2, batch=24, height=256, width=256)
            d = plan.describe()
            smax[split] = max(nd.get('wg3', 0) for nd in d['nodes'] if nd['op'] == 'conv')
            del plan
    finally:
        set_planner_option('f32_split', 1)
    assert smax[1] == 192 and smax[0] == 256, smax


def test_planner_options_read_back_and_are_restored_between_tests():
    """cunet_get_planner_option (round 5): every option of cu_net_amd._lib.PLANNER_OPTIONS reads back what cunet_set_planner_option stored,
    an unknown name is refused, and the defaults are the ones include/cunet.h documents for the shipped library (f32_split 1,
    dgrad_rows_v 2, popcount_pixels 1, stem_fuse_dz 1, wgrad_fork_group 0 = by depth).  The autouse fixture of tests/conftest.py
    restores whatever a test changes: this test leaves f32_split at 0 ON PURPOSE and the next one checks it came back."""
    from cu_net_amd._lib import CUNetError, PLANNER_OPTIONS, get_planner_option, planner_options_snapshot, set_planner_option
    snap = planner_options_snapshot()
    assert set(snap) == set(PLANNER_OPTIONS)
    if not os.environ.get('CUNET_TEST_PLANNER_OPTS'):
        assert (snap['f32_split'], snap['dgrad_rows_v'], snap['popcount_pixels'], snap['stem_fuse_dz'], snap['wgrad_fork_group'], snap['dgrad_rows']) == (1, 2, 1, 1, 0, -1)
        assert (snap['fuse_pool_gather'], snap['fuse_z_gather'], snap['stem_wgrad_caller'], snap['wgrad_split_planes'], snap['stem_wgrad_planes']) == (1, 0, 0, 0, 1)      # round 6
    with pytest.raises(CUNetError):
        get_planner_option('no_such_option')
    for name in PLANNER_OPTIONS:
        v = get_planner_option(name)
        set_planner_option(name, 3)
        assert get_planner_option(name) == 3
        set_planner_option(name, v)
        assert get_planner_option(name) == v
    set_planner_option('f32_split', 0)          # (left behind deliberately)


def test_planner_options_were_restored():
    from cu_net_amd._lib import get_planner_option
    if not os.environ.get('CUNET_TEST_PLANNER_OPTS'):
        assert get_planner_option('f32_split') == 1


def test_debug_plan_option_is_limited_to_launch_time_choices():
    """cunet_debug_set_plan_option (round 5): only wgrad_bf16_dma -- a choice between bit-identical kernels made at launch time -- may
    change in a live plan's snapshot; every option that shaped the plan's layout is refused."""
    from cu_net_amd._lib import CUNetError, PlanHandle, check, lib
    plan = PlanHandle(2, 8, 16, 5, 2, 1, 2, batch=2, height=128, width=128)
    check(lib().cunet_debug_set_plan_option(plan.h, b'wgrad_bf16_dma', 0), 'set')
    check(lib().cunet_debug_set_plan_option(plan.h, b'wgrad_bf16_dma', 1), 'set')
    for name in (b'f32_split', b'wgrad3_max_splits', b'no_such_option'):
        with pytest.raises(CUNetError):
            check(lib().cunet_debug_set_plan_option(plan.h, name, 1), 'set')
    with pytest.raises(CUNetError):
        check(lib().cunet_debug_set_plan_option(plan.h, b'wgrad_bf16_dma', -1), 'set')
