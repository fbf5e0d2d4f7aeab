"""CPU: the C++ plan builder (cu_net_amd/csrc/plan.cpp, reached through the C ABI) wires the network
exactly like the reference: executing its node list with torch ops reproduces the golden outputs."""
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
