"""CPU: the nn.Module surface of create_cu_net (no kernels run here): state_dict layout, reference-
identical initialisation, conv enumeration order used by the quantisers, arena aliasing, error
behaviour, and that the shared library exports the whole C ABI of include/cunet.h."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

import cu_net_amd
from cu_net_amd import _lib
from oracle import cunet_ref as O
from tests._golden import GOLDEN_DIR, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'cunet.h')).read()
    declared = set(re.findall(r'\b(cunet_[a-z0-9_]+)\s*\(', hdr)) - {'cunet_bucket_cb'}
    L = ctypes.CDLL(_lib.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), f'{sym} declared in include/cunet.h but not exported'
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    _lib.lib()      # argtypes bind


def test_state_dict_matches_reference_layout():
    for cfg in [(4, 32, 128, 68, 2, 1, 2), (4, 32, 128, 16, 4, 2, 2), (2, 4, 8, 3, 3, 0, 1)]:
        net = cu_net_amd.create_cu_net(*cfg)
        ents = O.state_entries(O.Spec(*cfg))
        sd = net.state_dict()
        assert list(sd.keys()) == [e[0] for e in ents]
        for (k, v), (_, shp, kind) in zip(sd.items(), ents):
            assert tuple(v.shape) == tuple(shp), k
            assert v.dtype == (torch.int64 if kind == 'counter' else torch.float32)


def test_init_is_identical_to_the_reference():
    z = np.load(os.path.join(GOLDEN_DIR, 'G_init_L2K16.npz'))
    cfg = [int(v) for v in z['cfg']]
    torch.manual_seed(int(z['seed']))
    net = cu_net_amd.create_cu_net(*cfg)
    sd = net.state_dict()
    for k, (s, a) in zip(z['names'].tolist(), z['sums']):
        assert float(sd[k].double().sum()) == s and float(sd[k].double().abs().sum()) == a, k
    assert np.array_equal(sd['hg.up_blocks.2.adapters_ahead.1.adapter_conv.weight'][:4, :8, 0, 0].numpy(), z['probe'])
    # utils/quantize.py:81-102 enumerates nn.Conv2d in modules() order and skips the first and the last
    convs = [n for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
    assert convs == z['conv_order'].tolist()
    assert convs[0] == 'features.conv0' and convs[-1] == 'intermedia.adapters.0.adapter_conv'


def test_parameters_alias_one_flat_arena_and_survive_data_replacement():
    net = cu_net_amd.create_cu_net(2, 4, 8, 3, 2, 1, 2)
    base = net._param_arena.data_ptr()
    for p, (off, n, shape, _) in zip(net._param_list, net._param_meta):
        assert p.data_ptr() == base + 4 * off
    w = dict(net.named_parameters())['hg.down_blocks.1.layers.0.conv2.weight']
    w.data = w.data.add(1.0)                      # what utils/quantize.py:115 does
    assert w.data_ptr() != base + 4 * [m for p, m in zip(net._param_list, net._param_meta) if p is w][0][0]
    net._check_aliasing()
    off = [m for p, m in zip(net._param_list, net._param_meta) if p is w][0][0]
    assert w.data_ptr() == net._param_arena.data_ptr() + 4 * off
    assert net._param_arena.data_ptr() == base      # arena kept, values copied back in
    g = Golden('G1_L2_o1')
    net.load_state_dict(g.group('state0'))
    for k, v in g.group('state0').items():
        assert torch.equal(net.state_dict()[k], v)


def test_errors_mirror_the_reference():
    with pytest.raises(AssertionError):
        cu_net_amd.create_cu_net(4, 32, 128, 16, 2, 1, 16)      # README default loss_num trips cu_net.py:274
    with pytest.raises(ValueError):
        cu_net_amd.create_cu_net(4, 32, 128, 16, 2, 2, 2)       # cu_net.py:285-287 (exit() there)
    net = cu_net_amd.create_cu_net(2, 4, 8, 3, 2, 1, 2)
    with pytest.raises(cu_net_amd.CUNetError):
        net(torch.rand(1, 3, 64, 64))                           # no CPU fallback
    with pytest.raises(cu_net_amd.CUNetError):
        cu_net_amd.get_preds(torch.rand(1, 2, 8, 8))


def test_reference_import_path():
    from models.cu_net import create_cu_net
    assert create_cu_net is cu_net_amd.create_cu_net
