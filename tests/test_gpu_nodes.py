"""GPU (-m gpu): every backward kernel path, node by node, against torch autograd on IDENTICAL inputs.

Whole-network gradient comparisons are chaotic in fp32 (a ReLU/max-pool decision that flips because
z differs in the 7th digit changes a gradient element by O(1), and the reference's own fp32 result
differs from an fp64 evaluation by 1e-2 on these tiny nets), so the exact checks are local: for each
node of the plan the node's inputs are the activations the GPU itself produced, d(loss)/d(output) is
a seeded random tensor, and the HIP kernels' input gradients / weight gradient / dgamma / dbeta are
compared with torch's on the CPU.  fp32 tolerance: max|hip-ref| <= 2e-4 * max|ref| for all but a
1e-3 fraction of elements (z == 0 +- rounding can still flip a mask), and relative L2 error <= 1e-3
over the remaining elements.
"""
import pytest
import torch
import torch.nn.functional as F

import cu_net_amd
from tests._golden import Golden

pytestmark = pytest.mark.gpu


def _close(name, got, ref, bad, rtol=2e-4, frac=1e-3, l2=1e-3, slack=None):
    """slack: per-element absolute allowance that the CALLER derived from the inputs (see _quantiser_tie_slack) -- it is
    subtracted from the difference before any of the tolerances is applied, element by element."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    mag = ref.abs().max().item()
    diff = (got - ref).abs()
    if slack is not None:
        diff = (diff - slack.detach().float().cpu()).clamp_min(0.0)
    allowed = max(2, int(frac * ref.numel()))      # elements whose ReLU mask may legitimately flip (z == 0 +- rounding)
    nbad = int((diff > rtol * mag + 1e-7).sum())
    # relative L2 without the `allowed` worst elements: one flipped mask under a large dy would otherwise dominate it
    d = diff.flatten().double()
    if d.numel() > allowed:
        d = d.sort().values[:d.numel() - allowed]
    rel2 = (d.norm() / (ref.double().norm() + 1e-30)).item()
    if nbad > allowed or rel2 > l2 or not bool(torch.isfinite(got).all()):
        bad.append(f'{name}: {nbad}/{ref.numel()} elements off, max err {diff.max().item():.3e} (mag {mag:.3e}), relL2 {rel2:.2e}')


@pytest.mark.parametrize('tag', ['G9_L2_o1_c32', 'G2_L3_o2', 'G4_L2_o0', 'G6_L2_o1_hw64'])
def test_every_node_backward_matches_autograd(tag):
    g = Golden(tag)
    x = g.t('x')
    st = g.group('state0')
    _check_all_nodes(g.cfg, st, x)


def test_every_node_backward_full_width():
    """The production channel widths (4 / 32 / 128, K = 68), one 256x256 image: these are the shapes that select
    the wide-tile, tap-split, multi-consumer-gather and vector-operand weight-gradient kernel variants."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=21)
    x, _ = O.synthetic_batch(1, 68, 256, seed=22)
    _check_all_nodes(cfg, st, x)


@pytest.mark.parametrize('n', [1, 5])
def test_every_node_backward_with_the_row_ring_kernels_forced(n):
    """The row-walking kernels of the split contraction on EVERY launch they support instead of the large ones only (planner options
    dgrad3_ring = 1: dgrad3x3_ring_split_kernel; conv3x3_ring_min_rows = 1: conv3x3_ring_split_kernel; dgrad_rows = 1:
    dgrad1x1_rows_split_kernel): one and five images (five: 320 image rows at 64 x 64 in ranges of three, so a workgroup's row range starts and ends inside an image and crosses image
    boundaries (the rows above / below an image edge must contribute zeros, not the neighbouring image's rows) and the launch has
    fewer row blocks than CUs.  Node by node against autograd on identical inputs (models/cu_net.py:45-48 and its backward)."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=23)
    x, _ = O.synthetic_batch(n, 68, 256, seed=24)
    try:
        set_planner_option('dgrad3_ring', 1)
        set_planner_option('conv3x3_ring_min_rows', 1)
        set_planner_option('dgrad_rows', 1)
        _check_all_nodes(cfg, st, x, check_forward=True)
    finally:
        set_planner_option('dgrad3_ring', 768)               # (the defaults)
        set_planner_option('conv3x3_ring_min_rows', 512)
        set_planner_option('dgrad_rows', -1)


@pytest.mark.parametrize('mode', [False, True, 2])
def test_every_node_backward_full_width_wgrad3(mode):
    """The LDS-staged, atomics-free 1x1 weight gradient (wgrad3: fp32 MFMA on fp32 / bf16 activations, bf16 MFMA with bf16
    gradient tensors) forced onto every eligible node, N = 2: split-K and channel-halved tile ownerships (Ccat 128 ... 320),
    the up-sample gather, ragged last chunks (32 .. 8192 rows) and the per-node partial reduce."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=27)
    x, _ = O.synthetic_batch(2, 68, 256, seed=28)
    _check_all_nodes(cfg, st, x, bf16=mode, wgrad3_all=True)


@pytest.mark.parametrize('tag', ['full', 'G9_L2_o1_c32'])
def test_every_node_backward_with_quan_input(tag):
    """Quantised-input mode (QuanInput2d in front of the 3x3 convs and the heads, utils/quantize.py:47-63): the data
    gradient applies the straight-through mask (no gradient where the activation is >= 1) and the weight gradient contracts
    with the QUANTISED activation -- node by node against torch autograd through the oracle's QuanInput function."""
    from oracle import cunet_ref as O
    if tag == 'full':
        cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
        spec = O.Spec(**cfg)
        st = O.init_state(spec, seed=29)
        x, _ = O.synthetic_batch(1, 68, 256, seed=30)
    else:
        g = Golden(tag)
        cfg, st, x = g.cfg, g.group('state0'), g.t('x')
    for k in st:                                       # wider BatchNorm outputs: plenty of activations beyond 1 (the STE mask acts)
        if k.endswith('norm2.weight') or k.endswith('.norm.weight'):
            st[k] = st[k] * 2.0
    _check_all_nodes(cfg, st, x, quan_input_bits=8)


@pytest.mark.parametrize('mode', [False, 2])
def test_every_node_backward_rectangular_full_width(mode):
    """A 128 x 256 input at production widths, N = 4: image rows of 64 / 32 / 16 / 8 / 4 pixels with half as many rows per image
    as a square input has -- the row-walking weight-gradient kernels (3x3 ring in fp32 and on bf16 MFMA, stem) cross image
    boundaries at other places, the last workgroup of a range is ragged."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=35)
    gen = torch.Generator().manual_seed(36)
    x = torch.rand(4, 3, 128, 256, generator=gen)
    _check_all_nodes(cfg, st, x, bf16=mode, wgrad3_all=True)


@pytest.mark.parametrize('n,h,w', [(1, 256, 256), (3, 128, 256), (5, 256, 128), (24, 256, 256), (30, 128, 128)])
def test_stem_forward_on_the_input_row_ring(n, h, w):
    """stem_fwd_split_kernel (planner option stem_split with f32_split): conv0 7x7 / 2 pad 3 (models/cu_net.py:300) computed from an LDS
    ring of input rows cut into bf16 pieces, k re-ordered to (channel, kernel row) x 8 consecutive pixels.  Against torch's conv2d on the
    same image and weights, and the per-channel batch statistics the kernel emits against the output's own sums: one image (fewer output
    rows than workgroups), rectangular images (64 / 128 output columns; row ranges that cross image boundaries: the rows above an
    image's top edge are zeros, not the previous image's rows), the bench batch, many small images."""
    import torch.nn.functional as F
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=37)
    gen = torch.Generator().manual_seed(38)
    x = torch.rand(n, 3, h, w, generator=gen) * 2.0 - 0.5
    outs = {}
    try:
        for opt in (1, 0):
            set_planner_option('stem_split', opt)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(n, h, w, True)
            plan.forward(x.cuda(), True, want_outputs=False)
            torch.cuda.synchronize()
            d = plan.handle.describe()
            assert d['nodes'][0]['op'] == 'stem_conv'
            name = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][0]['out']][0]
            outs[opt] = plan.debug_tensor(name).clone()
            del plan, net
    finally:
        set_planner_option('stem_split', STEM_SPLIT_DEFAULT)
    ref = F.conv2d(x.double(), st['features.conv0.weight'].double(), None, 2, 3).float().cuda()
    scale = float(ref.abs().max())
    for opt in (1, 0):
        assert outs[opt].shape == ref.shape
        err = float((outs[opt] - ref).abs().max())
        assert err <= 2e-6 * scale, (opt, err, scale)


STEM_SPLIT_DEFAULT = 1      # planner option stem_split as plan.h ships it


@pytest.mark.parametrize('n,h,w', [(1, 256, 256), (3, 128, 256), (5, 256, 128), (30, 128, 128)])
def test_stem_weight_gradient_shapes(n, h, w):
    """The LDS-staged stem weight gradient (wgrad3_stem_kernel) over its planning range: one output row per workgroup (N = 1),
    rectangular images (64 / 128 output columns: one / two dY chunks per row, ragged last workgroup of an image), many images
    (several rows per workgroup, two rounds of workgroups) -- against autograd's conv2d weight gradient, and the per-wave
    atomic kernel it replaces on the same inputs."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=33)
    gen = torch.Generator().manual_seed(34)
    x = torch.rand(n, 3, h, w, generator=gen)
    _check_all_nodes(cfg, st, x, wgrad3_all=True, only_ops=('stem_conv',))
    _check_all_nodes(cfg, st, x, wgrad3_all=False, only_ops=('stem_conv',))


def test_every_node_backward_full_width_bf16_activations():
    """The same node-by-node check with bf16 activation storage (FusedTrainer(bf16=True)): the backward kernels read x
    as bf16 and compute in fp32, so against torch autograd fed with the SAME bf16-rounded activations the fp32 tolerance
    holds unchanged (gradients, dz and weights stay fp32)."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=23)
    x, _ = O.synthetic_batch(2, 68, 256, seed=24)       # the bf16 kernels need 32-row tiles at every level: N * 16 rows at the neck
    _check_all_nodes(cfg, st, x, bf16=True)


def test_every_node_backward_full_width_bf16_gradient_tensors():
    """bf16 activations AND bf16 gradient tensors (FusedTrainer(bf16_grads=True)): d(loss)/d(out) is poked as bf16 and
    the reference differentiates with the same rounded values, so weight / BatchNorm parameter gradients (fp32
    accumulation from identical inputs) keep the fp32 tolerance; the input gradients are STORED as bf16 (8 mantissa
    bits): 1e-2 of the tensor's magnitude per element, 6e-3 relative L2."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=25)
    x, _ = O.synthetic_batch(2, 68, 256, seed=26)
    _check_all_nodes(cfg, st, x, bf16=2)


TIE_BAND = 2e-3      # in quantiser steps: an activation this close to a rounding boundary may land on either neighbouring level


def _relu_tie_slack(x, z, gamma, beta, dact):
    """Per-channel allowance (dbeta, dgamma) for ReLU decisions that may fall either way: elements with |z| within 8 * 2^-24 of the magnitudes
    that cancel into it (|x * scale| + |mean * scale| + |beta|), i.e. z = 0 to the rounding of either evaluation order.  A flip changes dbeta[c] by |d(loss)/d(relu out)| and dgamma[c] by that times |xhat|."""
    if dact is None:
        zero = torch.zeros(z.shape[1])
        return zero, zero
    mean = x.mean((0, 2, 3), keepdim=True)
    var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    xhat = (x - mean) / torch.sqrt(var + 1e-5)
    # the magnitudes that cancel into z in EITHER evaluation: torch's (x - mean) * invstd * gamma + beta and the kernels' fma(x, scale, shift) with
    # scale = gamma * invstd, shift = beta - mean * scale (a channel whose mean is many standard deviations from zero has |x * scale| and |shift|
    # far above |z|: the rounding of scale and shift alone moves z by 2^-24 of THOSE)
    g = (gamma.view(1, -1, 1, 1) / torch.sqrt(var + 1e-5)).abs()
    terms = g * x.abs() + g * mean.abs() + beta.view(1, -1, 1, 1).abs()
    tie = (z.abs() <= 8.0 * 2.0 ** -24 * terms).float()
    w = tie * dact.abs()
    return w.sum((0, 2, 3)), (w * xhat.abs()).sum((0, 2, 3))


def _quantiser_tie_slack(pre_q, dy, bits, taps):
    """The weight gradient at a QuanInput2d site contracts dY with the QUANTISED activation round(a * 2^(bits-1)) / 2^(bits-1)
    (utils/quantize.py:33-42,47-63).  The reference side of this test recomputes `a` (BatchNorm + ReLU) on the CPU from the GPU's
    tensors; the kernel computes the same `a` with its own fp64-statistics BatchNorm table, equal to a few ulps -- so an `a` that
    sits within rounding of a bucket boundary (k + 1/2) * 2^-(bits-1) lands on the OTHER level on one side.  That is a legitimate,
    discontinuous difference of exactly one quantiser step of one activation, and each such activation (pixel p, channel c) moves
    dW[o, c, tap] by dy[o, p + tap] * 2^-(bits-1).  Rather than widening a tolerance, the bound is computed from the inputs:
        slack[o, c, tap] = 2^-(bits-1) * sum_{p : a[c, p] within TIE_BAND steps of a boundary}  |dy[o, p + tap]|
    i.e. the weight gradient of the tie mask against |dY|.  TIE_BAND = 2e-3 steps = 1.6e-5 absolute at 8 bits, ~100 ulps of a value
    near 1 (the BatchNorm tables differ by fp32 rounding of scale and shift: |x * scale| and |shift| of a few units each);
    about 4e-3 of the positive activations are inside the band, each worth <= |dy| / 128 in 9 x Cout elements, so everything NOT
    explained by a possible tie flip is still held to the unchanged fp32 tolerance.  (round 4: `hg.down_blocks.2.layers.0.conv2
    dW: 135/36864 elements off, max err 2.386e-02` = 3.05 / 128 -- tools/diag_quan_tie.py identifies the flipped activation.)"""
    step = 2.0 ** (bits - 1)
    m = _quantiser_tie_mask(pre_q, bits)
    pad = 1 if taps == 9 else 0
    # weight gradient of conv2d(m, W) w.r.t. W under output gradient |dy|
    wz = torch.zeros(dy.shape[1], m.shape[1], 3 if taps == 9 else 1, 3 if taps == 9 else 1, requires_grad=True)
    F.conv2d(m, wz, None, 1, pad).backward(dy.abs())
    return wz.grad / step


def _quantiser_tie_mask(pre_q, bits):
    step = 2.0 ** (bits - 1)
    t = pre_q.double() * step
    fr = t - torch.floor(t)
    return (((fr - 0.5).abs() <= TIE_BAND) & (pre_q > 0) & (pre_q.double() < 1.0 - 1.0 / step + TIE_BAND / step)).float()


def _quantiser_tie_slack_forward(pre_q, w, bits, taps):
    """Forward twin of _quantiser_tie_slack: an activation inside the tie band may differ by one step 2^-(bits-1) between the two
    sides, which moves output (o, p) by |w[o, c, tap]| * 2^-(bits-1): slack = conv2d(tie mask, |W|) / 2^(bits-1)."""
    with torch.no_grad():
        return F.conv2d(_quantiser_tie_mask(pre_q, bits), w.abs(), None, 1, 1 if taps == 9 else 0) / 2.0 ** (bits - 1)


def _check_all_nodes(cfg, st, x, bf16=False, wgrad3_all=False, quan_input_bits=0, only_ops=None, check_forward=False, node_filter=None):
    """node_filter(k, node, desc) -> bool: check only the nodes it accepts (deep plans at the bench batch: every node of the first and the
    last U-Nets, the nodes whose tensors straddle a 4 GB boundary of the workspace and a sample of the rest).
    wgrad3_all = True: every eligible 1x1 weight gradient on the LDS-staged atomics-free kernel (the planner's default);
    False: the planner is told to keep them on the per-wave atomic kernel (wgrad2), which stays the path of narrow heads,
    non-32-multiple concats and the stem and must remain covered at production widths.
    check_forward: every conv / pool node's FORWARD output is compared too, with torch's on the GPU's own inputs (bf16 storage: on
    the bf16-rounded operands the kernels multiply, output rounded as the kernel stores it)."""
    from cu_net_amd._lib import set_planner_option
    set_planner_option('wgrad3_min_rows', 0 if wgrad3_all else 1 << 30)
    try:
        return _check_all_nodes_impl(cfg, st, x, bf16, wgrad3_all, quan_input_bits, only_ops, check_forward, node_filter)
    finally:
        set_planner_option('wgrad3_min_rows', 0)


def _check_all_nodes_impl(cfg, st, x, bf16, wgrad3_all, quan_input_bits=0, only_ops=None, check_forward=False, node_filter=None):
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    if quan_input_bits:
        net.set_quant_input(quan_input_bits, ())        # QuanInput2d sites on MFMA (weights are not ternary here)
    n, _, h, w = x.shape
    plan = net._get_plan(n, h, w, True, bf16=bf16)
    xd = x.cuda()
    gb = bf16 == 2                                          # gradient tensors stored as bf16
    dx_tol = dict(rtol=1e-2, l2=6e-3) if gb else {}
    if bf16:
        plan.forward_bf16(xd, 2 if gb else 1, want_outputs=False)
    else:
        plan.forward(xd, True, want_outputs=False)
    torch.cuda.synchronize()
    desc = plan.handle.describe()
    T = desc['tensors']
    if wgrad3_all:
        assert sum(1 for nd in desc['nodes'] if nd.get('wg3', 0) > 0) >= 20 * cfg['layer_num'], 'the planner did not select wgrad3'
        if only_ops and 'stem_conv' in only_ops:
            assert desc['nodes'][0]['op'] == 'stem_conv' and desc['nodes'][0]['wg3'] > 0, 'the planner did not select the LDS-staged stem kernel'
    elif only_ops and 'stem_conv' in only_ops:
        assert desc['nodes'][0]['wg3'] == 0
    if node_filter is not None:      # (only the tensors the selected nodes touch leave the GPU)
        picked = [k for k, nd in enumerate(desc['nodes']) if node_filter(k, nd, desc)]
        assert picked, 'node_filter selected nothing'
        need = set()
        for k in picked:
            need.add(desc['nodes'][k]['out'])
            need.update(sg['t'] for sg in desc['nodes'][k]['segs'])
        acts = {T[i]['name']: plan.debug_tensor(T[i]['name']).cpu() for i in sorted(need)}
        picked = set(picked)
    else:
        picked = None
        acts = {t['name']: plan.debug_tensor(t['name']).cpu() for t in T}
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}

    def pgrad(name):
        o, nmel, shape = off[name]
        return net._grad_arena[o:o + nmel].view(shape).cpu()

    bad = []
    for k, nd in enumerate(desc['nodes']):
        if only_ops is not None and nd['op'] not in only_ops:
            continue
        if picked is not None and k not in picked:
            continue
        gen = torch.Generator().manual_seed(1000 + k)
        oname = T[nd['out']]['name']
        dy = torch.randn(acts[oname].shape, generator=gen)
        if gb and nd['op'] != 'stem_conv':
            dy = dy.bfloat16().float()                      # what the kernels will read
        op = nd['op']
        if op == 'conv':
            leaves = [acts[T[s['t']]['name']].clone().requires_grad_(True) for s in nd['segs']]
            parts = [F.interpolate(l, scale_factor=2, mode='nearest') if s['ups'] else l for l, s in zip(leaves, nd['segs'])]
            cat = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
            gamma = st[nd['bn'] + '.weight'].clone().requires_grad_(True)
            beta = st[nd['bn'] + '.bias'].clone().requires_grad_(True)
            wt = st[nd['conv'] + '.weight'].clone()
            head_on_bf16 = nd.get('head', -1) >= 0 and T[nd['out']].get('gld16', T[nd['out']]['ld']) != T[nd['out']]['ld']
            if gb and (nd.get('head', -1) < 0 or head_on_bf16):
                # the bf16-MFMA data gradient contracts with bf16-rounded weights (heads: when their gradient tensor is padded to
                # the MFMA's K, i.e. whenever it fits the tensor's slot -- the operand their bf16 forward multiplied as well)
                wt = wt.bfloat16().float()
            wt.requires_grad_(True)
            pre = F.batch_norm(cat, None, None, gamma, beta, True, 0.1, 1e-5)
            relu_out = F.relu(pre)
            relu_out.retain_grad()                              # d(loss)/d(relu output): what a flipped ReLU mask adds to / removes from a sum
            act = relu_out
            if quan_input_bits and (nd['taps'] == 9 or nd.get('head', -1) >= 0):
                from oracle.cunet_ref import _QuanInputFn       # QuanInput2d site: quantised forward, straight-through backward
                pre_q = act.detach()
                act = _QuanInputFn.apply(act, quan_input_bits)
            dw_tol = {}
            if quan_input_bits and (nd['taps'] == 9 or nd.get('head', -1) >= 0):
                dw_tol = dict(slack=_quantiser_tie_slack(pre_q, dy, quan_input_bits, nd['taps']))
            if gb and nd.get('wg3', 0) > 0 and (nd['taps'] == 1 or T[nd['out']]['W'] in (16, 32, 64)):
                # (an activation whose fp32 value sits within an ulp of a bf16 rounding boundary rounds the other way in the kernel:
                # one bf16 step of one activation, 4e-3 * |act * dy|, in a sum over a few hundred rows is above 2e-4 of max|dW|)
                dw_tol = dict(rtol=1e-3)
                # the bf16-MFMA weight gradient (1x1: wgrad3_bf16_kernel; 3x3 at W = 16 / 32 / 64: wgrad3_3x3_bf16_kernel) contracts
                # dY with the bf16-ROUNDED activation -- exactly the operand the bf16 forward multiplied the weights with, i.e. the
                # exact gradient of that forward (straight-through here)
                act = act + (act.bfloat16().float() - act).detach()
            y = F.conv2d(act, wt, None, 1, 1 if nd['taps'] == 9 else 0)
            if check_forward:
                with torch.no_grad():
                    if bf16:      # bf16 MFMA: bf16-rounded activation x bf16-rounded weight, fp32 accumulation; stored as bf16 (heads: fp32)
                        yf = F.conv2d(act.bfloat16().float(), wt.bfloat16().float(), None, 1, 1 if nd['taps'] == 9 else 0)
                        if nd.get('head', -1) < 0:
                            yf = yf.bfloat16().float()
                        _close(f'{nd["name"]} forward', acts[oname], yf, bad, rtol=1e-2, l2=6e-3)
                    else:
                        _close(f'{nd["name"]} forward', acts[oname], y, bad)
            y.backward(dy)
            plan.debug_poke(oname, dy, grad=True)
            plan.debug_run_node_backward(k)
            torch.cuda.synchronize()
            for l, s in zip(leaves, nd['segs']):
                nm = T[s['t']]['name']
                _close(f'{nd["name"]} dX[{nm}]', plan.debug_tensor(nm, grad=True), l.grad, bad, **dx_tol)
            _close(f'{nd["name"]} dW', pgrad(nd['conv'] + '.weight'), wt.grad, bad, **dw_tol)
            # dgamma / dbeta are sums over ALL rows of a channel: ONE ReLU decision that falls the other way (z = gamma * xhat + beta within its own
            # rounding error of 0: the kernel's fused multiply-add from fp64-derived scale / shift against torch's operation order) moves the
            # whole channel by |d(loss)/d(relu out)| of that element -- at the bench batch (98 304 rows per channel at 64 x 64) about one channel
            # per node has such an element, and with them the count of "elements off" is the count of ties, not a kernel property (round 6: a
            # session on the fp32 matrix pipe found three in one node of the CU-Net-16 plan).  As for the quantiser ties of round 5 the
            # allowance is derived from the inputs, per channel: the sum of |d(loss)/d(relu out)| (times |xhat| for dgamma) over the elements whose z
            # is within 8 ulp of the magnitudes that cancel into it of zero.
            bn_slack = _relu_tie_slack(cat.detach(), pre.detach(), gamma.detach(), beta.detach(), relu_out.grad)
            _close(f'{nd["name"]} dgamma', pgrad(nd['bn'] + '.weight'), gamma.grad, bad, slack=bn_slack[1])
            _close(f'{nd["name"]} dbeta', pgrad(nd['bn'] + '.bias'), beta.grad, bad, slack=bn_slack[0])
        elif op == 'pool':
            nm = T[nd['segs'][0]['t']]['name']
            leaf = acts[nm].clone().requires_grad_(True)
            yp = F.max_pool2d(leaf, 2, 2)
            if check_forward:
                assert torch.equal(acts[oname], yp.detach()), nd['name'] + ' forward'               # a selection: bit-exact in every storage mode
            yp.backward(dy)
            plan.debug_poke(oname, dy, grad=True)
            plan.debug_run_node_backward(k)
            torch.cuda.synchronize()
            assert torch.equal(plan.debug_tensor(nm, grad=True).cpu(), leaf.grad), nd['name']   # index map: bit-exact (bf16: a copy)
        elif op == 'stem_bnpool':
            nm = T[nd['segs'][0]['t']]['name']
            leaf = acts[nm].clone().requires_grad_(True)
            gamma = st[nd['bn'] + '.weight'].clone().requires_grad_(True)
            beta = st[nd['bn'] + '.bias'].clone().requires_grad_(True)
            F.max_pool2d(F.relu(F.batch_norm(leaf, None, None, gamma, beta, True, 0.1, 1e-5)), 2, 2).backward(dy)
            plan.debug_poke(oname, dy, grad=True)
            plan.debug_run_node_backward(k)
            torch.cuda.synchronize()
            _close(f'{nd["name"]} dX', plan.debug_tensor(nm, grad=True), leaf.grad, bad)
            _close(f'{nd["name"]} dgamma', pgrad(nd['bn'] + '.weight'), gamma.grad, bad)
            _close(f'{nd["name"]} dbeta', pgrad(nd['bn'] + '.bias'), beta.grad, bad)
        elif op == 'stem_conv':
            wt = st[nd['conv'] + '.weight'].clone().requires_grad_(True)
            F.conv2d(x, wt, None, 2, 3).backward(dy)
            plan.debug_poke(oname, dy, grad=True)
            plan.debug_run_node_backward(k)
            torch.cuda.synchronize()
            _close(f'{nd["name"]} dW', pgrad(nd['conv'] + '.weight'), wt.grad, bad)
    assert not bad, f'{len(bad)} mismatches:\n' + '\n'.join(bad[:40])
    return desc


# ---------------------------------------------------------------------------------------------------------------
# Composition of the whole backward, exact: after ONE real backward pass every tensor's gradient must equal the sum of
# the contributions of ALL its consumer nodes, each differentiated by torch from the GPU's own d(loss)/d(output) of
# that consumer and the GPU's own activations.  Unlike a whole-network comparison with a CPU run this does not chain
# rounding through the (chaotic) network, so it keeps the per-kernel tolerance while covering what the node tests do
# not: the per-tensor consumer lists (order-K FIFO, skip connections, intermedia carries), the 4-child sums behind
# the up-sample maps, pool routing, and that no contribution is dropped or counted twice.
def _check_composition(cfg, st, x, target, check_params=False, quan_bits_w=0, quan_input_bits=0, popcount=False):
    """quan_bits_w / quan_input_bits / popcount: the quantised loop of cu-net-prev-version-wig.py:163-190 -- QuanOp.quantization()
    before the forward, QuanInput2d in front of the 3x3 / head convs, optionally their forward on AND-popcount.  The torch side
    then differentiates the SAME quantised network: the weights it uses are read back from the GPU arena after quantization().
    Returns (net, qop) with the raw (not yet rewritten) gradients in the arena and the weights still quantised."""
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    n, _, h, w = x.shape
    qop = None
    if quan_bits_w:
        from cu_net_amd.quant import QuanOp
        qop = QuanOp(net, bits_w=quan_bits_w, bits_i=quan_input_bits or 8, bits_g=8)
    if quan_input_bits:
        net.set_quant_input(quan_input_bits, qop.target_names if popcount else ())
    if qop is not None:
        qop.quantization()
        st = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    plan = net._get_plan(n, h, w, True)
    assert (plan.popcount_nodes > 0) == bool(popcount)
    plan.forward(x.cuda(), True, want_outputs=False)
    plan.loss_mse(target.cuda())
    plan.backward(None)
    torch.cuda.synchronize()
    desc = plan.handle.describe()
    T = desc['tensors']
    acts = {t['name']: plan.debug_tensor(t['name']).cpu() for t in T}
    produced = {T[nd['out']]['name'] for nd in desc['nodes']}
    grads = {nm: plan.debug_tensor(nm, grad=True).cpu() for nm in produced}
    expect = {}
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
    pbad, bad_fwd = [], []

    def pcheck(label, name, ref, **tol):
        if check_params:
            o, nmel, shape = off[name]
            _close(label, net._grad_arena[o:o + nmel].view(shape), ref, pbad, **tol)

    def add(nm, g):
        expect[nm] = g if nm not in expect else expect[nm] + g

    heads = 0
    for nd in desc['nodes']:
        oname = T[nd['out']]['name']
        dy = grads[oname]
        if nd.get('head', -1) >= 0:           # d(loss)/d(head) = 2 (out - target) / numel (cu-net.py:175-178)
            ref = 2.0 * (acts[oname] - target) / target.numel()
            assert (dy - ref).abs().max().item() <= 1e-6 * ref.abs().max().item() + 1e-12, oname
            heads += 1
        op = nd['op']
        if op == 'conv':
            leaves = [acts[T[s['t']]['name']].clone().requires_grad_(True) for s in nd['segs']]
            parts = [F.interpolate(l, scale_factor=2, mode='nearest') if s['ups'] else l for l, s in zip(leaves, nd['segs'])]
            cat = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
            gamma = st[nd['bn'] + '.weight'].clone().requires_grad_(check_params)
            beta = st[nd['bn'] + '.bias'].clone().requires_grad_(check_params)
            wt = st[nd['conv'] + '.weight'].clone().requires_grad_(check_params)
            act = F.relu(F.batch_norm(cat, None, None, gamma, beta, True, 0.1, 1e-5))
            site = bool(quan_input_bits and (nd['taps'] == 9 or nd.get('head', -1) >= 0))
            if site:
                from oracle.cunet_ref import _QuanInputFn
                pre_q = act.detach()
                act = _QuanInputFn.apply(act, quan_input_bits)
            y = F.conv2d(act, wt, None, 1, 1 if nd['taps'] == 9 else 0)
            # the forward itself, node by node on the GPU's own inputs, in EVERY configuration (fp32 tolerance).  Quantised inputs: with
            # ternary weights and 2^-7-grid activations the 3x3 / head convs are exact on both sides -- except that an activation within
            # rounding of a quantiser boundary lands on the neighbouring 2^-7 level on one side only (with fan-in 1152 about 1 % of the
            # outputs contain such a flip): the per-element allowance for exactly those activations, |w| / 128 each, comes from the
            # inputs (_quantiser_tie_slack_forward); everything else is held to the plain fp32 tolerance
            fwd_tol = dict(slack=_quantiser_tie_slack_forward(pre_q, wt.detach(), quan_input_bits, nd['taps'])) if site else {}
            _close(f'{nd["name"]} forward', acts[oname], y, bad_fwd, **fwd_tol)
            y.backward(dy)
            for l, s in zip(leaves, nd['segs']):
                add(T[s['t']]['name'], l.grad)
            if check_params:
                # (at a QuanInput site the weight gradient contracts dY with the QUANTISED activation: an activation on a quantiser
                # boundary that lands on the other 2^-7 level moves a dW element by dy / 128 -- allowed for exactly those activations)
                site_tol = dict(slack=_quantiser_tie_slack(pre_q, dy, quan_input_bits, nd['taps'])) if site else {}
                pcheck(f'{nd["name"]} dW', nd['conv'] + '.weight', wt.grad, **site_tol)
                pcheck(f'{nd["name"]} dgamma', nd['bn'] + '.weight', gamma.grad)
                pcheck(f'{nd["name"]} dbeta', nd['bn'] + '.bias', beta.grad)
        elif op == 'pool':
            nm = T[nd['segs'][0]['t']]['name']
            leaf = acts[nm].clone().requires_grad_(True)
            yp = F.max_pool2d(leaf, 2, 2)
            if not torch.equal(acts[oname], yp.detach()):
                bad_fwd.append(f'{nd["name"]} forward: max-pool output is not the selection torch makes')
            yp.backward(dy)
            add(nm, leaf.grad)
        elif op == 'stem_bnpool':
            nm = T[nd['segs'][0]['t']]['name']
            leaf = acts[nm].clone().requires_grad_(True)
            gamma = st[nd['bn'] + '.weight'].clone().requires_grad_(check_params)
            beta = st[nd['bn'] + '.bias'].clone().requires_grad_(check_params)
            yp = F.max_pool2d(F.relu(F.batch_norm(leaf, None, None, gamma, beta, True, 0.1, 1e-5)), 2, 2)
            _close(f'{nd["name"]} forward', acts[oname], yp, bad_fwd)
            yp.backward(dy)
            add(nm, leaf.grad)
            if check_params:
                pcheck(f'{nd["name"]} dgamma', nd['bn'] + '.weight', gamma.grad)
                pcheck(f'{nd["name"]} dbeta', nd['bn'] + '.bias', beta.grad)
        elif op == 'stem_conv':
            wt = st[nd['conv'] + '.weight'].clone().requires_grad_(check_params)
            ys = F.conv2d(x, wt, None, 2, 3)
            _close(f'{nd["name"]} forward', acts[oname], ys, bad_fwd)
            if check_params:
                ys.backward(dy)
                pcheck(f'{nd["name"]} dW', nd['conv'] + '.weight', wt.grad)
    assert heads == cfg['loss_num']
    bad = []
    for nm, g in expect.items():
        _close('d ' + nm, grads[nm], g, bad)
    assert len(expect) >= len(produced) - cfg['loss_num']
    assert not bad, f'{len(bad)} tensors whose gradient is not the sum of their consumers\' contributions:\n' + '\n'.join(bad[:30])
    assert not pbad, f'{len(pbad)} parameter gradients off:\n' + '\n'.join(pbad[:30])
    assert not bad_fwd, f'{len(bad_fwd)} node forwards off:\n' + '\n'.join(bad_fwd[:30])
    return net, qop


@pytest.mark.parametrize('tag', ['G2_L3_o2', 'G3_L4_o1_ln2', 'G4_L2_o0', 'G9_L2_o1_c32'])
def test_whole_backward_is_the_sum_of_consumer_contributions(tag):
    g = Golden(tag)
    _check_composition(g.cfg, g.group('state0'), g.t('x'), g.t('target'))


def test_whole_backward_composition_full_width_cu_net4():
    """Production widths, L = 4 (U-Net indices beyond 1: intermedia adapters i >= 2 and FIFO pops at 128-wide channels)."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=4, order=1, loss_num=4)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=71)
    x, target = O.synthetic_batch(1, 16, 256, seed=72)
    _check_composition(cfg, st, x, target)


@pytest.mark.parametrize('split', [0, 1])
def test_whole_backward_composition_bench_batch(split):
    """(split = 1: planner option f32_split -- the 1x1 / 3x3 forward, the heads and the data gradients contract on the bf16 matrix pipe,
    every operand value cut into three bf16 pieces; same checks, same tolerances.)
    BASELINE config 2 exactly as bench.py runs it -- CU-Net-2, K = 68, N = 24, 256 x 256, default planner options: every
    weight-gradient workgroup walks 12 or more chunks (the double-buffer reuse of wgrad3_kernel needs >= 3), the 3x3 forward
    is on the LDS row ring, the data gradients run their multi-tile loops, 1536 image rows per level.  Composition check of
    the tensor gradients after ONE real backward pass, and the parameter gradients of every conv / BatchNorm against
    autograd on the GPU's own activations and output gradients."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=81)
    x, target = O.synthetic_batch(24, 68, 256, seed=82)
    from cu_net_amd._lib import set_planner_option
    set_planner_option('f32_split', split)
    try:
        _check_composition(cfg, st, x, target, check_params=True)      # (incl. every node's FORWARD at N = 24 against torch on the GPU's own inputs)
    finally:
        set_planner_option('f32_split', 1)      # (the default)


@pytest.mark.parametrize('mode', [True, 2])
def test_every_node_bench_batch_bf16(mode):
    """BASELINE config 3's kernels in the regime bench.py runs them in -- bf16 storage (mode True: bf16 activations; 2: bf16 gradient
    tensors too), N = 24 at production widths (CU-Net-2, K = 68: the same node shapes as CU-Net-8): at 64 x 64 a wave of
    conv_bf16[_pair]_kernel / dgrad_bf16[_pair]_kernel walks three tiles (the cross-tile logic: next tile's dY behind the epilogue's x
    requests, one-pass LDS epilogue), conv3x3_ring_bf16_kernel runs several strips per workgroup with ring reuse, the bf16 weight
    gradients their steady-state loops.  Node by node against torch on identical (bf16-rounded) inputs: forward output, input
    gradients, weight and BatchNorm parameter gradients (models/cu_net.py:11-17,43-48; cu-net.py:182)."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=85)
    x, _ = O.synthetic_batch(24, 68, 256, seed=86)
    _check_all_nodes(cfg, st, x, bf16=mode, wgrad3_all=True, check_forward=True)


@pytest.mark.parametrize('mode', [False, 2])
def test_weight_gradient_steady_state_loops(mode):
    """The LDS-staged weight gradients with FEW, LONG splits (planner options wgrad3_max_splits = 8 and the bf16 pair): at N = 2
    a workgroup of the 64 x 64 nodes owns 1024 pixels = 32 chunks (1x1: wgrad3_kernel / wgrad3_bf16_kernel) or 16 image rows
    (3x3: wgrad3_3x3_kernel / wgrad3_3x3_bf16_kernel), i.e. the steady state of their double-buffered loops, which the default
    split policy only reaches at bench batch sizes.  Conv nodes only, against autograd on identical inputs."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=83)
    x, _ = O.synthetic_batch(2, 68, 256, seed=84)
    saved = {'wgrad3_max_splits': 0, 'wgrad3_max_splits_bf16': 128}      # (the defaults; 0 = by f32_split)
    for k in saved:
        set_planner_option(k, 8)
    try:
        desc = _check_all_nodes(cfg, st, x, bf16=mode, wgrad3_all=True, only_ops=('conv',))
    finally:
        for k, v in saved.items():
            set_planner_option(k, v)
    key = 'wg3_bf16' if mode == 2 else 'wg3'
    big = [nd for nd in desc['nodes'] if nd['op'] == 'conv' and nd.get(key, 0) > 0 and desc['tensors'][nd['out']]['W'] == 64]
    assert big and all(nd[key] <= 8 for nd in big), [nd[key] for nd in big]


def _deep_plan_filter(layer_num, stride):
    """Nodes of a deep plan worth a CPU autograd pass at N = 24: every node of U-Net 0 and of the last U-Net (whose tensors sit at the far
    end of the workspace), every node one of whose tensors straddles a multiple of 2^30 floats (4 GB: the offsets that no longer fit 32
    bits), and every `stride`-th node of the U-Nets in between (their shapes repeat U-Net 1's; what differs are the offsets and buckets)."""
    def f(k, nd, desc):
        T = desc['tensors']
        b = nd.get('bucket', 0)
        if b in (0, layer_num - 1, layer_num):
            return True
        for i in [nd['out']] + [sg['t'] for sg in nd['segs']]:
            t = T[i]
            n = t['N'] * t['H'] * t['W'] * t['ld']
            for o in (t['act'], t['grad']):
                if o >= 0 and (o >> 30) != ((o + n) >> 30):
                    return True
        return k % stride == 0
    return f


def test_every_node_bench_batch_cu_net8_bf16_grads():
    """BASELINE config 3's PLAN at the batch bench.py times it at -- CU-Net-8, K = 68, N = 24, bf16 activations and gradient tensors: nine
    gradient buckets, a 10 GB workspace whose float offsets pass 2^31.  Node by node against autograd on the GPU's own activations
    (forward output, input gradients, weight and BatchNorm parameter gradients): U-Net 0, U-Net 7, the stem, every node next to a 4 GB
    boundary and every 6th node in between (models/cu_net.py:11-17,43-48,115-144; cu-net.py:182)."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=8, order=1, loss_num=8)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=91)
    x, _ = O.synthetic_batch(24, 68, 256, seed=92)
    _check_all_nodes(cfg, st, x, bf16=2, wgrad3_all=True, check_forward=True, node_filter=_deep_plan_filter(8, 6))


def test_every_node_bench_batch_cu_net16_binary_weights():
    """BASELINE config 5's plan at the bench batch -- CU-Net-16, K = 16, N = 24, fp32, with the weights QuanOp(bits_w = 1).quantization()
    leaves in the arena (utils/quantize.py:125-149; the step bench.py times runs exactly these kernels on exactly such weights): 17
    buckets, a 15 GB workspace.  The quantised weights are read back and the same node-by-node autograd check runs on them: U-Net 0,
    U-Net 15, the stem, 4 GB-boundary nodes and every 12th node in between."""
    from cu_net_amd.quant import QuanOp
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=16, order=1, loss_num=16)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=93)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    QuanOp(net, bits_w=1, bits_i=8, bits_g=8).quantization()
    torch.cuda.synchronize()
    stq = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    changed = sum(1 for k in st if k.endswith('conv2.weight') and not torch.equal(st[k], stq[k]))
    assert changed >= 9 * 16, 'quantization() left the 3x3 weights untouched'
    del net
    torch.cuda.empty_cache()
    x, _ = O.synthetic_batch(24, 16, 256, seed=94)
    _check_all_nodes(cfg, stq, x, wgrad3_all=True, check_forward=True, node_filter=_deep_plan_filter(16, 12))
