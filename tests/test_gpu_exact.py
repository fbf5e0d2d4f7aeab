"""GPU (-m gpu): small exactness checks of the hot path's non-GEMM pieces, each on identical inputs.

  * fused RMSprop on a flat arena vs torch.optim.RMSprop (cu-net.py:60-61,183), three steps, <= 2 ulp;
  * MaxPool2d(2,2) forward and the nearest-Upsample(2) gather folded into the consumer's loads
    (models/cu_net.py:249-250): bit-exact (`torch.equal`) -- north_star: "bit-exact for the upsample index maps";
  * whole-network backward against an fp64 evaluation of the same plan: the HIP path's per-tensor error must stay
    in line with the error torch's own fp32 CPU path shows against fp64 (median ratio <= 1.5, no tensor beyond 6x;
    whole-net fp32 gradients are ill-conditioned, so the yardstick is fp64, not another fp32 result).
"""
import os

import pytest
import torch
import torch.nn.functional as F

import cu_net_amd
from cu_net_amd._lib import check, lib
from cu_net_amd.module import _ptr, _stream_ptr
from cu_net_amd.trainer import FusedTrainer
from oracle import cunet_ref as O
from tests._golden import Golden
from tests._plan_interp import run_plan

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DGRAD3_NT_DEFAULT = 1      # planner option dgrad3_nt as plan.h ships it


@pytest.mark.parametrize('n,gscale', [(1 << 20, 1.0), (100003, 0.125), (7, 1.0)])
def test_rmsprop_step_matches_torch(n, gscale):
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g) * 0.1
    grads = [torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g))) for _ in range(3)]
    grads[1][::5] = 0.0                                        # exact zeros: sqrt(v) + eps path
    lr, alpha, eps = 2.5e-4, 0.99, 1e-8
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([ref], lr=lr, alpha=alpha, eps=eps, momentum=0, weight_decay=0)
    p = p0.clone().cuda()
    v = torch.zeros(n, device='cuda')
    carried = torch.zeros(n)                                   # rounding of the earlier updates stays in p
    for step, gr in enumerate(grads):
        ref.grad = (gr * gscale).clone()                       # 1/world folded into the kernel == scaling the gradient first
        opt.step()
        gd = gr.cuda()
        check(lib().cunet_rmsprop_step(_ptr(p), _ptr(gd), _ptr(v), n, lr, alpha, eps, gscale, _stream_ptr(p.device)),
              'cunet_rmsprop_step')
        torch.cuda.synchronize()
        vref = opt.state[ref]['square_avg']
        ulp = torch.finfo(torch.float32).eps
        dv = (v.cpu() - vref).abs()
        assert bool((dv <= 2 * ulp * vref.abs() + 1e-45).all()), float((dv / (vref.abs() + 1e-30)).max())
        # p' = p - lr * g / (sqrt(v) + eps): 1 ulp of p + 16 eps (2e-6 relative) of every update applied so far (torch divides
        # (lr * g) / (sqrt(v) + eps), the kernel multiplies lr * (g / (sqrt(v) + eps)); v itself is within 2 ulp)
        dp = (p.cpu() - ref.detach()).abs()
        carried += 16 * ulp * lr * (gr * gscale).abs() / (vref.sqrt() + eps)
        bound = (step + 1) * ulp * ref.detach().abs() + carried + 1e-45      # one rounding of p per step on either side
        assert bool((dp <= bound).all()), float((dp / bound).max())


def _toy_cfg():
    return dict(neck_size=2, growth_rate=16, init_chan_num=32, class_num=6, layer_num=2, order=1, loss_num=2)


@pytest.mark.parametrize('full', [False, True])
def test_pool_forward_and_upsample_gather_bit_exact(full):
    """Pool: every pool node's output equals F.max_pool2d of the node's own input, bit for bit.
    Up-sample gather: every conv node that reads a segment through the nearest-upsample index map is turned into an
    exact channel selection (BatchNorm made the identity in eval mode, 1x1 weights a 0/1 selection matrix), so its
    output must EQUAL relu(cat(upsample(x), skip))[:, sel] computed by torch on the tensors the GPU produced."""
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2) if full else _toy_cfg()
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=51)
    hw = 256 if full else 128
    x, _ = O.synthetic_batch(2, cfg['class_num'], hw, seed=52)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net.cuda().eval()
    plan = net._get_plan(2, hw, hw, False)
    d = plan.handle.describe()
    T = d['tensors']
    ups_nodes = [nd for nd in d['nodes'] if nd['op'] == 'conv' and nd['taps'] == 1 and any(s['ups'] for s in nd['segs'])]
    assert len(ups_nodes) >= 4 * cfg['layer_num']          # the first layer / adapters of every up block
    sd = net.state_dict()
    sels = {}
    with torch.no_grad():
        for nd in ups_nodes:
            ccat = sum(T[s['t']]['C'] for s in nd['segs'])
            w = sd[nd['conv'] + '.weight']
            cout = w.shape[0]
            sel = [(o * 5 + 3) % ccat for o in range(cout)]           # hits every segment
            sels[nd['name']] = sel
            w.zero_()
            for o, c in enumerate(sel):
                w[o, c, 0, 0] = 1.0
            sd[nd['bn'] + '.weight'].fill_(1.0)
            sd[nd['bn'] + '.bias'].zero_()
            sd[nd['bn'] + '.running_mean'].zero_()
            sd[nd['bn'] + '.running_var'].fill_(1.0 - 1e-5)           # var + eps == 1 to within half an ulp: scale == 1.0f
    with torch.no_grad():
        net(x.cuda())
    torch.cuda.synchronize()
    npool = 0
    for nd in d['nodes']:
        if nd['op'] == 'pool':
            a = plan.debug_tensor(T[nd['segs'][0]['t']]['name']).cpu()
            y = plan.debug_tensor(T[nd['out']]['name']).cpu()
            assert torch.equal(y, F.max_pool2d(a, 2, 2)), nd['name']
            npool += 1
    assert npool == 4 * cfg['layer_num']
    for nd in ups_nodes:
        parts = []
        for s in nd['segs']:
            a = plan.debug_tensor(T[s['t']]['name']).cpu()
            parts.append(F.interpolate(a, scale_factor=2, mode='nearest') if s['ups'] else a)
        cat = torch.relu(torch.cat(parts, 1))
        y = plan.debug_tensor(T[nd['out']]['name']).cpu()
        assert torch.equal(y, cat[:, sels[nd['name']]]), nd['name']


@pytest.mark.parametrize('split', [0, 1])
@pytest.mark.parametrize('tag', ['G9_L2_o1_c32', 'G1_L2_o1', 'G2_L3_o2'])
def test_backward_error_vs_fp64_tracks_torch_fp32(tag, split):
    """Every gradient tensor and parameter gradient of a train step against the fp64 run of the same graph, with torch's fp32 CPU
    error on the same tensor as the yardstick.  split = 1: the convolutions contract on the bf16 matrix pipe with every fp32 operand
    value cut into three bf16 pieces (planner option f32_split, conv_body's XBG = 6) -- the same bound: that path is an fp32 path."""
    from cu_net_amd._lib import set_planner_option
    set_planner_option('f32_split', split)
    try:
        _backward_error_vs_fp64(tag, '_split' if split else '')
    finally:
        set_planner_option('f32_split', 1)      # (the default)


def _backward_error_vs_fp64(tag, suffix):
    g = Golden(tag)
    x, target = g.t('x'), g.t('target')
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(g.group('state0'))
    net = net.cuda().train()
    tr = FusedTrainer(net)
    n, _, h, w = x.shape
    plan = net._get_plan(n, h, w, True)
    desc = plan.handle.describe()

    def ref(dtype):
        st = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in g.group('state0').items()}
        for k in st:
            if st[k].is_floating_point() and 'running' not in k:
                st[k].requires_grad_(True)
        outs, acts, grads, loss = run_plan(desc, st, x.to(dtype), True, True, target.to(dtype))
        pg = {k: v.grad for k, v in st.items() if v.is_floating_point() and v.grad is not None}
        return acts, grads, pg

    a64, g64, p64 = ref(torch.float64)
    a32, g32, p32 = ref(torch.float32)
    tr.step(x.cuda(), target.cuda())
    torch.cuda.synchronize()

    def rel2(a, b):
        return ((a.double() - b).norm() / (b.norm() + 1e-300)).item()

    lines, bad = [], []
    worst_ratio = 0.0
    rows = []
    for t in reversed(desc['tensors']):
        nm = t['name']
        if nm in g64:
            rows.append(('grad ' + nm, rel2(plan.debug_tensor(nm, grad=True).cpu(), g64[nm]), rel2(g32[nm], g64[nm])))
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
    for k, v in p64.items():
        o, nmel, shape = off[k]
        rows.append(('dpar ' + k, rel2(net._grad_arena[o:o + nmel].view(shape).cpu(), v), rel2(p32[k], v)))
    # the yardstick: torch fp32's own error on this tensor, floored by its median over all tensors (a tensor on which
    # the CPU path happens to land within 1e-7 of fp64 is luck, not a bound).  The errors are not smooth: they jump when
    # one ReLU / max-pool decision flips on a 1e-7 difference (in the G9 report both fp32 paths jump from 3e-5 to 1.4e-2
    # at the same tensor), and the two fp32 paths do not flip the same elements.  So: the MEDIAN of hip/cpu32 over all
    # tensors must be <= 1.5 (no systematic loss of accuracy), and no single tensor may be beyond 6x.
    med = sorted(r[2] for r in rows)[len(rows) // 2]
    ratios = []
    for name, eh, ec in rows:
        ratio = eh / max(ec, med, 1e-30)
        ratios.append(ratio)
        worst_ratio = max(worst_ratio, ratio)
        ok = eh <= 6.0 * max(ec, med) + 1e-6
        lines.append(f'{"ok " if ok else "BAD"} {name:70s} hip_vs_f64={eh:.3e} cpu32_vs_f64={ec:.3e}')
        if not ok:
            bad.append(name)
    med_ratio = sorted(ratios)[len(ratios) // 2]
    lines.append(f'hip/cpu32 error ratio: median {med_ratio:.2f}, worst {worst_ratio:.2f} (median cpu32 error {med:.2e})')
    if med_ratio > 1.5:
        bad.append(f'median ratio {med_ratio:.2f}')
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', f'backward_vs_f64_{tag}{suffix}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    except OSError:
        pass
    assert not bad, (len(bad), bad[:5], lines[-1])


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16_grads'])
@pytest.mark.parametrize('class_num', [68, 3])
def test_fused_head_loss_equals_the_separate_loss_pass(mode, class_num):
    """cunet_loss_mse_fused (MSE and d(loss)/d(out) in the head convolutions' epilogues) against cunet_loss_mse after the same
    forward: same loss (fp64 partial sums in another order), identical d(loss)/d(out) -- so identical parameter gradients up to
    the atomics of the kernels behind it.  class_num = 3: a head tensor with a pad column (ld 4), which must read as zero."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=class_num, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=61)
    x, target = O.synthetic_batch(2, class_num, 256, seed=62)
    res = {}
    for fused in (False, True):
        net = cu_net_amd.create_cu_net(**cfg)
        net.load_state_dict(st)
        net = net.cuda().train()
        plan = net._get_plan(2, 256, 256, True, bf16=mode != 'fp32')
        xd, td = x.cuda(), target.cuda()
        if fused:
            loss = plan.stage_target(td)
        if mode == 'fp32':
            plan.forward(xd, True, want_outputs=False)
        else:
            plan.forward_bf16(xd, 2 if mode == 'bf16_grads' else 1, want_outputs=False)
        if not fused:
            loss = plan.loss_mse(td)
        torch.cuda.synchronize()
        heads = sorted((nd['head'], plan.handle.describe()['tensors'][nd['out']]['name'])
                       for nd in plan.handle.describe()['nodes'] if nd.get('head', -1) >= 0)
        dout = [plan.debug_tensor(name, grad=True).cpu() for _, name in heads]
        plan.backward(None)
        torch.cuda.synchronize()
        res[fused] = (float(loss), dout, net._grad_arena.clone().cpu())
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        # (two separate forwards: the fp64 statistics atomics of the fp32 nodes commit in another order, so the heat maps of the
        # two runs agree to a few ulp, not bit for bit)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    ga, gb = res[True][2], res[False][2]
    assert float((ga - gb).norm() / gb.norm()) <= 1e-5


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16_grads'])
def test_adapter_pair_launches_equal_the_single_launches(mode):
    """The ahead and the skip adapter of a down block in ONE launch (conv_pair_kernel / conv1x1_splitk_pair_kernel /
    conv_bf16_pair_kernel forward, conv_pair_kernel<LD_PLAIN, EP_BWD> / dgrad_bf16_pair_kernel data gradient; planner option
    pair_adapters, models/cu_net.py:139-142) against one launch each on the same state: the bodies are the same device functions, so
    the adapters' outputs agree to the order of the statistics atomics in front of them, and so do loss and parameter gradients.
    The launch counts show that the pairs ran: 4 per U-Net fewer 1x1 forwards, and as many fewer 1x1 data gradients where the
    storage mode has a pair kernel (fp32 and bf16 gradient tensors; bf16 activations with fp32 gradients run them one by one)."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    L = 2
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=L, order=1, loss_num=L)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=71)
    x, target = O.synthetic_batch(4, 16, 256, seed=72)
    res = {}
    try:
        for pair in (1, 0):
            set_planner_option('pair_adapters', pair)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(4, 256, 256, True, bf16=mode != 'fp32')
            desc = plan.handle.describe()
            assert sum(nd.get('pair', 0) for nd in desc['nodes']) == (4 * L if pair else 0)
            plan.handle.profile_begin(1)
            plan.handle.profile_reset()
            loss = plan.stage_target(target.cuda())
            if mode == 'fp32':
                plan.forward(x.cuda(), True, want_outputs=False)
            else:
                plan.forward_bf16(x.cuda(), 2 if mode == 'bf16_grads' else 1, want_outputs=False)
            torch.cuda.synchronize()
            outs = {}
            for nd in desc['nodes']:
                if '.adapters_ahead.' in nd['name'] or '.adapters_skip.' in nd['name']:
                    outs[nd['name']] = plan.debug_tensor(desc['tensors'][nd['out']]['name']).float().cpu()
            plan.backward(None)
            torch.cuda.synchronize()
            counts = {k: v[0] for k, v in plan.handle.profile_collect().items()}
            plan.handle.profile_begin(0)
            res[pair] = (float(loss), outs, net._grad_arena.clone().cpu(), counts)
    finally:
        set_planner_option('pair_adapters', 1)
    sfx = '' if mode == 'fp32' else '_bf16'
    fwd = 'conv1x1_fwd' + sfx
    assert res[0][3][fwd] - res[1][3][fwd] == 4 * L, (res[0][3][fwd], res[1][3][fwd])
    bwd1 = sum(res[1][3].get(k, 0) for k in ('conv1x1_bwd_data', 'conv1x1_bwd_data_bf16'))
    bwd0 = sum(res[0][3].get(k, 0) for k in ('conv1x1_bwd_data', 'conv1x1_bwd_data_bf16'))
    assert bwd0 - bwd1 == (0 if mode == 'bf16' else 4 * L), (bwd0, bwd1)
    assert abs(res[1][0] - res[0][0]) <= 1e-5 * abs(res[0][0])
    first = sorted(res[0][1])[0]
    for name, b in res[0][1].items():
        a = res[1][1][name]
        # (the first pair sees bit-identical inputs up to the statistics atomics of the nodes in front of it; bf16 outputs may
        # round the other way at a rounding boundary: one bf16 step on a few elements)
        tol = 1e-5 if mode == 'fp32' else 2 ** -7
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))
        if mode != 'fp32' and name == first:
            assert float(((a - b).abs() > 0).float().mean()) <= 2e-3, name
    # (gradients: the pair launcher picks its kernel for half of the chip -- at 32 x 32 the single launches run the split-K kernel,
    # the pair the weight-stationary one -- so activations differ by summation order, ~1e-7 relative, and the ReLU masks of
    # the elements that close to zero flip: a fraction f of flipped elements moves the gradient by ~sqrt(f) in relative L2.  The
    # tight checks of the pair kernels are the node tests (cunet_debug_run_node_backward runs an adapter's data gradient in the
    # pair's launch) and the composition tests of tests/test_gpu_nodes.py.)
    ga, gb = res[1][2], res[0][2]
    assert float((ga - gb).norm() / gb.norm()) <= (1e-2 if mode == 'fp32' else 3e-2)


def test_dgrad_channel_tiles_per_wave_agree():
    """fp32 1x1 data gradient with 1, 2, 3 and 4 channel tiles of dz per wave (planner option dgrad_nt; 10 + k = "as many as fit up to
    k" whatever the launcher's balance model prefers): the same MFMA chains over K per element, so dz is bit-identical; only the order
    of the fp64 BatchNorm reductions differs.  BASELINE config 2's shapes (N = 24: below that the launcher keeps one tile per wave),
    whole backward, every parameter gradient and the stem's input-side gradient tensor against the one-tile run.  Covers the slices of
    unequal tile counts (5 tiles = 3 + 2: row blocks dealt in proportion), the pair launches of the adapters and the two-waves-per-SIMD
    instantiations (models/cu_net.py:11-17 backward; cu-net.py:182)."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=87)
    x, target = O.synthetic_batch(24, 68, 256, seed=88)
    xd, td = x.cuda(), target.cuda()
    res = {}
    try:
        for opt in (1, 12, 13, 14, 4, 'pf2'):
            set_planner_option('dgrad_nt', 1 if opt == 'pf2' else opt)
            set_planner_option('dgrad_prefetch', 2 if opt == 'pf2' else 1)      # ('pf2': two chunks of dY in flight per wave, conv_body's PF2 loop)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(24, 256, 256, True)
            loss = plan.stage_target(td)
            plan.forward(xd, True, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            d = plan.handle.describe()
            first_pool = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][1]['out']][0]      # the stem's pooled output
            res[opt] = (float(loss), net._grad_arena.clone(), plan.debug_tensor(first_pool, grad=True))
            del plan, net
    finally:
        set_planner_option('dgrad_nt', 4)
        set_planner_option('dgrad_prefetch', 1)
    l0, g0, t0 = res[1]
    assert torch.isfinite(g0).all() and float(g0.norm()) > 0
    for opt in (12, 13, 14, 4, 'pf2'):
        l1, g1, t1 = res[opt]
        assert abs(l1 - l0) <= 1e-6 * abs(l0), (opt, l1, l0)      # the forward does not depend on the option (fp64 atomics order only)
        assert float((g1 - g0).norm() / g0.norm()) <= 1e-5, (opt, float((g1 - g0).norm() / g0.norm()))
        assert float((t1 - t0).abs().max()) <= 1e-5 * float(t0.abs().max()), (opt, float((t1 - t0).abs().max()), float(t0.abs().max()))


@pytest.mark.parametrize('split', [1, 0])
def test_3x3_data_gradient_channel_tiles_per_wave_agree(split):
    """Planner option dgrad3_nt: the 3x3 data gradient with 2 and 4 channel tiles of dz per wave (the nine shifted taps of dY gathered -- and
    on the split contraction cut -- once per 64 / 128 output channels instead of once per 32) against one tile per wave: the same chain of
    products per element, so dz is bit-identical; parameter gradients agree to the order of the fp64 BatchNorm reductions.  BASELINE
    config 2's shapes, whole backward (models/cu_net.py:45-48 backward)."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=93)
    x, target = O.synthetic_batch(24, 68, 256, seed=94)
    xd, td = x.cuda(), target.cuda()
    res = {}
    try:
        set_planner_option('f32_split', split)
        for opt in (1, 2, 4):
            set_planner_option('dgrad3_nt', opt)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(24, 256, 256, True)
            loss = plan.stage_target(td)
            plan.forward(xd, True, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            d = plan.handle.describe()
            first_pool = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][1]['out']][0]
            res[opt] = (float(loss), net._grad_arena.clone(), plan.debug_tensor(first_pool, grad=True))
            del plan, net
    finally:
        set_planner_option('dgrad3_nt', DGRAD3_NT_DEFAULT)
        set_planner_option('f32_split', 1)      # (the default)
    l0, g0, t0 = res[1]
    assert torch.isfinite(g0).all() and float(g0.norm()) > 0
    for opt in (2, 4):
        l1, g1, t1 = res[opt]
        assert abs(l1 - l0) <= 1e-6 * abs(l0), (opt, l1, l0)
        assert float((g1 - g0).norm() / g0.norm()) <= 1e-5, (opt, float((g1 - g0).norm() / g0.norm()))
        assert float((t1 - t0).abs().max()) <= 1e-5 * float(t0.abs().max()), (opt, float((t1 - t0).abs().max()), float(t0.abs().max()))


def test_split_contraction_agrees_with_the_fp32_matrix_pipe():
    """Planner option f32_split: the fp32 convolutions (1x1 forward, heads with the fused loss, 1x1 / 3x3 data gradients, adapter pairs)
    contract on the bf16 matrix pipe -- x = h + m + l in bf16 pieces, six products per pair of operands.  Not bit-identical to the fp32
    MFMA (other roundings of the same sums) but the same arithmetic to fp32 accuracy: BASELINE config 2's shapes, one whole train step,
    loss / every parameter gradient / the stem-side gradient tensor against the fp32-pipe run at the tolerance two fp32 summation
    orders differ by (models/cu_net.py:11-17, 45-48 and their autograd; cu-net.py:175-182)."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=91)
    x, target = O.synthetic_batch(24, 68, 256, seed=92)
    xd, td = x.cuda(), target.cuda()
    res = {}
    try:
        for split in (0, 1):
            set_planner_option('f32_split', split)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(24, 256, 256, True)
            loss = plan.stage_target(td)
            outs = plan.forward(xd, True, want_outputs=True)
            plan.backward(None)
            torch.cuda.synchronize()
            d = plan.handle.describe()
            first_pool = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][1]['out']][0]
            res[split] = (float(loss), net._grad_arena.clone(), plan.debug_tensor(first_pool, grad=True), [o.clone() for o in outs])
            del plan, net
    finally:
        set_planner_option('f32_split', 1)      # (the default)
    l0, g0, t0, o0 = res[0]
    l1, g1, t1, o1 = res[1]
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    e_loss = abs(l1 - l0) / abs(l0)
    e_max = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(o1, o0))
    e_nrm = max(float((a - b).norm() / b.norm()) for a, b in zip(o1, o0))
    e_par = float((g1 - g0).norm() / g0.norm())
    e_stem = float((t1 - t0).abs().max() / t0.abs().max())
    print(f'split vs fp32 pipe: loss {e_loss:.2e}, heat maps max {e_max:.2e} / norm {e_nrm:.2e}, parameter gradients {e_par:.2e}, stem-side gradient {e_stem:.2e}')
    # Two fp32 evaluations of the same step: the heat maps agree to 2e-5 of their range (BASELINE.json's bar: 1e-4).  The gradients of a
    # whole backward do NOT agree to fp32 accuracy between ANY two fp32 paths: a ReLU / max-pool decision that flips on a 1e-7 difference
    # moves a gradient tensor by 1e-2 (test_backward_error_vs_fp64_tracks_torch_fp32 sees torch's CPU fp32 and this library jump from 3e-5
    # to 1.4e-2 at the same tensor) -- here 5e-3 on the parameter gradients.  What bounds the split path's accuracy is that test (error
    # against fp64 as small as the fp32 pipe's, tensor by tensor) and the node-by-node check at this batch
    # (test_whole_backward_composition_bench_batch[1]); this one catches an instantiation that only runs at N = 24 going wrong grossly.
    assert e_loss <= 2e-6 and e_max <= 1e-4 and e_nrm <= 5e-5 and e_par <= 3e-2 and e_stem <= 0.3, (e_loss, e_max, e_nrm, e_par, e_stem)


@pytest.mark.parametrize('n', [4, 24])
def test_bf16_weight_gradient_on_lds_dma_is_bit_identical(n):
    """wgrad4_bf16_kernel (round 4: dY and x streamed into an LDS ring by global_load_lds_dwordx4, MFMA operands by
    ds_read_b64_tr_b16, BatchNorm + ReLU on the transposed fragment) against wgrad3_bf16_kernel (operands transposed through VGPRs
    on the way into LDS) on the SAME plan state, node by node: the two contract the same bf16 operands in the same order, so every
    1x1 weight gradient must agree bit for bit.  N = 4: few splits, 2 - 8 ring slots per workgroup (shorter than the ring: its
    prologue / tail paths); N = 24: the bench's geometry (96 splits of 1024 pixels = 32 slots, the ring wraps).  Up-sampled
    segments (up blocks), split-K (<= 5 channel tiles) and two-half ownerships incl. odd tile counts (7, 9) all occur in CU-Net-2.
    Planner option wgrad_bf16_dma; autograd wgrad of models/cu_net.py:24,43."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=93)
    x, _ = O.synthetic_batch(n, 16, 256, seed=94)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    plan = net._get_plan(n, 256, 256, True, bf16=True)
    plan.forward_bf16(x.cuda(), 2, want_outputs=False)
    torch.cuda.synchronize()
    desc = plan.handle.describe()
    T = desc['tensors']
    off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
    nodes = [(k, nd) for k, nd in enumerate(desc['nodes']) if nd['op'] == 'conv' and nd['taps'] == 1 and nd.get('wg3_bf16', 0) > 0]
    assert len(nodes) >= 40
    seen_ct, bad = set(), []
    try:
        for k, nd in nodes:
            o, nmel = off[nd['conv'] + '.weight']
            t = T[nd['out']]
            gen = torch.Generator().manual_seed(2000 + k)
            plan.debug_poke(t['name'], torch.randn((t['N'], t['C'], t['H'], t['W']), generator=gen), grad=True)      # this node's d(loss)/d(out)
            got = {}
            for dma in (1, 0):
                plan.debug_set_option('wgrad_bf16_dma', dma)        # (the plan's own snapshot: nothing process-wide changes)
                plan.debug_run_node_backward(k)
                torch.cuda.synchronize()
                got[dma] = net._grad_arena[o:o + nmel].clone()
            seen_ct.add(nmel // (128 * 32))
            if not torch.equal(got[1], got[0]):
                d = (got[1] - got[0]).abs()
                bad.append(f'{nd["name"]}: {int((d > 0).sum())}/{nmel} elements differ, max {float(d.max()):.3e} of {float(got[0].abs().max()):.3e}')
            assert float(got[0].abs().max()) > 0
    finally:
        plan.debug_set_option('wgrad_bf16_dma', 1)
    assert not bad, '\n'.join(bad[:20])
    assert {4, 5, 6, 8, 9, 10} <= seen_ct, seen_ct


@pytest.mark.parametrize('n', [2, 24])
def test_fused_weight_gradient_equals_the_separate_launches(n):
    """Planner option fuse_wgrad = 1: the fp32 data gradient of a 1x1 node also contracts dY^T with relu(bn(x)) -- both tiles are in
    the wave's hands -- and writes partial weight-gradient tiles that the bucket's reduce sums; 0 (the default: measured faster in the
    overlapped step): the node's own wgrad3 launch on the side stream as in rounds 2-3.  Same state, same batch, whole backward: every parameter gradient agrees to fp32 summation order, the
    BatchNorm / tensor gradients likewise (the data-gradient arithmetic is the same code).  N = 24 is the bench's geometry (two partial
    regions alternating by bucket, the caller's stream waiting for the reduce two buckets back); N = 2 has single-tile waves and the
    ragged levels that keep their own launch (4 x 4: 32 rows).  Also: the fused launches really ran (no 1x1 weight-gradient launches
    left except the heads').  autograd wgrad + dgrad of models/cu_net.py:24,43; cu-net.py:182."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=95)
    x, target = O.synthetic_batch(n, 68, 256, seed=96)
    xd, td = x.cuda(), target.cuda()
    res = {}
    try:
        set_planner_option('f32_split', 0)      # (the fused tile loop is an fp32-matrix-pipe kernel: both arms on that pipe, launch for launch)
        for fuse in (0, 1):
            set_planner_option('fuse_wgrad', fuse)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(n, 256, 256, True)
            desc = plan.handle.describe()
            n1x1 = sum(1 for nd in desc['nodes'] if nd['op'] == 'conv' and nd['taps'] == 1 and nd.get('head', -1) < 0)
            assert n1x1 == 45 and sum(nd.get('fuse_wgrad', 0) for nd in desc['nodes']) == (n1x1 if fuse else 0)      # every 1x1 conv of CU-Net-2 but the two heads
            plan.handle.profile_begin(1)
            plan.handle.profile_reset()
            loss = plan.stage_target(td)
            plan.forward(xd, True, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            counts = {k: v[0] for k, v in plan.handle.profile_collect().items()}
            plan.handle.profile_begin(0)
            first_pool = [t['name'] for t in desc['tensors'] if t['id'] == desc['nodes'][1]['out']][0]
            res[fuse] = (float(loss), net._grad_arena.clone(), plan.debug_tensor(first_pool, grad=True), counts)
            del plan, net
    finally:
        set_planner_option('fuse_wgrad', 0)
        set_planner_option('f32_split', 1)      # (the default)
    (l0, g0, t0, c0), (l1, g1, t1, c1) = res[0], res[1]
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert torch.isfinite(g1).all()
    assert float((g1 - g0).norm() / g0.norm()) <= 2e-5, float((g1 - g0).norm() / g0.norm())
    assert float((g1 - g0).abs().max()) <= 2e-4 * float(g0.abs().max()), (float((g1 - g0).abs().max()), float(g0.abs().max()))
    assert float((t1 - t0).abs().max()) <= 1e-5 * float(t0.abs().max())
    assert c0['conv1x1_bwd_weight'] >= 40
    assert c1['conv1x1_bwd_weight'] <= c0['conv1x1_bwd_weight'] - (40 if n == 2 else 45), (c0['conv1x1_bwd_weight'], c1['conv1x1_bwd_weight'])
    assert c1['conv1x1_bwd_data'] == c0['conv1x1_bwd_data']


@pytest.mark.parametrize('split', [1, 0])
def test_row_tile_data_gradient_agrees_with_the_sliced_kernel(split):
    """(split: planner option f32_split -- 1: dgrad1x1_rows_split_kernel, the tile cut into its bf16 operand planes once per workgroup,
    against conv_body's split mode, which cuts it once per column slice: the same six products in the same order per element.)
    dgrad1x1_rows_kernel (planner option dgrad_rows; round 4: all input channels of a row tile in one workgroup, weights in registers, dY
    staged once per workgroup by LDS-DMA in MFMA fragment order) against the column-sliced kernel (dgrad_rows = 0) on BASELINE config 2's
    shapes, whole backward: the same k-ordered MFMA chain per element, so dz is bit-identical and everything downstream agrees to the
    order of the fp64 BatchNorm reductions.  dgrad_rows = 1 sends EVERY eligible launch to the new kernel (grids smaller than the chip,
    one tile per workgroup), 512 only the 64 x 64 and 32 x 32 levels (the ring wraps, several tiles per workgroup).  The shipped default
    is 0: the kernel measured equal alone and slower in the step.
    Round 5: with f32_split the row-tile kernel is dgrad1x1_rows_split2_kernel (planner option dgrad_rows_v = 2: counted waits that leave
    the previous tile's stores in flight, x by asm requests, plane reads one step ahead, BatchNorm sums in registers); the round-4 kernel
    (dgrad_rows_v = 1) runs here as a fourth arm.  All arms: same products in the same order per element."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=97)
    x, target = O.synthetic_batch(24, 68, 256, seed=98)
    xd, td = x.cuda(), target.cuda()
    res = {}
    try:
        set_planner_option('f32_split', split)
        for opt in (0, 1, 512) + ((1001,) if split else ()):      # 1001: every launch on the ROUND-4 row-tile kernel (dgrad_rows_v = 1)
            set_planner_option('dgrad_rows', opt % 1000)
            set_planner_option('dgrad_rows_v', 1 if opt >= 1000 else 2)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(24, 256, 256, True)
            loss = plan.stage_target(td)
            plan.forward(xd, True, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            d = plan.handle.describe()
            first_pool = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][1]['out']][0]
            res[opt] = (float(loss), net._grad_arena.clone(), plan.debug_tensor(first_pool, grad=True))
            del plan, net
    finally:
        set_planner_option('dgrad_rows', -1)     # (the default: by f32_split)
        set_planner_option('f32_split', 1)      # (the default)
    l0, g0, t0 = res[0]
    assert torch.isfinite(g0).all() and float(g0.norm()) > 0
    for opt in [k for k in res if k]:
        l1, g1, t1 = res[opt]
        assert abs(l1 - l0) <= 1e-6 * abs(l0), (opt, l1, l0)
        assert float((g1 - g0).norm() / g0.norm()) <= 1e-5, (opt, float((g1 - g0).norm() / g0.norm()))
        assert float((t1 - t0).abs().max()) <= 1e-5 * float(t0.abs().max()), (opt, float((t1 - t0).abs().max()), float(t0.abs().max()))


@pytest.mark.parametrize('mode', ['fp32', 'bf16_grads'])
def test_heads_on_the_side_stream_equal_heads_in_node_order(mode):
    """Planner option heads_on_side (default 2; 1 and 2 differ in where the last head's data gradient runs): in a training pass the heat-map heads run on the internal side stream -- forward
    (with the fused loss) next to the following U-Net, data and weight gradient up front at the start of backward -- instead of in
    node order on the caller's stream.  Same kernels, same arguments, only the stream differs: loss, every head's d(loss)/d(out)
    and the parameter gradients agree to the order of the fp64 atomics.  Three U-Nets, three heads; repeated so that a missing
    wait would have to hide three times."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=3, order=1, loss_num=3)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=81)
    x, target = O.synthetic_batch(2, 16, 256, seed=82)
    res = {}
    try:
        for side in (1, 2, 0):      # (2: the last U-Net's head runs its data gradient on the caller's stream, the rest as with 1)
            set_planner_option('heads_on_side', side)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(2, 256, 256, True, bf16=mode != 'fp32')
            desc = plan.handle.describe()
            heads = sorted((nd['head'], desc['tensors'][nd['out']]['name']) for nd in desc['nodes'] if nd.get('head', -1) >= 0)
            runs = []
            for rep in range(3 if side else 1):
                loss = plan.stage_target(target.cuda())
                if mode == 'fp32':
                    plan.forward(x.cuda(), True, want_outputs=False)
                else:
                    plan.forward_bf16(x.cuda(), 2, want_outputs=False)
                plan.backward(None)
                torch.cuda.synchronize()
                dout = [plan.debug_tensor(name, grad=True).float().cpu() for _, name in heads]
                runs.append((float(loss), dout, net._grad_arena.clone().cpu()))
            res[side] = runs
    finally:
        set_planner_option('heads_on_side', 2)      # (the default)
    ref = res[0][0]
    for got in res[1] + res[2]:
        assert abs(got[0] - ref[0]) <= 1e-5 * abs(ref[0])
        for a, b in zip(got[1], ref[1]):
            assert float((a - b).abs().max()) <= (1e-5 if mode == 'fp32' else 2 ** -7) * float(b.abs().max())
        rel = float((got[2] - ref[2]).norm() / ref[2].norm())
        assert rel <= (1e-3 if mode == 'fp32' else 3e-2), rel


@pytest.mark.parametrize('n,h,w', [(24, 256, 256), (3, 128, 256), (1, 256, 128)])
def test_stem_weight_gradient_with_the_dz_pass_fused(n, h, w):
    """Planner option stem_fuse_dz = 1 (default): wgrad3_stem_kernel<true> computes d(loss)/d(conv0 output) -- BatchNorm, ReLU and the 2 x 2
    max-pool backward of models/cu_net.py:300-303 -- while it stages its chunks, from conv0's output, the pooled features' gradient and
    the first stem pass's reductions; stem_bwd_kernel<1> and its N x 128 x H/2 x W/2 tensor drop out of the step.
    (1) Inside ONE run, bit for bit: after the fused backward the debug reader materialises that tensor (cunet_debug_materialise: the
    second stem pass on the state the backward left behind) and the UNFUSED weight-gradient kernel, run on it as a single node, must
    reproduce conv0's weight gradient exactly -- same values, same partial tiles, same reduce.
    (2) Against a second run with stem_fuse_dz = 0 (two stem passes, the weight gradient reads the tensor): conv0's weight gradient, the
    stem BatchNorm's parameter gradients and the tensor agree to fp32 rounding (the U-Nets in front of the stem ran on identical kernels,
    but their fp64 statistics atomics land in another order from run to run, so the stem's INPUT agrees to the last bits only).
    The bench batch, a rectangular batch (output rows of 64 / 128 pixels: one / two chunks per row) and a single image."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=101)
    gen = torch.Generator().manual_seed(102)
    x = torch.rand(n, 3, h, w, generator=gen)
    target = torch.rand(n, 16, h // 4, w // 4, generator=gen)
    xd, td = x.cuda(), target.cuda()
    names = ('features.conv0.weight', 'features.norm0.weight', 'features.norm0.bias')
    res = {}
    for fuse in (1, 0):
        set_planner_option('stem_fuse_dz', fuse)
        net = cu_net_amd.create_cu_net(**cfg)
        net.load_state_dict(st)
        net = net.cuda().train()
        plan = net._get_plan(n, h, w, True)
        plan.stage_target(td)
        plan.forward(xd, True, want_outputs=False)
        plan.backward(None)
        torch.cuda.synchronize()
        d = plan.handle.describe()
        assert d['nodes'][0]['op'] == 'stem_conv' and d['nodes'][0]['wg3'] > 0
        conv0_out = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][0]['out']][0]
        off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
        grads = {k: net._grad_arena[off[k][0]:off[k][0] + off[k][1]].clone() for k in names}
        dz = plan.debug_tensor(conv0_out, grad=True).clone()          # (fuse = 1: written now, by the second stem pass)
        if fuse:
            plan.debug_run_node_backward(0)                           # the unfused kernel on the materialised tensor (zeroes the arena first)
            torch.cuda.synchronize()
            o, nmel = off['features.conv0.weight']
            again = net._grad_arena[o:o + nmel]
            assert float(again.abs().max()) > 0
            assert torch.equal(again, grads['features.conv0.weight']), float((again - grads['features.conv0.weight']).abs().max())
        res[fuse] = (grads, dz)
        del plan, net
    for k in names:
        a, b = res[1][0][k], res[0][0][k]
        assert float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
    a, b = res[1][1], res[0][1]
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize('n,h,w', [(24, 256, 256), (3, 128, 256), (2, 256, 128), (1, 128, 128), (5, 192, 256)])
def test_stem_weight_gradient_with_both_operands_cut_once(n, h, w):
    """wgrad3_stem_planes_kernel (round 6, planner option stem_wgrad_planes): the image rows in an LDS ring of three bf16 planes de-interleaved by
    column parity, the dY chunk cut behind the fused BatchNorm / ReLU / pool backward into three planes read by ds_read_b64_tr_b16, four consumer waves
    that only multiply and four producer waves that only load / cut, chunks walked by pooling-window row pair -- against wgrad3_stem_kernel<*, true>
    (fp32 rows and chunk in LDS, every wave cuts what it reads) on the SAME plan state.  conv0's weight gradient (autograd wgrad of models/cu_net.py:300):
    (1) bit for bit: the fused planes kernel of a real backward pass == the unfused planes kernel on the materialised d(loss)/d(conv0 output) -- the dz
    arithmetic inside the producers is stem_bwd_kernel<1>'s, operation for operation;
    (2) against the staging kernel on the same tensor: the same pieces enter the same six products over the same rows per workgroup, but a workgroup's
    pixels enter its fp32 accumulators row pair by row pair instead of row by row -- agreement to the summation order: 2e-6 of the gradient's magnitude
    per element (the sum runs over N x 128 x 128 products; measured ~2e-7).
    The bench batch, rectangular batches (2 / 4 column blocks per output row; 13-row workgroups start on odd rows: single rows of a pair at either
    end of a workgroup's range), a small image."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=111)
    gen = torch.Generator().manual_seed(112)
    x = torch.rand(n, 3, h, w, generator=gen) * 2.0 - 0.7
    target = torch.rand(n, 16, h // 4, w // 4, generator=gen)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    plan = net._get_plan(n, h, w, True)
    d = plan.handle.describe()
    assert d['nodes'][0]['op'] == 'stem_conv' and d['nodes'][0]['wg3'] > 0
    o, nmel = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}['features.conv0.weight']
    try:
        plan.debug_set_option('stem_wgrad_planes', 1)
        plan.stage_target(target.cuda())
        plan.forward(x.cuda(), True, want_outputs=False)
        plan.backward(None)
        torch.cuda.synchronize()
        fused = net._grad_arena[o:o + nmel].clone()
        assert float(fused.abs().max()) > 0 and torch.isfinite(fused).all()
        conv0_out = [t['name'] for t in d['tensors'] if t['id'] == d['nodes'][0]['out']][0]
        plan.debug_tensor(conv0_out, grad=True)                  # materialises d(loss)/d(conv0 output) from the state the backward left
        got = {}
        for planes in (1, 0):
            plan.debug_set_option('stem_wgrad_planes', planes)
            plan.debug_run_node_backward(0)                          # the UNFUSED kernels on the materialised tensor (zeroes the arena first)
            torch.cuda.synchronize()
            got[planes] = net._grad_arena[o:o + nmel].clone()
    finally:
        plan.debug_set_option('stem_wgrad_planes', 0)
    if not torch.equal(fused, got[1]):
        dd = (fused - got[1]).abs()
        raise AssertionError(f'fused vs unfused planes kernel: {int((dd > 0).sum())}/{nmel} elements differ, max {float(dd.max()):.3e} of {float(got[1].abs().max()):.3e}')
    scale = float(got[0].abs().max())
    err = float((got[1] - got[0]).abs().max())
    print(f'planes vs staging kernel: max |diff| {err:.3e} of {scale:.3e}')
    assert err <= 2e-6 * scale, (err, scale)


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16_grads'])
def test_pool_backward_fused_with_the_gathers_around_it(mode):
    """Round 6, planner option fuse_pool_gather (default 1): in front of a down block's adapter pair backward runs gather(pool output) ->
    pool backward and gather(skip adapter output) as ONE launch (gather_pool_pair_kernel) instead of three.  After a real backward pass:
    the pre-pool tensor's gradient must be EXACTLY the pooled tensor's (stored) gradient routed to the first arg-max of every 2 x 2 window
    (models/cu_net.py:260, torch's max_pool2d backward; torch.equal), the pool class must have no launch left and the gather class 8 per
    U-Net fewer than with the option off, and loss / parameter gradients of the two selections agree to the order of the fp64 atomics."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    L = 2
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=L, order=1, loss_num=L)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=73)
    x, target = O.synthetic_batch(4, 16, 256, seed=74)
    res = {}
    try:
        for fuse in (1, 0):
            set_planner_option('fuse_pool_gather', fuse)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(4, 256, 256, True, bf16=mode != 'fp32')
            desc = plan.handle.describe()
            plan.handle.profile_begin(1)
            plan.handle.profile_reset()
            loss = plan.stage_target(target.cuda())
            if mode == 'fp32':
                plan.forward(x.cuda(), True, want_outputs=False)
            else:
                plan.forward_bf16(x.cuda(), 2 if mode == 'bf16_grads' else 1, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            counts = {k: v[0] for k, v in plan.handle.profile_collect().items()}
            plan.handle.profile_begin(0)
            T = desc['tensors']
            npool = 0
            for nd in desc['nodes']:
                if nd['op'] != 'pool':
                    continue
                npool += 1
                xin = plan.debug_tensor(T[nd['segs'][0]['t']]['name']).float()
                gy = plan.debug_tensor(T[nd['out']]['name'], grad=True).float()
                gx = plan.debug_tensor(T[nd['segs'][0]['t']]['name'], grad=True).float()
                leaf = xin.clone().requires_grad_(True)
                torch.nn.functional.max_pool2d(leaf, 2, 2).backward(gy)
                assert torch.equal(gx, leaf.grad), (fuse, nd['name'])
            assert npool == 4 * L
            res[fuse] = (float(loss), net._grad_arena.clone().cpu(), counts)
    finally:
        set_planner_option('fuse_pool_gather', 1)
    assert res[1][2].get('pool_bwd', 0) == 0 and res[0][2]['pool_bwd'] == 4 * L, (res[1][2], res[0][2])
    assert res[0][2]['bn_bwd_apply'] - res[1][2]['bn_bwd_apply'] == 4 * L, (res[0][2]['bn_bwd_apply'], res[1][2]['bn_bwd_apply'])
    assert abs(res[1][0] - res[0][0]) <= 1e-6 * abs(res[0][0])
    ga, gb = res[1][1], res[0][1]
    assert float((ga - gb).norm() / gb.norm()) <= (1e-5 if mode == 'fp32' else 2e-2)


@pytest.mark.parametrize('mode', ['bf16_grads'])
def test_single_consumer_gather_folded_into_the_data_gradient(mode):
    """Round 6, planner option fuse_z_gather (default 0: measured slower): the gradient of a dense layer's bottleneck output z (models/cu_net.py:43-48: read
    by norm2 -> relu -> conv2 only) is not gathered by its own launch -- conv1's data gradient assembles A * dz + E - D * z on its operand
    load from coefficient tables it derives itself, rounds it where the gather would have stored it and writes the tensor as it goes (the
    weight gradient of conv1 and this test read it).  Against the same step with the option off: 9 gather launches per U-Net (+ 1) fewer, every
    z gradient tensor and the parameter gradients equal to the rounding of the statistics atomics."""
    from cu_net_amd._lib import set_planner_option
    from oracle import cunet_ref as O
    L = 2
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=L, order=1, loss_num=L)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=75)
    x, target = O.synthetic_batch(4, 16, 256, seed=76)
    res = {}
    try:
        for fuse in (1, 0):
            set_planner_option('fuse_z_gather', fuse)
            net = cu_net_amd.create_cu_net(**cfg)
            net.load_state_dict(st)
            net = net.cuda().train()
            plan = net._get_plan(4, 256, 256, True, bf16=True)
            desc = plan.handle.describe()
            plan.handle.profile_begin(1)
            plan.handle.profile_reset()
            loss = plan.stage_target(target.cuda())
            plan.forward_bf16(x.cuda(), 2, want_outputs=False)
            plan.backward(None)
            torch.cuda.synchronize()
            counts = {k: v[0] for k, v in plan.handle.profile_collect().items()}
            plan.handle.profile_begin(0)
            zg = {}
            for nd in desc['nodes']:
                if nd['op'] == 'conv' and nd['name'].endswith('.conv1'):
                    zg[nd['name']] = plan.debug_tensor(desc['tensors'][nd['out']]['name'], grad=True).float().cpu()
            assert len(zg) == 9 * L
            res[fuse] = (float(loss), net._grad_arena.clone().cpu(), counts, zg)
    finally:
        set_planner_option('fuse_z_gather', 0)
    # (9 bottleneck outputs per U-Net + the last U-Net's output, whose only consumer is its head)
    assert res[0][2]['bn_bwd_apply'] - res[1][2]['bn_bwd_apply'] == 9 * L + 1, (res[0][2]['bn_bwd_apply'], res[1][2]['bn_bwd_apply'])
    assert abs(res[1][0] - res[0][0]) <= 1e-6 * abs(res[0][0])
    worst = 0.0
    for name, a in res[1][3].items():
        b = res[0][3][name]
        assert torch.isfinite(a).all(), name
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel <= 2e-2, (name, rel)                 # (bf16 tensors downstream of atomics-ordered fp64 sums: a few one-ulp flips)
    ga, gb = res[1][1], res[0][1]
    assert float((ga - gb).norm() / gb.norm()) <= 2e-2
    print(f'worst relative difference of a z gradient tensor, fused vs gathered: {worst:.3e}')


@pytest.mark.parametrize('n', [4, 24])
def test_split_weight_gradient_with_operands_cut_once_is_bit_identical(n):
    """wgrad5_split_kernel (round 6, planner option wgrad_split_planes, default 0 -- 8 % less time for its class alone, nothing in the step: fp32 operands cut into their three bf16 pieces ONCE, on the way into
    LDS behind BatchNorm + ReLU; MFMA fragments by ds_read_b64_tr_b16 from three bf16 planes) against wgrad3_kernel<..., EMU> (fp32 tiles in
    LDS, every wave cuts the fragments it reads) on the SAME plan state, node by node: the same pieces enter the same six products per pair in
    the same k order, so every 1x1 weight gradient of a slice of at most 8 channel tiles must agree bit for bit (288- and 320-channel slices
    run wgrad3_kernel under either setting).  N = 4: few short splits; N = 24: the bench's geometry.  Up-sampled segments, split-K (<= 5 tiles) and
    two-half ownerships incl. odd tile counts (7, 9) all occur in CU-Net-2.  Autograd wgrad of models/cu_net.py:24,43."""
    from oracle import cunet_ref as O
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=95)
    x, _ = O.synthetic_batch(n, 16, 256, seed=96)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().train()
    plan = net._get_plan(n, 256, 256, True)
    plan.forward(x.cuda(), True, want_outputs=False)
    torch.cuda.synchronize()
    desc = plan.handle.describe()
    T = desc['tensors']
    off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
    nodes = [(k, nd) for k, nd in enumerate(desc['nodes']) if nd['op'] == 'conv' and nd['taps'] == 1 and nd.get('wg3', 0) > 0]
    assert len(nodes) >= 40
    seen_ct, bad = set(), []
    try:
        for k, nd in nodes:
            o, nmel = off[nd['conv'] + '.weight']
            t = T[nd['out']]
            gen = torch.Generator().manual_seed(3000 + k)
            plan.debug_poke(t['name'], torch.randn((t['N'], t['C'], t['H'], t['W']), generator=gen), grad=True)      # this node's d(loss)/d(out)
            got = {}
            for planes in (1, 0):
                plan.debug_set_option('wgrad_split_planes', planes)      # (the plan's own snapshot: nothing process-wide changes)
                plan.debug_run_node_backward(k)
                torch.cuda.synchronize()
                got[planes] = net._grad_arena[o:o + nmel].clone()
            seen_ct.add(nmel // (128 * 32))
            if not torch.equal(got[1], got[0]):
                d = (got[1] - got[0]).abs()
                bad.append(f'{nd["name"]}: {int((d > 0).sum())}/{nmel} elements differ, max {float(d.max()):.3e} of {float(got[0].abs().max()):.3e}')
            assert float(got[0].abs().max()) > 0
    finally:
        plan.debug_set_option('wgrad_split_planes', 0)
    assert not bad, '\n'.join(bad[:20])
    assert {4, 5, 6, 8, 9, 10} <= seen_ct, seen_ct
