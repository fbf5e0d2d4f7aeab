"""GPU (-m gpu): the HIP path, called through the C ABI, against the reference's golden vectors and
against a tensor-by-tensor CPU execution of the same plan.

Tolerance (BASELINE.json north_star): fp32 heat maps within 1e-4 relative.  Forward tensors are
checked as  max|hip - ref| <= RTOL * max|ref| + ATOL  with RTOL = 1e-4 on the well-conditioned cases
(full-width CU-Net-2 at 256x256, G9 at 128x128) and 2e-3 on the 64x64 toy nets whose neck is 1x1 with
N = 2..3 (BatchNorm over 2-3 samples amplifies 1e-7 rounding by 1e3).  Whole-network GRADIENTS are
chaotic in fp32 (a flipped ReLU / arg-max decision changes an element by O(1); torch's own fp32
gradients are 1e-2 away from an fp64 evaluation on these nets, see tools/diag_backward.py), so here
they get a relative-L2 sanity bound; the exact per-kernel gradient checks on identical inputs are in
tests/test_gpu_nodes.py.  Integer outputs (num_batches_tracked) are bit-exact.  Per-tensor report in
gpurun_out/.
"""
import os

import pytest
import torch

import cu_net_amd
from cu_net_amd.trainer import FusedTrainer
from oracle import cunet_ref as O
from tests._golden import TINY, Golden
from tests._plan_interp import run_plan

pytestmark = pytest.mark.gpu
RTOL_ACT, RTOL_TOY, ATOL = 1e-4, 2e-3, 1e-6
L2_GRAD = 0.25
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(tag, lines):
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f'parity_{tag}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    except OSError:
        pass


def _cmp(name, got, ref, rtol, lines, bad, l2=None):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item()
    rel2 = ((got - ref).double().norm() / (ref.double().norm() + 1e-30)).item()
    if l2 is None:
        ok = err <= rtol * mag + ATOL and bool(torch.isfinite(got).all())
    else:
        ok = rel2 <= l2 and bool(torch.isfinite(got).all())
    lines.append(f'{"ok " if ok else "BAD"} {name:70s} err={err:.3e} mag={mag:.3e} rel={err / (mag + 1e-30):.2e} relL2={rel2:.2e}')
    if not ok:
        bad.append(name)


def _make_net(g):
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(g.group('state0'))
    return net.cuda()


@pytest.mark.parametrize('tag', TINY)
def test_train_step_tensorwise(tag):
    g = Golden(tag)
    x, target = g.t('x'), g.t('target')
    net = _make_net(g)
    net.train()
    tr = FusedTrainer(net)
    xd, td = x.cuda(), target.cuda()
    n, _, h, w = x.shape
    # CPU execution of the same plan for per-tensor references
    plan = net._get_plan(n, h, w, True)
    st = g.group('state0')
    for k in st:
        if st[k].is_floating_point() and 'running' not in k:
            st[k].requires_grad_(True)
    outs_ref, acts_ref, grads_ref, loss_ref = run_plan(plan.handle.describe(), st, x, True, True, target)

    lines, bad = [], []
    rtol = RTOL_TOY if (h < 128 or n * (h // 64) * (w // 64) < 16) else RTOL_ACT   # BN samples at the neck
    loss = tr.step(xd, td)
    torch.cuda.synchronize()
    desc = plan.handle.describe()
    for t in desc['tensors']:
        _cmp('act  ' + t['name'], plan.debug_tensor(t['name']), acts_ref[t['name']], rtol, lines, bad)
    for a, b in zip(tr.last_outputs(x.shape), g.list('out')):
        _cmp('golden out', a, b, rtol, lines, bad)
    _cmp('loss', loss, g.t('loss'), rtol, lines, bad)
    # G6 (64x64 input, 1x1 neck): BatchNorm over 4 samples makes whole-net gradients noise-dominated;
    # its backward is covered kernel-by-kernel in tests/test_gpu_nodes.py
    check_grads = not tag.startswith('G6')
    for t in reversed(desc['tensors']):
        if check_grads and t['name'] in grads_ref:
            _cmp('grad ' + t['name'], plan.debug_tensor(t['name'], grad=True), grads_ref[t['name']], 0, lines, bad, l2=L2_GRAD)
    gg = g.group('grad')
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
    for k, v in gg.items():
        if not check_grads:
            break
        o, nmel, shape = off[k]
        _cmp('dparam ' + k, net._grad_arena[o:o + nmel].view(shape), v, 0, lines, bad, l2=L2_GRAD)
    for k in g.z['grad_none'].tolist():
        o, nmel, shape = off[k]
        assert float(net._grad_arena[o:o + nmel].abs().max()) == 0.0
    sd = net.state_dict()
    for k, v in g.group('state1').items():
        if 'num_batches_tracked' in k:
            assert int(sd[k]) == int(v), k
        elif not check_grads and 'running' not in k:
            continue
        elif 'running' in k:
            _cmp('state1 ' + k, sd[k], v, 10 * rtol, lines, bad)
        else:
            # RMSprop's first step moves a weight by ~10*lr*sign(g) whatever |g| is, so compare only
            # where the reference gradient is clearly non-zero (elsewhere sign(g) is rounding noise)
            gr = gg.get(k)
            if gr is None:
                _cmp('state1 ' + k, sd[k], v, 1e-6, lines, bad)
            else:
                m = gr.abs() > 0.05 * gr.abs().max()
                if bool(m.any()):
                    _cmp('state1 ' + k, sd[k].cpu()[m], v[m], 0, lines, bad, l2=1e-2)
    _report(tag, lines)
    assert not bad, f'{len(bad)} tensors out of tolerance, first: {bad[:5]} (see gpurun_out/parity_{tag}.txt)'


@pytest.mark.parametrize('tag', ['G1_L2_o1', 'G2_L3_o2'])
def test_eval_and_nograd_forward(tag):
    g = Golden(tag)
    x = g.t('x').cuda()
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(g.group('state1'))
    net.cuda().eval()
    with torch.no_grad():
        outs = net(x)
    for a, b in zip(outs, g.list('eval')):
        err = (a.cpu() - b).abs().max().item()
        assert err <= RTOL_TOY * b.abs().max().item() + ATOL, err   # state1 itself carries toy-net noise
    # train-mode forward without backward: single running-stat update, counters +1 (bit-exact)
    net.train()
    with torch.no_grad():
        net(x)
    sd = net.state_dict()
    for k, v in g.group('state2').items():
        if 'tracked' in k:
            assert int(sd[k]) == int(v), k
        else:
            assert (sd[k].cpu() - v).abs().max().item() <= 10 * RTOL_TOY * v.abs().max().item() + ATOL, k


@pytest.mark.parametrize('tag', ['G1_L2_o1', 'G3_L4_o1_ln2'])
def test_autograd_path_matches_reference(tag):
    """The drop-in loop of cu-net.py:171-183 with torch's own loss, backward() and RMSprop."""
    g = Golden(tag)
    x, target = g.t('x').cuda(), g.t('target').cuda()
    net = _make_net(g)
    net.train()
    opt = torch.optim.RMSprop(net.parameters(), lr=2.5e-4, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0)
    out = net(x)
    loss = 0
    for o in out:
        t = (o - target) ** 2
        loss = loss + t.sum() / t.numel()
    opt.zero_grad()
    loss.backward()
    gg = g.group('grad')
    none = set(g.z['grad_none'].tolist())
    for k, p in net.named_parameters():
        if k in none:
            assert p.grad is None, k
        else:
            rel2 = ((p.grad.cpu() - gg[k]).double().norm() / (gg[k].double().norm() + 1e-30)).item()
            assert rel2 <= L2_GRAD, (k, rel2)
    opt.step()
    assert abs(float(loss) - float(g.t('loss'))) <= RTOL_TOY * abs(float(g.t('loss')))
    sd = net.state_dict()
    for k, v in g.group('state1').items():
        if 'tracked' in k:
            assert int(sd[k]) == int(v), k


def test_full_width_matches_reference():
    """Full-width CU-Net-2 (K=68), N=1, 256x256 on the oracle's deterministic init."""
    g = Golden('G5_full_L2K68')
    spec = O.Spec(**g.cfg)
    st = O.init_state(spec, seed=int(g.z['init_seed']))
    x, target = O.synthetic_batch(1, spec.class_num, 256, seed=int(g.z['batch_seed']))
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(st)
    net.cuda().train()
    tr = FusedTrainer(net)
    loss = tr.step(x.cuda(), target.cuda())
    assert abs(float(loss) - float(g.z['loss'])) <= 1e-4 * float(g.z['loss'])
    for i, o in enumerate(tr.last_outputs(x.shape)):
        ref = g.t(f'out_sub/{i}')
        err = (o.cpu()[:, ::4, ::4, ::4] - ref).abs().max().item()
        assert err <= RTOL_ACT * ref.abs().max().item() + ATOL, (i, err)
    names = g.z['grad_norm_names'].tolist()
    off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
    for k, nrm in zip(names, g.z['grad_norms']):
        o, nmel = off[k]
        got = float(net._grad_arena[o:o + nmel].double().norm())
        assert abs(got - nrm) <= 5e-2 * nrm + 1e-9, (k, got, nrm)   # whole-net fp32 gradients: sanity bound only


@pytest.fixture
def ring_3x3_everywhere():
    """The LDS row-ring 3x3 forward is planned for batches with >= 512 image rows; force it onto small batches."""
    from cu_net_amd._lib import set_planner_option
    set_planner_option('conv3x3_ring_min_rows', 2)
    try:
        yield
    finally:
        set_planner_option('conv3x3_ring_min_rows', 512)


def test_full_width_matches_reference_on_the_3x3_row_ring(ring_3x3_everywhere):
    """The same reference vectors with the 64x64 3x3 convolutions on conv3x3_ring_kernel (train-mode forward: batch-statistic
    BatchNorm on the way into the ring, output statistics, heat maps and loss against the reference)."""
    test_full_width_matches_reference()


@pytest.mark.parametrize('qin', [0, 8])
def test_3x3_row_ring_agrees_with_the_weight_stationary_kernel(ring_3x3_everywhere, qin):
    """(qin = 8: with the QuanInput2d quantiser in front of the 3x3 convolutions, applied on the way into the ring.)
    Eval- and train-mode forward of a full-width net, N = 3 (ragged last workgroup: 192 image rows in ranges of 2), on the ring
    kernel against the weight-stationary kernel on the same state: the two differ in summation order only."""
    from cu_net_amd._lib import set_planner_option
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=51)
    x, _ = O.synthetic_batch(3, 16, 256, seed=52)
    outs = {}
    for mode in ('ring', 'plain'):
        set_planner_option('conv3x3_ring_min_rows', 2 if mode == 'ring' else 1 << 30)
        net = cu_net_amd.create_cu_net(**cfg)
        net.load_state_dict(st)
        net.cuda().eval()
        if qin:
            net.set_quant_input(qin, ())
        with torch.no_grad():
            ev = [o.cpu() for o in net(x.cuda())]
        net.train()
        plan = net._get_plan(3, 256, 256, True)
        plan.forward(x.cuda(), True, want_outputs=False)
        torch.cuda.synchronize()
        desc = plan.handle.describe()
        first3 = [n for n in desc['nodes'] if n['op'] == 'conv' and n['taps'] == 9 and desc['tensors'][n['out']]['W'] == 64][0]
        tr = plan.debug_tensor(desc['tensors'][first3['out']]['name']).cpu()
        outs[mode] = (ev, tr, {k: v.clone().cpu() for k, v in net.state_dict().items() if 'running' in k})
    for a, b in zip(outs['ring'][0], outs['plain'][0]):
        # (quantised activations: a BatchNorm output within rounding of a quantiser step may land on either side downstream)
        assert (a - b).abs().max().item() <= (2e-3 if qin else 2e-5) * b.abs().max().item() + 1e-7
    a, b = outs['ring'][1], outs['plain'][1]
    assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()          # one node, identical inputs
    for k, v in outs['plain'][2].items():
        if qin:      # flipped quantiser steps propagate through the later BatchNorms: aggregate bound only
            assert float((outs['ring'][2][k] - v).norm() / (v.norm() + 1e-12)) <= 2e-2, k
        else:
            assert torch.allclose(outs['ring'][2][k], v, rtol=1e-4, atol=1e-6), k


def test_full_width_eval_forward_matches_oracle():
    """Inference path at full width (running statistics, every full-width kernel variant incl. the tap-split 3x3):
    HIP eval forward vs the oracle's eval forward on the same state, after one training step moved the
    running statistics away from their initial values.  N=2 so that the 8x8 and 4x4 levels have several tiles."""
    g = Golden('G5_full_L2K68')
    spec = O.Spec(**g.cfg)
    st = O.init_state(spec, seed=int(g.z['init_seed']))
    x, target = O.synthetic_batch(2, spec.class_num, 256, seed=7)
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(st)
    net.cuda().train()
    FusedTrainer(net).step(x.cuda(), target.cuda())
    st1 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    net.eval()
    with torch.no_grad():
        outs = net(x.cuda())
    refs = O.forward(spec, st1, x, training=False)
    assert len(outs) == len(refs)
    for i, (o, r) in enumerate(zip(outs, refs)):
        err = (o.cpu() - r).abs().max().item()
        assert err <= RTOL_ACT * r.abs().max().item() + ATOL, (i, err, r.abs().max().item())


@pytest.mark.parametrize('layer_num,class_num', [(8, 68), (16, 16)])
def test_deep_eval_forward_matches_oracle(layer_num, class_num):
    """north_star's 1e-4 on EVERY head of the deep configurations (BASELINE configs 3 / 4 / 5's networks: CU-Net-8 K = 68, CU-Net-16
    K = 16, full width), where it is well defined: the eval-mode forward (models/cu_net.py:336-360 with running statistics -- no
    batch-statistics feedback, so rounding is not amplified U-Net after U-Net as it is in train mode).  The running
    statistics are first driven to this batch's own statistics by 40 train-mode forwards (1 - 0.9^40 = 0.985), so that every
    BatchNorm output keeps unit scale through all 8 / 16 U-Nets; then HIP eval forward vs the oracle's eval forward on that state."""
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=class_num, layer_num=layer_num, order=1, loss_num=layer_num)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=90 + layer_num)
    x, _ = O.synthetic_batch(2, class_num, 256, seed=92)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net.cuda().train()
    xd = x.cuda()
    with torch.no_grad():
        for _ in range(40):
            net(xd)
    st1 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    assert int(st1['features.norm0.num_batches_tracked']) == 40
    net.eval()
    with torch.no_grad():
        outs = net(xd)
    refs = O.forward(spec, st1, x, training=False)
    assert len(outs) == len(refs) == layer_num
    lines, bad = [], []
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert r.abs().max().item() > 1e-3, (i, 'degenerate heat map')
        _cmp(f'head {i}', o, r, RTOL_ACT, lines, bad)
    _report(f'deep_eval_L{layer_num}K{class_num}', lines)
    assert not bad, '\n'.join(lines)


def test_get_preds_bit_exact():
    torch.manual_seed(3)
    s = torch.randn(3, 5, 64, 64)
    s[0, 0] = -1.0                      # all <= 0 -> zeros
    s[0, 1] = 0.0
    s[1, 2, 10, 20] = 9.0; s[1, 2, 40, 7] = 9.0     # tie -> lowest flat index
    s[2, 4, 63, 63] = 50.0              # border maximum
    s[2, 3, 0, 0] = 50.0
    maxval, idx = torch.max(s.view(3, 5, -1), 2)     # pylib/Evaluation.py:11-21 restated
    px = (idx % 64 + 1).float()
    py = torch.floor(idx.float() / 64) + 1
    ref = torch.stack([px, py], 2) * maxval.gt(0).unsqueeze(2).float()
    got = cu_net_amd.get_preds(s.cuda()).cpu()
    assert torch.equal(got, ref)


def test_decode_matches_reference_vectors():
    """get_preds / final_preds against vectors produced by the reference's pylib/Evaluation.py (G8): bit-exact."""
    import numpy as np
    from tests._golden import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'G8_decode.npz'))
    hm = torch.from_numpy(z['heat']).cuda()
    assert torch.equal(cu_net_amd.get_preds(hm).cpu(), torch.from_numpy(z['get_preds']))
    fp = cu_net_amd.final_preds(hm, torch.from_numpy(z['center']), torch.from_numpy(z['scale']), [64, 64],
                                torch.zeros(hm.shape[0]))
    assert torch.equal(fp.cpu(), torch.from_numpy(z['final_preds']))


def test_decode_with_rotation_matches_reference_vectors():
    """final_preds with rot != 0 (the rotation branch of GetTransform, pylib/Evaluation.py:163-178; one image of the batch keeps
    rot == 0) against the EXECUTED reference (G8r): bit-exact integer coordinates."""
    import numpy as np
    from tests._golden import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'G8r_decode_rot.npz'))      # (angles + result; maps, centres and scales are G8's)
    z0 = np.load(os.path.join(GOLDEN_DIR, 'G8_decode.npz'))
    hm = torch.from_numpy(z0['heat']).cuda()
    fp = cu_net_amd.final_preds(hm, torch.from_numpy(z0['center']), torch.from_numpy(z0['scale']), [64, 64], torch.from_numpy(z['rot']))
    assert torch.equal(fp.cpu(), torch.from_numpy(z['final_preds']))
    # and the affine entry point reproduces the rot == 0 vectors as well
    from cu_net_amd.trainer import _inverse_crop_transforms, _ptr, _stream_ptr
    from cu_net_amd._lib import check, lib
    inv = torch.from_numpy(_inverse_crop_transforms(torch.from_numpy(z0['center']), torch.from_numpy(z0['scale']), torch.zeros(hm.shape[0]), 64)).cuda()
    preds = torch.empty((hm.shape[0], hm.shape[1], 2), dtype=torch.float32, device='cuda')
    check(lib().cunet_final_preds_affine(_ptr(hm), _ptr(inv), _ptr(preds), hm.shape[0], hm.shape[1], 64, 64, 64, 64, _stream_ptr(hm.device)), 'affine')
    assert torch.equal(preds.cpu(), torch.from_numpy(z0['final_preds']))


def test_flip_merge_and_accuracy_bit_exact():
    """Validation-loop pieces on the GPU vs the reference's vectors (G10) and the oracle: flip-TTA merge is
    bit-exact ((a + b) / 2 in fp32), PCK accuracy is exact (integer coordinates, IEEE sqrt / divide)."""
    import numpy as np
    from oracle import decode_ref as DR
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G10_tta_accuracy.npz'))
    o1, o2, tgt = (torch.from_numpy(z[k]) for k in ('out1', 'out2', 'target'))
    m = cu_net_amd.flip_merge(o1.cuda(), o2.cuda(), z['flip_index'])
    assert torch.equal(m.cpu(), DR.flip_merge(o1, o2, z['flip_index']))
    assert torch.equal(m.cpu()[:, :, ::4, ::4], torch.from_numpy(z['merged_sub']))
    acc = cu_net_amd.accuracy(m, tgt.cuda(), z['idxs'].tolist())
    assert torch.equal(acc, torch.from_numpy(z['accuracy']))
    # original-resolution PCKh against the oracle's composition of final_preds + calc_dists + dist_acc
    z8 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G8_decode.npz'))
    hm, center, scale = (torch.from_numpy(z8[k]) for k in ('heat', 'center', 'scale'))
    gen = torch.Generator().manual_seed(3)
    fp = DR.final_preds(hm, center, scale, [64, 64], torch.zeros(hm.shape[0]))
    gt = fp + torch.randint(-6, 7, fp.shape, generator=gen).float()
    gt[0, 2] = 0.0                                          # missing joint
    normalizers = torch.rand(hm.shape[0], generator=gen) * 30 + 20
    idxs = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15]
    d = DR.calc_dists(fp, gt, normalizers, use_zero=True)
    ref = torch.zeros(len(idxs) + 1)
    avg, cnt = 0, 0
    for i, j in enumerate(idxs):
        ref[i + 1] = DR.dist_acc(d[j])
        if ref[i + 1] >= 0:
            avg = avg + ref[i + 1]; cnt += 1
    ref[0] = avg / cnt
    got = cu_net_amd.accuracy_origin_res(hm.cuda(), center, scale, [64, 64], gt, normalizers)
    assert torch.equal(got, ref), (got, ref)


def test_driver_trains_validates_and_resumes(tmp_path):
    """train.py's loop on the GPU: two short synthetic epochs (loss must fall), flip-TTA validation, checkpoint in the
    reference's layout, resume from it at the next epoch."""
    from cu_net_amd import driver as D
    args = ['--exp_id', 'gpu', '--exp_dir', str(tmp_path), '--layer_num', '2', '--order', '1', '--class_num', '16',
            '--loss_num', '2', '--bs', '4', '--synthetic', '6', '--print_freq', '100', '--lr', '1e-3']
    h = D.main(args + ['--nEpochs', '2'])
    assert [e['epoch'] for e in h.epoch] == [0, 1]
    assert h.loss[1]['train_loss'] < h.loss[0]['train_loss']
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'gpu')))
    assert 'lr-0.001-1.pth.tar' in files and 'opt.txt' in files
    h2 = D.main(args + ['--nEpochs', '3', '--resume_prefix', 'lr-0.001-1.pth.tar'])
    assert [e['epoch'] for e in h2.epoch] == [0, 1, 2]
    assert h2.loss[2]['train_loss'] < h.loss[0]['train_loss']


def test_driver_with_device_side_sample_preparation(tmp_path):
    """--augment: raw variable-size samples go through cu_net_amd.prepare_batch (jitter, flip, colour, crop, targets on the GPU)
    inside the driver's loop."""
    from cu_net_amd import driver as D
    args = ['--exp_id', 'aug', '--exp_dir', str(tmp_path), '--layer_num', '2', '--order', '1', '--class_num', '16',
            '--loss_num', '2', '--bs', '4', '--synthetic', '3', '--augment', '--print_freq', '100', '--lr', '1e-3', '--nEpochs', '1']
    h = D.main(args)
    assert [e['epoch'] for e in h.epoch] == [0] and h.loss[0]['train_loss'] > 0 and h.loss[0]['train_loss'] == h.loss[0]['train_loss']


def test_target_synthesis_bit_exact():
    """Gaussian target maps rendered on the GPU vs the oracle (== the reference, G11): bit-exact, sigma 1 and 2."""
    import numpy as np
    from oracle import decode_ref as DR
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G11_targets.npz'))
    pts = torch.from_numpy(z['pts'])
    for sigma in (1, 2):
        got = cu_net_amd.pts2heatmap(pts.cuda(), (64, 64), sigma).cpu().numpy()
        ref = DR.pts2heatmap(z['pts'].copy(), (64, 64), sigma)[0].astype(np.float32)
        assert np.array_equal(got, ref), sigma
        assert np.array_equal(got[:, ::2, ::2], z[f'heat_s{sigma}'])
    batch = cu_net_amd.pts2heatmap(pts.view(2, 20, 2).cuda(), (64, 64))
    ref1 = DR.pts2heatmap(z['pts'].copy(), (64, 64), 1)[0].astype(np.float32)
    assert batch.shape == (2, 20, 64, 64) and torch.equal(batch.view(40, 64, 64).cpu(), torch.from_numpy(ref1))


@pytest.mark.parametrize('n,h,w', [(2, 128, 192), (3, 192, 128), (1, 256, 128)])
def test_non_square_and_odd_batches_match_oracle(n, h, w):
    """Rectangular inputs and odd batch sizes (ragged 32-row tiles at the coarse levels): forward outputs, loss,
    running statistics and parameter gradients of one train step against the oracle evaluated on the spot."""
    cfg = dict(neck_size=2, growth_rate=8, init_chan_num=16, class_num=5, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=31)
    gen = torch.Generator().manual_seed(32)
    x = torch.rand(n, 3, h, w, generator=gen)
    target = torch.rand(n, 5, h // 4, w // 4, generator=gen) * 0.2
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net.cuda().train()
    tr = FusedTrainer(net)
    loss = tr.step(x.cuda(), target.cuda())
    outs = tr.last_outputs(x.shape)
    st_ref = {k: v.clone() for k, v in st.items()}
    ref_loss, ref_outs, ref_grads = O.train_step(spec, st_ref, x, target)
    assert abs(float(loss) - float(ref_loss)) <= RTOL_TOY * abs(float(ref_loss))
    for a, b in zip(outs, ref_outs):
        assert a.shape == b.shape
        assert (a.cpu() - b).abs().max().item() <= RTOL_TOY * b.abs().max().item() + ATOL
    sd = net.state_dict()
    for k, v in st_ref.items():
        if 'running' in k:
            assert (sd[k].cpu() - v).abs().max().item() <= 10 * RTOL_TOY * v.abs().max().item() + ATOL, k
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
    num = den = 0.0
    for k, gref in ref_grads.items():
        if gref is None:
            continue
        o, nmel, shape = off[k]
        got = net._grad_arena[o:o + nmel].view(shape).cpu().double()
        num += float((got - gref.double()).pow(2).sum()); den += float(gref.double().pow(2).sum())
    assert (num / den) ** 0.5 <= L2_GRAD, (num / den) ** 0.5


def test_bf16_inference_forward_close_to_fp32():
    """bf16-storage inference (cunet_forward_bf16) against the fp32 HIP eval forward and the fp32 oracle at production
    widths.  Tolerance for bf16 storage (8 mantissa bits, re-rounded after every BatchNorm+ReLU and every conv):
    max |diff| <= 3e-2 of the heat-map range; the arg-max landmarks of clear peaks must not move by more than a pixel."""
    g = Golden('G5_full_L2K68')
    spec = O.Spec(**g.cfg)
    st = O.init_state(spec, seed=int(g.z['init_seed']))
    x, target = O.synthetic_batch(2, spec.class_num, 256, seed=11)
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(st)
    net.cuda().train()
    tr = FusedTrainer(net, lr=1e-3)
    for _ in range(3):                                   # move weights and running statistics off their initial values
        tr.step(x.cuda(), target.cuda())
    net.eval()
    with torch.no_grad():
        ref = net(x.cuda())
        got = net.forward_bf16(x.cuda())
    assert len(got) == len(ref)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape and a.dtype == torch.float32
        rng = (b.max() - b.min()).item()
        err = (a - b).abs().max().item()
        assert err <= 3e-2 * rng, (i, err, rng)
        rel2 = ((a - b).double().norm() / b.double().norm()).item()
        assert rel2 <= 2e-2, (i, rel2)
    st1 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    oref = O.forward(spec, st1, x, training=False)
    assert (got[-1].cpu() - oref[-1]).abs().max().item() <= 3e-2 * (oref[-1].max() - oref[-1].min()).item()
    try:
        net.train()
        net.forward_bf16(x.cuda())
        assert False, 'forward_bf16 must refuse training mode'
    except cu_net_amd.CUNetError:
        pass


def test_bf16_activation_train_step_tracks_fp32():
    """FusedTrainer(bf16=True): bf16 activation storage + bf16 MFMA forward, fp32 gradients / weights / RMSprop.  The
    per-kernel exactness is in test_gpu_nodes.py; here the whole step: first loss within 2e-2 of the fp32 step's, and
    ten steps bring the loss down by the same amount to within 10 %."""
    g = Golden('G5_full_L2K68')
    spec = O.Spec(**g.cfg)
    st = O.init_state(spec, seed=int(g.z['init_seed']))
    x, target = O.synthetic_batch(4, spec.class_num, 256, seed=13)
    xd, td = x.cuda(), target.cuda()
    hist = {}
    for mode in ('fp32', 'bf16', 'bf16_grads'):
        net = cu_net_amd.create_cu_net(**g.cfg)
        net.load_state_dict(st)
        net.cuda().train()
        tr = FusedTrainer(net, lr=2.5e-4, bf16=mode != 'fp32', bf16_grads=mode == 'bf16_grads')
        hist[mode] = [float(tr.step(xd, td)) for _ in range(10)]
        assert all(torch.isfinite(torch.tensor(hist[mode])))
    a = hist['fp32']
    for mode in ('bf16', 'bf16_grads'):          # (bf16_grads: dY, dz, dX of every node stored as bf16 as well)
        b = hist[mode]
        assert abs(b[0] - a[0]) <= 2e-2 * a[0], (mode, a[0], b[0])
        assert b[-1] < b[0]
        assert abs((b[0] - b[-1]) - (a[0] - a[-1])) <= 0.1 * (a[0] - a[-1]), (mode, a, b)


def test_bf16_3x3_row_ring_agrees_with_the_other_bf16_kernels(ring_3x3_everywhere):
    """bf16 storage: the 3x3 forward of the 64- and 32-pixel-wide levels on conv3x3_ring_bf16_kernel (rows through an LDS ring, the
    whole weight operand in LDS, a wave per tile) against the tap-split / weight-stationary bf16 kernels on the same state, N = 3
    (192 image rows: ragged last workgroup).  Same bf16 operands, fp32 accumulation in another order: an output within summation
    noise of a bf16 rounding boundary may round the other way -- at most one bf16 step on a small fraction of the elements; the
    batch statistics that ride on the epilogue agree to fp32 accuracy."""
    from cu_net_amd._lib import set_planner_option
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=53)
    x, _ = O.synthetic_batch(4, 16, 256, seed=54)        # (the bf16 kernels need 32-row tiles at the neck: N * 16 rows)
    outs = {}
    for mode in ('ring', 'plain'):
        set_planner_option('conv3x3_ring_min_rows', 2 if mode == 'ring' else 1 << 30)
        net = cu_net_amd.create_cu_net(**cfg)
        net.load_state_dict(st)
        net = net.cuda().train()
        plan = net._get_plan(4, 256, 256, True, bf16=True)
        plan.forward_bf16(x.cuda(), 1, want_outputs=False)
        torch.cuda.synchronize()
        desc = plan.handle.describe()
        got = {}
        for wdt in (64, 32):
            nd = [n for n in desc['nodes'] if n['op'] == 'conv' and n['taps'] == 9 and desc['tensors'][n['out']]['W'] == wdt][0]
            got[wdt] = plan.debug_tensor(desc['tensors'][nd['out']]['name']).cpu()
        outs[mode] = (got, {k: v.clone().cpu() for k, v in net.state_dict().items() if 'running' in k})
    for wdt in (64, 32):
        a, b = outs['ring'][0][wdt], outs['plain'][0][wdt]
        mag = float(b.abs().max())
        d = (a - b).abs()
        assert float(d.max()) <= 2 ** -7 * mag, (wdt, float(d.max()), mag)
        if wdt == 64:       # the first 3x3 node: identical inputs on both sides
            assert float((d > 0).float().mean()) <= 2e-3, (wdt, float((d > 0).float().mean()))
    for k, v in outs['plain'][1].items():
        assert float((outs['ring'][1][k] - v).norm() / (v.norm() + 1e-12)) <= 2e-2, k


def test_bf16_inference_of_a_single_image():
    """`net.forward_bf16` on ONE 256 x 256 image: the bf16 kernels need whole 32-row tiles at the 4 x 4 neck (N * 16 rows), so the
    binding pads the batch with a zero image and drops its heat maps -- eval-mode images do not interact, the result equals the
    first image's heat maps in a batch of two."""
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=55)
    x, _ = O.synthetic_batch(2, 16, 256, seed=56)
    net = cu_net_amd.create_cu_net(**cfg)
    net.load_state_dict(st)
    net = net.cuda().eval()
    with torch.no_grad():
        one = net.forward_bf16(x[:1].cuda())
        two = net.forward_bf16(x.cuda())
    assert len(one) == 2 and tuple(one[0].shape) == (1, 16, 64, 64)
    for a, b in zip(one, two):
        assert torch.equal(a[0], b[0])
