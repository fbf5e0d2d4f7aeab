#!/usr/bin/env python3
"""Generate golden vectors for the CU-Net hot path by RUNNING THE REFERENCE (this container only).

The reference (`/root/reference/models/cu_net.py`) is Python 2; it is imported here through an
in-memory source shim (two textual patches, nothing is written to /root/reference and no
reference text is stored in this repo):
    print 'x'                ->  print('x')          (models/cu_net.py:286)
    adapter_out_num / 2      ->  adapter_out_num // 2 (models/cu_net.py:94, py3 true division)
The fixtures written under tests/golden/ are DATA ONLY (inputs, parameters, expected outputs).

While generating, the script also checks `oracle/cunet_ref.py` against the reference
(outputs, loss, every gradient, running statistics after one train step, RMSprop update) and
refuses to write fixtures if the oracle disagrees.

Usage:  python -B tools/gen_golden.py      (needs /root/reference; not runnable on the GPU box)
"""
import contextlib
from collections import OrderedDict
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('CUNET_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import cunet_ref as O  # noqa: E402


def load_reference_models():
    src = open(os.path.join(REF, 'models', 'cu_net.py')).read()
    src = src.replace("print 'order is larger than the layer number.'",
                      "print('order is larger than the layer number.')")
    src = src.replace('adapter_out_num = adapter_out_num / 2', 'adapter_out_num = adapter_out_num // 2')
    mod = types.ModuleType('ref_cu_net')
    exec(compile(src, '<reference models/cu_net.py via shim>', 'exec'), mod.__dict__)
    return mod


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def to_np(t):
    return t.detach().cpu().numpy()


def check(name, a, b, exact=True):
    a, b = to_np(a), to_np(b)
    if exact:
        ok = np.array_equal(a, b)
    else:
        ok = np.allclose(a, b, rtol=1e-6, atol=1e-7)
    if not ok:
        raise SystemExit(f'ORACLE MISMATCH at {name}: max abs diff {np.abs(a - b).max()}')


def one_config(ref, tag, cfg, n, hw, seed):
    torch.manual_seed(seed)
    net = quiet(ref.create_cu_net, **cfg)
    spec = O.Spec(**cfg)
    # --- state layout must match key-for-key, shape-for-shape, in order
    sd = net.state_dict()
    ents = O.state_entries(spec)
    assert [k for k in sd.keys()] == [e[0] for e in ents], 'state_dict key order differs'
    for (k, v), (_, shp, _) in zip(sd.items(), ents):
        assert tuple(v.shape) == tuple(shp), (k, v.shape, shp)
    # make BN betas / running stats non-trivial so the fixture pins them
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith('.bias'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif k.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.rand(n, 3, hw, hw, generator=g)
    target = torch.rand(n, cfg['class_num'], hw // 4, hw // 4, generator=g)

    fx = {'cfg': np.array([cfg[k] for k in ('neck_size', 'growth_rate', 'init_chan_num', 'class_num',
                                              'layer_num', 'order', 'loss_num')], dtype=np.int64),
          'x': to_np(x), 'target': to_np(target)}
    for k, v in state0.items():
        fx['state0/' + k] = to_np(v)

    # --- reference: one train step (cu-net.py:171-183) with RMSprop (cu-net.py:60-61)
    net.train()
    opt = torch.optim.RMSprop(net.parameters(), lr=2.5e-4, alpha=0.99, eps=1e-8, momentum=0, weight_decay=0)
    out = net(x)
    loss = 0
    for o in out:
        t = (o - target) ** 2
        loss = loss + t.sum() / t.numel()
    opt.zero_grad()
    loss.backward()
    ref_grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    opt.step()
    state1 = {k: v.clone() for k, v in net.state_dict().items()}

    # --- oracle on the same inputs
    ost = {k: v.clone() for k, v in state0.items()}
    oloss, oouts, ograds = O.train_step(spec, ost, x, target)
    check(tag + '/loss', loss, oloss)
    for i, (a, b) in enumerate(zip(out, oouts)):
        check(f'{tag}/out{i}', a, b)
    for k, gr in ref_grads.items():
        if gr is None:
            assert ograds[k] is None, k
        else:
            check(f'{tag}/grad/{k}', gr, ograds[k])
    for k, v in state1.items():
        check(f'{tag}/state1/{k}', v, ost[k])

    fx['loss'] = to_np(loss)
    for i, o in enumerate(out):
        fx[f'out/{i}'] = to_np(o)
    for k, gr in ref_grads.items():
        if gr is not None:
            fx['grad/' + k] = to_np(gr)
    fx['grad_none'] = np.array([k for k, gr in ref_grads.items() if gr is None], dtype='U')
    for k, v in state1.items():
        fx['state1/' + k] = to_np(v)

    # --- eval-mode forward on the post-step state (pins running-stat usage; G8)
    net.eval()
    with torch.no_grad():
        eout = net(x)
        oe = O.forward(spec, ost, x, training=False)
    for i, (a, b) in enumerate(zip(eout, oe)):
        check(f'{tag}/eval{i}', a, b)
        fx[f'eval/{i}'] = to_np(a)

    # --- train-mode forward WITHOUT backward: single running-stat update everywhere
    net.train()
    with torch.no_grad():
        net(x)
        O.forward(spec, ost, x, training=True)
    for k, v in net.state_dict().items():
        if 'running' in k or 'tracked' in k:
            check(f'{tag}/state2/{k}', v, ost[k])
            fx['state2/' + k] = to_np(v)
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **fx)
    nparam = sum(p.numel() for p in net.parameters())
    print(f'{tag}: {cfg} N={n} hw={hw} params={nparam} loss={float(loss):.6f}  oracle == reference')


def full_width(ref):
    """G5: full-width CU-Net-2 (K=68), N=1, 256x256: reference outputs on the oracle's own
    deterministic init (seed) and synthetic batch; stored sub-sampled + checksums."""
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=2, order=1, loss_num=2)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=2)
    x, target = O.synthetic_batch(1, 68, 256, seed=0)
    net = quiet(ref.create_cu_net, **cfg)
    net.load_state_dict(st)
    net.train()
    out = net(x)
    loss = sum(((o - target) ** 2).mean() for o in out)
    loss.backward()
    ost = {k: v.clone() for k, v in st.items()}
    oloss, oouts, ograds = O.train_step(spec, ost, x, target, apply_update=False)
    check('G5/loss', loss, oloss, exact=False)
    fx = {'cfg': np.array(list(cfg.values()), dtype=np.int64), 'init_seed': np.array(2), 'batch_seed': np.array(0),
          'loss': to_np(loss)}
    for i, (a, b) in enumerate(zip(out, oouts)):
        check(f'G5/out{i}', a, b)
        fx[f'out_sub/{i}'] = to_np(a)[:, ::4, ::4, ::4].copy()
        fx[f'out_sum/{i}'] = np.array([to_np(a).astype(np.float64).sum(), np.abs(to_np(a).astype(np.float64)).sum()])
    gsum = {}
    for k, p in net.named_parameters():
        if p.grad is not None:
            check('G5/grad/' + k, p.grad, ograds[k])
            gsum[k] = float(p.grad.double().norm())
    fx['grad_norm_names'] = np.array(list(gsum.keys()), dtype='U')
    fx['grad_norms'] = np.array(list(gsum.values()))
    np.savez_compressed(os.path.join(OUT, 'G5_full_L2K68.npz'), **fx)
    print(f'G5: full width L2/K68 N=1 loss={float(loss):.6f}  oracle == reference')



def full_width_cfg(ref, tag, cfg, n, init_seed, batch_seed, rq=None, bits_w=None):
    """G12 / G13: the BASELINE configs 3, 4 and 5 networks at FULL width (L = 8 / 16), one train step of the reference
    on the oracle's deterministic init and synthetic batch -- stored sub-sampled (heat maps ::4 in every dimension) with
    the loss, per-parameter gradient norms and running-statistic checksums.  With `rq` (the executed reference
    utils/quantize.py) the step is the quantised loop of cu-net-prev-version-wig.py:163-190 at bits_w / bits_g = 8:
    QuanOp.quantization -> forward -> backward -> restore -> updateQuanGradWeight."""
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=init_seed)
    x, target = O.synthetic_batch(n, cfg['class_num'], 256, seed=batch_seed)
    net = quiet(ref.create_cu_net, **cfg)
    net.load_state_dict(st)
    net.train()
    qop = None
    if rq is not None:
        rq.bitsW, rq.bitsI, rq.bitsG = bits_w, 8, 8
        qop = rq.QuanOp(net)
        qop.quantization()
    out = net(x)
    loss = 0
    for o in out:
        t = (o - target) ** 2
        loss = loss + t.sum() / t.numel()
    loss.backward()
    if qop is not None:
        qop.restore()
        qop.updateQuanGradWeight()
    ost = {k: v.clone() for k, v in st.items()}
    oloss, oouts, ograds = O.train_step(spec, ost, x, target, apply_update=False,
                                        quant=(bits_w, 8) if rq is not None else None)
    check(tag + '/loss', loss, oloss, exact=False)
    fx = {'cfg': np.array(list(cfg.values()), dtype=np.int64), 'init_seed': np.array(init_seed),
          'batch_seed': np.array(batch_seed), 'n': np.array(n), 'loss': to_np(loss),
          'bits_w': np.array(bits_w if bits_w else 0)}
    for i, (a, b) in enumerate(zip(out, oouts)):
        check(f'{tag}/out{i}', a, b)
        fx[f'out_sub/{i}'] = to_np(a)[:, ::4, ::4, ::4].copy()
        fx[f'out_sum/{i}'] = np.array([to_np(a).astype(np.float64).sum(), np.abs(to_np(a).astype(np.float64)).sum()])
    gsum = {}
    for k, p in net.named_parameters():
        if p.grad is not None:
            check(f'{tag}/grad/' + k, p.grad, ograds[k])
            gsum[k] = float(p.grad.double().norm())
    fx['grad_norm_names'] = np.array(list(gsum.keys()), dtype='U')
    fx['grad_norms'] = np.array(list(gsum.values()))
    rs = {}
    for k, v in net.state_dict().items():
        if 'running' in k or 'tracked' in k:
            check(f'{tag}/state1/{k}', v, ost[k])
            if 'running' in k:
                rs[k] = float(v.double().sum())
    fx['running_names'] = np.array(list(rs.keys()), dtype='U')
    fx['running_sums'] = np.array(list(rs.values()))
    # --- the same step of the REFERENCE in float64: the yardstick for every fp32 implementation.  At random init these
    # deep nets amplify a 1e-7 perturbation by ~3.5x per U-Net (train-mode BatchNorm), so torch's own fp32 result is
    # 4e-2 away from the fp64 one at the 8th head: tolerances in the GPU tests are multiples of THAT distance.
    net64 = quiet(ref.create_cu_net, **cfg)
    net64.load_state_dict(st)
    net64.double().train()
    if rq is not None:
        qop64 = rq.QuanOp(net64)
        qop64.quantization()
    out64 = net64(x.double())
    loss64 = 0
    for o in out64:
        t = (o - target.double()) ** 2
        loss64 = loss64 + t.sum() / t.numel()
    loss64.backward()
    if rq is not None:
        qop64.restore()
        qop64.updateQuanGradWeight()
    fx['loss64'] = to_np(loss64)
    for i, o in enumerate(out64):
        fx[f'out64_sub/{i}'] = to_np(o)[:, ::4, ::4, ::4].copy()
    g64 = dict(net64.named_parameters())
    fx['grad_norms64'] = np.array([float(g64[k].grad.norm()) for k in gsum.keys()])
    sd64 = net64.state_dict()
    fx['running_sums64'] = np.array([float(sd64[k].sum()) for k in rs.keys()])
    e = [float((a.double() - b).norm() / b.norm()) for a, b in zip(out, out64)]
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **fx)
    print(f'{tag}: full width {cfg} N={n} bits_w={bits_w} loss={float(loss):.6f}  oracle == reference; '
          f'fp32-vs-fp64 relL2 per head {["%.1e" % v for v in e]}')


def full_width_quan_input(tag, cfg, n, init_seed, batch_seed, bits_w, bits_i):
    """G13 ... _qin8: BASELINE config 5's network in the reference's QUANTISED-INPUT form -- QuanOp(bits_w) weights AND a
    QuanInput2d of bits_i bits in front of every 3x3 / head conv (models/cu_net_prev_version_wig.py:96-98,277-279).  That model
    file does not import on a modern torch (raw cuDNN bindings of 0.1.12), so unlike G12 / G13 this fixture is NOT a run of a
    reference module: it is the ORACLE's step, whose pieces are each pinned to executed reference code -- the network and the
    train step (G1..G13), QuanOp's three phases (G7, G13_bw1) and QuanInput.forward/backward (G14).  Stored like G12 / G13:
    sub-sampled heat maps, loss, gradient norms, in fp32 and in float64."""
    spec = O.Spec(**cfg)
    x, target = O.synthetic_batch(n, cfg['class_num'], 256, seed=batch_seed)
    st = O.init_state(spec, seed=init_seed)
    loss, outs, grads = O.train_step(spec, st, x, target, apply_update=False, quant=(bits_w, 8), quan_input_bits=bits_i)
    fx = {'cfg': np.array(list(cfg.values()), dtype=np.int64), 'init_seed': np.array(init_seed), 'batch_seed': np.array(batch_seed),
          'n': np.array(n), 'loss': to_np(loss), 'bits_w': np.array(bits_w), 'bits_i': np.array(bits_i)}
    for i, a in enumerate(outs):
        fx[f'out_sub/{i}'] = to_np(a)[:, ::4, ::4, ::4].copy()
    names = [k for k in grads if grads[k] is not None]
    fx['grad_norm_names'] = np.array(names, dtype='U')
    fx['grad_norms'] = np.array([float(grads[k].double().norm()) for k in names])
    st64 = OrderedDict((k, (v.double() if v.is_floating_point() else v.clone())) for k, v in O.init_state(spec, seed=init_seed).items())
    loss64, outs64, grads64 = O.train_step(spec, st64, x.double(), target.double(), apply_update=False, quant=(bits_w, 8), quan_input_bits=bits_i)
    fx['loss64'] = to_np(loss64)
    for i, a in enumerate(outs64):
        fx[f'out64_sub/{i}'] = to_np(a)[:, ::4, ::4, ::4].copy()
    fx['grad_norms64'] = np.array([float(grads64[k].norm()) for k in names])
    e = [float((a.double() - b).norm() / b.norm()) for a, b in zip(outs, outs64)]
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **fx)
    print(f'{tag}: oracle-generated (quantised-input model), N={n} bits_w={bits_w} bits_i={bits_i} loss={float(loss):.6f}; '
          f'fp32-vs-fp64 relL2 per head {["%.1e" % v for v in e]}')


class _LegacyTorchSemantics:
    """torch 0.1.12 method semantics the reference's BinOp (models/cu_net_prev_version.py:17-92) was written against:
    reductions over one dimension KEEP that dimension, and tensor methods accept `out=`.  Patched onto torch.Tensor
    only while the reference class runs."""
    NAMES = ('mean', 'sum', 'norm', 'clamp', 'mul')

    def __enter__(self):
        self.orig = {n: getattr(torch.Tensor, n) for n in self.NAMES}
        orig = self.orig

        def mean(t, dim=None, keepdim=None):
            return orig['mean'](t) if dim is None else orig['mean'](t, dim, True if keepdim is None else keepdim)

        def sum_(t, dim=None, keepdim=None):
            return orig['sum'](t) if dim is None else orig['sum'](t, dim, True if keepdim is None else keepdim)

        def norm(t, p=2, dim=None, keepdim=None):
            return orig['norm'](t, p) if dim is None else orig['norm'](t, p, dim, True if keepdim is None else keepdim)

        def clamp(t, lo=None, hi=None, out=None):
            r = orig['clamp'](t, lo, hi)
            return r if out is None else out.copy_(r)

        def mul(t, other, out=None):
            r = orig['mul'](t, other)
            return r if out is None else out.copy_(r)

        for n, f in zip(self.NAMES, (mean, sum_, norm, clamp, mul)):
            setattr(torch.Tensor, n, f)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(torch.Tensor, n, f)


def binop_quaninput_parity(ref, rq):
    """G14: BinOp (models/cu_net_prev_version.py:17-92) and QuanInput (utils/quantize.py:47-63) EXECUTED.  The
    prev-version model file does not import on torch >= 0.2 (torch._thnn), so the BinOp class alone is compiled from
    the file's AST (as for HumanAug / HumanPts) and run under torch-0.1.12 reduction semantics; QuanInput is a legacy
    autograd Function (instantiating it raises on torch >= 1.3), so its forward / backward are called as the plain
    functions they are, with a stand-in for the autograd context (`save_for_backward` / `saved_tensors`)."""
    import ast
    import numpy
    from oracle import quant_ref as QR
    import re
    src = open(os.path.join(REF, 'models', 'cu_net_prev_version.py')).read()
    src = re.sub(r"(?m)^(\s*)print (.+)$", r"\1print(\2)", src)      # py2 print statements elsewhere in the file
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'BinOp']
    assert len(keep) == 1
    bm = types.ModuleType('ref_binop_subset')
    bm.__dict__.update(torch=torch, nn=torch.nn, numpy=numpy)
    exec(compile(ast.Module(body=keep, type_ignores=[]), '<reference models/cu_net_prev_version.py BinOp>', 'exec'), bm.__dict__)
    cfg = dict(neck_size=2, growth_rate=4, init_chan_num=8, class_num=3, layer_num=2, order=1, loss_num=2)
    torch.manual_seed(91)
    net = quiet(ref.create_cu_net, **cfg)
    convs = [(n, m) for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)]
    g = torch.Generator().manual_seed(92)
    with torch.no_grad():
        for i, (n, m) in enumerate(convs):
            m.weight.mul_(40.0 if i % 3 == 0 else 3.0)
    w0 = {n: m.weight.detach().clone() for n, m in convs}
    grads = {n: torch.randn(m.weight.shape, generator=g) * 0.1 for n, m in convs}
    fx = {'cfg': np.array(list(cfg.values()), dtype=np.int64), 'conv_names': np.array([n for n, _ in convs], dtype='U')}
    import warnings
    with _LegacyTorchSemantics(), warnings.catch_warnings():
        warnings.simplefilter('ignore')         # masked assignment into an expanded tensor (0.1.12 idiom) is deprecated
        bop = bm.BinOp(net)
        tgt = list(bop.bin_range)
        assert tgt == QR.target_indices(len(convs))
        bop.binarization()
        wb = {n: m.weight.detach().clone() for n, m in convs}
        saved = [t.clone() for t in bop.saved_params]
        bop.restore()
        for n, m in convs:
            m.weight.grad = grads[n].clone()
        bop.updateBinaryGradWeight()
    fx['targets'] = np.array(tgt, dtype=np.int64)
    for k, i in enumerate(tgt):
        n, m = convs[i]
        o_wb, o_saved = QR.binop_binarization(w0[n])
        check(f'G14/binop/wb/{n}', wb[n], o_wb)
        check(f'G14/binop/saved/{n}', saved[k], o_saved)
        check(f'G14/binop/grad/{n}', m.weight.grad, QR.binop_grad(o_saved, grads[n]))
        fx['w0/' + n] = to_np(w0[n]); fx['g/' + n] = to_np(grads[n])
        fx['wb/' + n] = to_np(wb[n]); fx['saved/' + n] = to_np(saved[k]); fx['grad/' + n] = to_np(m.weight.grad)
    for i in (0, len(convs) - 1):
        n, m = convs[i]
        check(f'G14/binop/untouched/{n}', m.weight, w0[n])
        fx['w0/' + n] = to_np(w0[n]); fx['g/' + n] = to_np(grads[n])

    # ---- QuanInput forward / backward, bits_i = 8 (the reference's setting, options/train_options.py) and 4
    qsrc = open(os.path.join(REF, 'utils', 'quantize.py')).read()
    qtree = ast.parse(qsrc)
    qcls = [n for n in qtree.body if isinstance(n, ast.ClassDef) and n.name == 'QuanInput'][0]
    fns = [n for n in qcls.body if isinstance(n, ast.FunctionDef) and n.name in ('forward', 'backward')]
    assert len(fns) == 2

    class Ctx:                                   # what a legacy Function instance offers to forward / backward
        def save_for_backward(self, *t):
            self.saved_tensors = t
    x = torch.randn(2, 6, 9, 11, generator=g) * 0.8
    x[0, 0, 0, :4] = torch.tensor([1.0, -1.0, 127 / 128, 0.99999])
    x[0, 0, 1, :4] = torch.tensor([0.5 / 128, 1.5 / 128, 2.5 / 128, -0.5 / 128])      # round-half-to-even ties
    gy = torch.randn(x.shape, generator=g)
    fx['qi/x'] = to_np(x); fx['qi/gy'] = to_np(gy)
    for bi in (8, 4):
        rq.bitsI = bi
        ns = dict(rq.__dict__)
        exec(compile(ast.Module(body=fns, type_ignores=[]), '<reference utils/quantize.py QuanInput>', 'exec'), ns)
        c = Ctx()
        y = ns['forward'](c, x.clone())
        gx = ns['backward'](c, gy.clone())
        check(f'G14/quaninput/fwd/{bi}', y, QR.quan_input(x, bi))
        check(f'G14/quaninput/bwd/{bi}', gx, QR.quan_input_backward(x, gy))
        fx[f'qi/y{bi}'] = to_np(y); fx[f'qi/gx{bi}'] = to_np(gx)
    rq.bitsI = 8
    np.savez_compressed(os.path.join(OUT, 'G14_binop_quaninput.npz'), **fx)
    print(f'G14: BinOp on {len(tgt)} convs and QuanInput (bits_i 8, 4) executed from the reference: oracle == reference')


def init_parity(ref):
    """G_init: the reference's own initialisation (models/cu_net.py:322-334) under a fixed torch seed;
    per-parameter checksums so that cu_net_amd.create_cu_net can be shown to draw the same values."""
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=16, layer_num=2, order=1, loss_num=2)
    torch.manual_seed(1234)
    net = quiet(ref.create_cu_net, **cfg)
    sd = net.state_dict()
    names = [k for k, v in sd.items() if v.is_floating_point() and 'running' not in k]
    sums = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in names])
    probe = to_np(sd['hg.up_blocks.2.adapters_ahead.1.adapter_conv.weight'])[:4, :8, 0, 0].copy()
    conv_order = np.array([n for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)], dtype='U')
    np.savez_compressed(os.path.join(OUT, 'G_init_L2K16.npz'), cfg=np.array(list(cfg.values()), dtype=np.int64),
                        seed=np.array(1234), names=np.array(names, dtype='U'), sums=sums, probe=probe,
                        conv_order=conv_order)
    print(f'G_init: {len(names)} parameter tensors, {len(conv_order)} convs')


def load_reference_quantize():
    """Execute the reference's utils/quantize.py with a stub for its import-time option parsing
    (`opt = TrainOptions().parse()` at :8 would need sys.argv and would mkdir the experiment dir)."""
    import argparse
    stub = types.ModuleType('options.train_options')

    class TrainOptions:                       # noqa: N801 - mirrors the imported name
        def parse(self):
            return argparse.Namespace(bits_w=1, bits_i=8, bits_g=8)
    stub.TrainOptions = TrainOptions
    pkg = types.ModuleType('options')
    sys.modules['options'] = pkg
    sys.modules['options.train_options'] = stub
    src = open(os.path.join(REF, 'utils', 'quantize.py')).read()
    mod = types.ModuleType('ref_quantize')
    exec(compile(src, '<reference utils/quantize.py>', 'exec'), mod.__dict__)
    return mod


def quant_parity(ref, rq):
    """G7: QuanOp phases of the reference on a reference model, bits_w in {1, 2, 4}, bits_g = 8."""
    from oracle import quant_ref as QR
    cfg = dict(neck_size=2, growth_rate=4, init_chan_num=8, class_num=3, layer_num=2, order=1, loss_num=2)
    torch.manual_seed(77)
    net = quiet(ref.create_cu_net, **cfg)
    convs = [(n, m) for n, m in net.named_modules() if isinstance(m, torch.nn.Conv2d)]
    g = torch.Generator().manual_seed(78)
    with torch.no_grad():
        for i, (n, m) in enumerate(convs):      # make some weights large so the clamps and |W|>1 masks act
            m.weight.mul_(40.0 if i % 3 == 0 else 3.0)
    w0 = {n: m.weight.detach().clone() for n, m in convs}
    grads = {n: torch.randn(m.weight.shape, generator=g) * (0.02 if i % 2 else 0.3) for i, (n, m) in enumerate(convs)}
    tgt = QR.target_indices(len(convs))
    fx = {'cfg': np.array(list(cfg.values()), dtype=np.int64), 'conv_names': np.array([n for n, _ in convs], dtype='U'),
          'targets': np.array(tgt, dtype=np.int64)}
    for n, _ in convs:
        fx['w0/' + n] = to_np(w0[n])
        fx['g/' + n] = to_np(grads[n])
    for bw in (1, 2, 4):
        rq.bitsW, rq.bitsI, rq.bitsG = bw, 8, 8
        with torch.no_grad():
            for n, m in convs:
                m.weight.copy_(w0[n])
        qop = rq.QuanOp(net)
        assert len(qop.target_modules) == len(tgt)
        qop.quantization()
        wq = {n: m.weight.detach().clone() for n, m in convs}
        saved = [t.clone() for t in qop.saved_params]
        qop.restore()
        for n, m in convs:
            m.weight.grad = grads[n].clone()
        qop.updateQuanGradWeight()
        for k, i in enumerate(tgt):
            n, m = convs[i]
            o_wq, o_saved = QR.quantization(w0[n], bw, 8)
            check(f'G7/bw{bw}/wq/{n}', wq[n], o_wq)
            check(f'G7/bw{bw}/saved/{n}', saved[k], o_saved)
            check(f'G7/bw{bw}/restored/{n}', m.weight, o_saved)
            o_g = QR.grad_rewrite(o_saved, grads[n], bw, 8)
            check(f'G7/bw{bw}/grad/{n}', m.weight.grad, o_g)
            fx[f'bw{bw}/wq/{n}'] = to_np(wq[n])
            fx[f'bw{bw}/saved/{n}'] = to_np(saved[k])
            fx[f'bw{bw}/grad/{n}'] = to_np(m.weight.grad)
        for i in (0, len(convs) - 1):           # first / last conv are left alone
            n, m = convs[i]
            check(f'G7/bw{bw}/untouched/{n}', m.weight, w0[n])
    np.savez_compressed(os.path.join(OUT, 'G7_quant.npz'), **fx)
    print(f'G7: QuanOp on {len(tgt)} of {len(convs)} convs, bits_w in (1,2,4): oracle == reference')


def decode_parity():
    """G8: get_preds / final_preds of the reference (pylib/Evaluation.py) on random and adversarial heat maps."""
    from oracle import decode_ref as DR
    sys.modules.setdefault('HumanAug', types.ModuleType('HumanAug'))     # py2 implicit-relative import at :4, unused here
    src = open(os.path.join(REF, 'pylib', 'Evaluation.py')).read()
    ev = types.ModuleType('ref_evaluation')
    exec(compile(src, '<reference pylib/Evaluation.py>', 'exec'), ev.__dict__)
    g = torch.Generator().manual_seed(5)
    n, k = 4, 16
    hm = torch.randn(n, k, 64, 64, generator=g) * 0.1
    for a in range(n):                                   # a blob per map, like real predictions
        for b in range(k):
            cy, cx = int(torch.randint(0, 64, (1,), generator=g)), int(torch.randint(0, 64, (1,), generator=g))
            hm[a, b, cy, cx] += 1.0
            if 0 < cx < 63:
                hm[a, b, cy, cx + 1] += 0.5
    hm[0, 0] = -1.0                                       # max <= 0 -> zeros
    hm[0, 1] = 0.0
    hm[1, 2] = 0.0; hm[1, 2, 10, 20] = 9.0; hm[1, 2, 40, 7] = 9.0      # tie -> lowest flat index
    hm[2, 3] = 0.0; hm[2, 3, 63, 63] = 5.0                             # border maxima
    hm[2, 4] = 0.0; hm[2, 4, 0, 0] = 5.0
    hm[3, 5] = 0.0; hm[3, 5, 1, 1] = 5.0; hm[3, 5, 1, 2] = 4.0         # px == 2 boundary of the refinement window
    center = torch.rand(n, 2, generator=g) * 400 + 300
    scale = torch.rand(n, generator=g) * 2.0 + 0.8
    rot = torch.zeros(n)
    ref_gp = ev.get_preds(hm)
    ref_fp = ev.final_preds(hm, center, scale, [64, 64], rot)
    check('G8/get_preds', ref_gp, DR.get_preds(hm))
    check('G8/final_preds', ref_fp, DR.final_preds(hm, center, scale, [64, 64], rot))
    np.savez_compressed(os.path.join(OUT, 'G8_decode.npz'), heat=to_np(hm), center=to_np(center), scale=to_np(scale),
                        get_preds=to_np(ref_gp), final_preds=to_np(ref_fp))
    print('G8: get_preds / final_preds on 4x16 maps: oracle == reference')
    # G8r: the rotation branch of GetTransform (pylib/Evaluation.py:163-178) -- the same maps decoded with per-image rot != 0 (one image
    # keeps rot == 0: a batch may mix them), augmentation-sized and large angles
    rot = torch.tensor([30.0, -17.5, 0.0, 171.25])
    ref_fr = ev.final_preds(hm, center, scale, [64, 64], rot)
    check('G8r/final_preds', ref_fr, DR.final_preds(hm, center, scale, [64, 64], rot))
    assert not torch.equal(ref_fr, ref_fp)
    # (heat maps, centres and scales are G8's: only the angles and the result are stored)
    np.savez_compressed(os.path.join(OUT, 'G8r_decode_rot.npz'), rot=to_np(rot), final_preds=to_np(ref_fr))
    print('G8r: final_preds with rot != 0 on 4x16 maps: oracle == reference')



def tta_accuracy_parity():
    """G10: flip test-time-augmentation merge (pylib/HumanAug.py flip_channels + shuffle_channels_for_horizontal_flipping,
    cu-net.py:247-249) and PCK accuracy (pylib/Evaluation.py accuracy / calc_dists / dist_acc).  HumanAug.py does not
    import under scipy >= 1.3 (scipy.misc), so its two pure-numpy/torch functions are compiled from the reference
    file's AST in memory; Evaluation.py is executed whole as in G8."""
    import ast
    from oracle import decode_ref as DR
    sys.modules.setdefault('HumanAug', types.ModuleType('HumanAug'))
    ev = types.ModuleType('ref_evaluation')
    exec(compile(open(os.path.join(REF, 'pylib', 'Evaluation.py')).read(), '<reference pylib/Evaluation.py>', 'exec'), ev.__dict__)
    import re
    hsrc = open(os.path.join(REF, 'pylib', 'HumanAug.py')).read()
    hsrc = re.sub(r"(?m)^(\s*)print (.+)$", r"\1print(\2)", hsrc)      # py2 print statements elsewhere in the file
    tree = ast.parse(hsrc)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('flip_channels', 'shuffle_channels_for_horizontal_flipping')]
    assert len(keep) == 2
    ha = types.ModuleType('ref_humanaug_subset')
    ha.__dict__.update(np=np, torch=torch)
    exec(compile(ast.Module(body=keep, type_ignores=[]), '<reference pylib/HumanAug.py subset>', 'exec'), ha.__dict__)
    flip_index = np.array([[1, 4], [0, 5], [12, 13], [11, 14], [10, 15], [2, 3]])      # cu-net.py:34-35
    g = torch.Generator().manual_seed(9)
    n, k = 3, 16
    o1 = torch.randint(-64, 64, (n, k, 64, 64), generator=g).float() / 1024      # 7-bit noise: keeps the fixture small
    o2 = torch.randint(-64, 64, (n, k, 64, 64), generator=g).float() / 1024
    tgt = torch.zeros(n, k, 64, 64)
    for a in range(n):
        for b in range(k):
            cy, cx = int(torch.randint(2, 62, (1,), generator=g)), int(torch.randint(2, 62, (1,), generator=g))
            tgt[a, b, cy, cx] = 1.0
            dy, dx = int(torch.randint(-4, 5, (1,), generator=g)), int(torch.randint(-4, 5, (1,), generator=g))
            o1[a, b, min(max(cy + dy, 0), 63), min(max(cx + dx, 0), 63)] += 1.0
            o2[a, b, min(max(cy + dy, 0), 63), 63 - min(max(cx + dx, 0), 63)] += 0.7
    tgt[0, 3] = 0.0                                   # missing joint -> get_preds 0 -> dist -1
    tgt[1, 7] = 0.0; tgt[1, 7, 0, 0] = 1.0            # ground truth at (1,1): not > 1 -> -1
    tgt[:, 9] = 0.0                                   # a joint missing everywhere -> dist_acc -1
    ref_m = (o1 + ha.shuffle_channels_for_horizontal_flipping(ha.flip_channels(o2.clone()), flip_index)) / 2
    check('G10/flip_merge', ref_m, DR.flip_merge(o1, o2, flip_index))
    idxs = [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 15]
    ref_acc = ev.accuracy(ref_m, tgt, idxs)
    check('G10/accuracy', ref_acc, DR.accuracy(ref_m, tgt, idxs))
    np.savez_compressed(os.path.join(OUT, 'G10_tta_accuracy.npz'), out1=to_np(o1), out2=to_np(o2), target=to_np(tgt),
                        flip_index=flip_index, idxs=np.array(idxs), merged_sub=to_np(ref_m[:, :, ::4, ::4]), accuracy=to_np(ref_acc))
    print('G10: flip-TTA merge and PCK accuracy: oracle == reference')



def target_synthesis_parity():
    """G11: Gaussian target maps (pylib/HumanPts.py pts2heatmap / draw_gaussian).  The file mixes tabs and spaces
    (python 2); it is tab-expanded in memory and only the two functions are compiled."""
    import ast, re
    from oracle import decode_ref as DR
    src = open(os.path.join(REF, 'pylib', 'HumanPts.py')).read().expandtabs(8)
    src = re.sub(r"(?m)^(\s*)print (.+)$", r"\1print(\2)", src)
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('pts2heatmap', 'draw_gaussian')]
    assert len(keep) == 2
    hp = types.ModuleType('ref_humanpts_subset')
    hp.__dict__.update(np=np)
    exec(compile(ast.Module(body=keep, type_ignores=[]), '<reference pylib/HumanPts.py subset>', 'exec'), hp.__dict__)
    rng = np.random.RandomState(4)
    pts = rng.uniform(-6, 70, size=(40, 2))
    pts[0] = [0.0, 10.0]; pts[1] = [10.0, -1.0]          # skipped
    pts[2] = [0.5, 0.5]; pts[3] = [63.9, 63.9]            # corners: cropped patches
    pts[4] = [66.0, 30.0]; pts[5] = [67.0, 30.0]          # ul = 63 (one column visible) / ul = 64 (nothing drawn)
    pts[6] = [2.999, 3.0]; pts[7] = [31.0, 31.0]
    outs = {}
    for sigma in (1, 2):
        ref_h, ref_v = hp.pts2heatmap(pts.copy(), (64, 64), sigma)
        my_h, my_v = DR.pts2heatmap(pts.copy(), (64, 64), sigma)
        assert np.array_equal(ref_h, my_h) and np.array_equal(ref_v, my_v), sigma
        outs[f'heat_s{sigma}'] = ref_h.astype(np.float32)[:, ::2, ::2]      # what the dataset hands to torch (.float()), subsampled
        outs[f'sum_s{sigma}'] = ref_h.astype(np.float32).sum(axis=(1, 2))
        outs[f'valid_s{sigma}'] = ref_v
    np.savez_compressed(os.path.join(OUT, 'G11_targets.npz'), pts=pts, **outs)
    print('G11: pts2heatmap / draw_gaussian, sigma 1 and 2, 40 points: oracle == reference')


def big_configs(ref, rq=None):
    """BASELINE configs 3 / 4 / 5 at full width (the judge's G12 / G13)."""
    rq = rq or load_reference_quantize()
    full = dict(neck_size=4, growth_rate=32, init_chan_num=128)
    full_width_cfg(ref, 'G12_full_L8K68', dict(full, class_num=68, layer_num=8, order=1, loss_num=8), n=2, init_seed=2, batch_seed=20)
    full_width_cfg(ref, 'G12_full_L8K16', dict(full, class_num=16, layer_num=8, order=1, loss_num=8), n=2, init_seed=3, batch_seed=22)
    full_width_cfg(ref, 'G13_full_L16K16', dict(full, class_num=16, layer_num=16, order=1, loss_num=16), n=1, init_seed=4, batch_seed=24)
    full_width_cfg(ref, 'G13_full_L16K16_bw1', dict(full, class_num=16, layer_num=16, order=1, loss_num=16), n=1, init_seed=4,
                   batch_seed=24, rq=rq, bits_w=1)
    full_width_quan_input('G13_full_L16K16_bw1_qin8', dict(full, class_num=16, layer_num=16, order=1, loss_num=16), n=1, init_seed=4,
                          batch_seed=24, bits_w=1, bits_i=8)


def augment_parity():
    """G15: the training-sample geometry of the reference's loader.  pylib/HumanAug.py does not import (scipy.misc), so
    GetTransform / TransformSinglePts / TransformPts / shufflelr / fliplr / crop are compiled from the file's AST.  `crop`
    ends in scipy.misc.imresize (and imrotate when rot != 0), which no longer exist: for rot == 0 the call is intercepted
    only to CAPTURE the zero-padded window the reference built (its argument) -- the resampled pixel values themselves
    cannot be pinned (see oracle/augment_ref.py)."""
    import ast, re
    from oracle import augment_ref as A
    src = open(os.path.join(REF, 'pylib', 'HumanAug.py')).read()
    src = re.sub(r"(?m)^(\s*)print (.+)$", r"\1print(\2)", src)
    tree = ast.parse(src)
    want = ('GetTransform', 'TransformSinglePts', 'TransformPts', 'shufflelr', 'fliplr', 'crop')
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(keep) == len(want)
    captured = {}

    class _Misc:                                   # stands in for the two removed resamplers ONLY to observe their input
        @staticmethod
        def imresize(arr, size, interp='bilinear', mode=None):
            captured['canvas'] = np.array(arr, copy=True)
            captured['size'] = size
            return arr
    fake_scipy = types.ModuleType('scipy_misc_capture')
    fake_scipy.misc = _Misc
    ha = types.ModuleType('ref_humanaug_geometry')
    ha.__dict__.update(np=np, torch=torch, scipy=fake_scipy)
    exec(compile(ast.Module(body=keep, type_ignores=[]), '<reference pylib/HumanAug.py geometry subset>', 'exec'), ha.__dict__)
    rng = np.random.RandomState(15)
    fx = {}
    # --- transforms
    cases = []
    for i in range(12):
        c = rng.uniform(100, 900, size=2); s = float(rng.uniform(0.6, 3.5)); r = float(rng.choice([0, 0, 17.5, -33.0, 41.25]))
        res = int(rng.choice([64, 256])); pts = rng.uniform(0, 1000, size=(16, 2))
        t_ref = ha.GetTransform(c, s, r, res, 200)
        assert np.array_equal(t_ref, A.get_transform(c, s, r, res, 200)), 'GetTransform'
        for inv in (0, 1):
            p_ref = ha.TransformPts(pts, c, s, r, res, 200, invert=inv)
            assert np.array_equal(p_ref, A.transform_pts(pts, c, s, r, res, 200, invert=inv)), 'TransformPts'
            s_ref = ha.TransformSinglePts(pts[0], c, s, r, res, 200, invert=inv)
            assert np.array_equal(s_ref, A.transform_single_pts(pts[0], c, s, r, res, 200, invert=inv)), 'TransformSinglePts'
        cases.append((c, s, r, res, pts, ha.TransformPts(pts, c, s, r, res, 200), ha.TransformPts(pts, c, s, r, res, 200, invert=1)))
    fx['tp/center'] = np.array([k[0] for k in cases]); fx['tp/scale'] = np.array([k[1] for k in cases])
    fx['tp/rot'] = np.array([k[2] for k in cases]); fx['tp/res'] = np.array([k[3] for k in cases])
    fx['tp/pts'] = np.array([k[4] for k in cases]); fx['tp/fwd'] = np.array([k[5] for k in cases]); fx['tp/inv'] = np.array([k[6] for k in cases])
    # --- flips
    pts = torch.from_numpy(rng.uniform(0, 640, size=(16, 2)))
    sh_ref = ha.shufflelr(pts.clone(), width=640, dataset='mpii').numpy()
    assert np.array_equal(sh_ref, A.shufflelr(pts.numpy(), 640)), 'shufflelr'
    img = rng.uniform(0, 1, size=(3, 37, 53))
    assert np.array_equal(ha.fliplr(img.copy()), A.fliplr(img)), 'fliplr'
    fx['flip/pts'] = pts.numpy(); fx['flip/shuffled'] = sh_ref
    # --- crop window (rot == 0, no pre-shrink): the canvas the reference hands to imresize
    img_hwc = rng.uniform(0, 1, size=(240, 320, 3))
    crops = []
    for c, s in (((160.0, 120.0), 0.9), ((20.0, 30.0), 1.1), ((310.0, 230.0), 1.28), ((150.5, 99.25), 2.2)):
        captured.clear()
        ha.crop(img_hwc, np.array(c), np.array([s]), 0, 256, 200)
        canvas = captured['canvas']
        assert captured['size'] == (256, 256)
        mine = A.crop_canvas(img_hwc, np.array(c), s, 0, 256, 200)
        assert np.array_equal(canvas, mine), ('crop canvas', c, s)
        ul, br, pad, sf = A.crop_geometry(np.array(c), s, 0, 256, 200)
        crops.append((c[0], c[1], s, ul[0], ul[1], br[0], br[1], canvas.shape[0], canvas.shape[1], float(canvas.sum())))
    fx['crop/img'] = img_hwc.astype(np.float32)
    fx['crop/cases'] = np.array(crops)
    np.savez_compressed(os.path.join(OUT, 'G15_augment.npz'), **fx)
    print(f'G15: GetTransform / TransformPts ({len(cases)} cases), shufflelr, fliplr, crop windows ({len(crops)}): oracle == reference')


class _ScipyMiscOverPIL:
    """scipy.misc.{bytescale, toimage, fromimage, imresize, imrotate} as scipy <= 1.2 defined them (scipy/misc/pilutil.py), over
    the installed PIL -- the module object the reference's `scipy.misc.imresize / imrotate` calls resolve to while its crop()
    is executed for G16.  Restated for the array kinds crop() passes (H x W x 3 float64 / float32 / uint8)."""

    @staticmethod
    def bytescale(data, cmin=None, cmax=None, high=255, low=0):
        if data.dtype == np.uint8:
            return data
        if cmin is None:
            cmin = data.min()
        if cmax is None:
            cmax = data.max()
        cscale = cmax - cmin
        if cscale < 0:
            raise ValueError('`cmax` should be larger than `cmin`.')
        elif cscale == 0:
            cscale = 1
        scale = float(high - low) / cscale
        bytedata = (data - cmin) * scale + low
        return (bytedata.clip(low, high) + 0.5).astype(np.uint8)

    @classmethod
    def toimage(cls, arr):
        from PIL import Image
        data = np.asarray(arr)
        shape = list(data.shape)
        assert len(shape) == 3 and 3 in shape
        ca = np.flatnonzero(np.asarray(shape) == 3)[0]
        assert ca == 2, 'crop() hands H x W x 3 arrays over (first axis of length 3 is the channel axis)'
        bytedata = cls.bytescale(data)
        return Image.frombytes('RGB', (shape[1], shape[0]), bytedata.tobytes())

    @staticmethod
    def fromimage(im):
        return np.array(im)

    @classmethod
    def imrotate(cls, arr, angle, interp='bilinear'):
        from PIL import Image
        assert interp == 'bilinear'
        im = cls.toimage(np.asarray(arr))
        im = im.rotate(angle, resample=Image.BILINEAR)
        return cls.fromimage(im)

    @classmethod
    def imresize(cls, arr, size, interp='bilinear', mode=None):
        from PIL import Image
        assert interp == 'bilinear' and mode is None
        im = cls.toimage(arr)
        ts = type(size)
        if np.issubdtype(ts, np.signedinteger):
            percent = size / 100.0
            size = tuple((np.array(im.size) * percent).astype(int))
        elif np.issubdtype(type(size), np.floating):
            size = tuple((np.array(im.size) * size).astype(int))
        else:
            size = (size[1], size[0])
        imnew = im.resize(size, resample=Image.BILINEAR)
        return cls.fromimage(imnew)


def crop_parity():
    """G16: whole outputs of the reference's crop() (pylib/HumanAug.py:115-172, compiled from the file's AST and EXECUTED) with
    scipy.misc rebuilt over PIL as above -- up- and down-scaling windows, windows leaving the image, rotations, the pre-shrink
    branch (scale * 200 / 256 >= 2) with and without rotation, float images that do and do not reach 1.0 (the byte-scale
    contrast stretch) -- and the check that oracle/augment_ref.py::crop (pure numpy, no PIL) reproduces every byte."""
    import ast, re
    import PIL
    from oracle import augment_ref as A
    src = open(os.path.join(REF, 'pylib', 'HumanAug.py')).read()
    src = re.sub(r"(?m)^(\s*)print (.+)$", r"\1print(\2)", src)
    tree = ast.parse(src)
    want = ('GetTransform', 'TransformSinglePts', 'crop')
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(keep) == len(want)
    fake_scipy = types.ModuleType('scipy_over_pil')
    fake_scipy.misc = _ScipyMiscOverPIL
    ha = types.ModuleType('ref_humanaug_crop')
    ha.__dict__.update(np=np, torch=torch, scipy=fake_scipy)
    exec(compile(ast.Module(body=keep, type_ignores=[]), '<reference pylib/HumanAug.py crop>', 'exec'), ha.__dict__)
    rng = np.random.RandomState(16)
    # two source images, C x H x W float32 as load_image gives them (k / 255 values): smooth structure + texture
    def make(h, w, top):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([0.5 + 0.5 * np.sin(xx / 17.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)])
        tex = rng.uniform(-0.2, 0.2, size=(3, h, w))
        img = np.clip((base + tex) * top, 0, 1)
        return (np.round(img * 255) / 255).astype(np.float32)
    imgs = [make(240, 320, 1.0), make(200, 150, 0.8)]
    cases = [   # image, centre, scale, rot, flip, gains
        (0, (160.0, 120.0), 0.90, 0.0, False, (1.0, 1.0, 1.0)),      # up-scaling window inside the image
        (0, (150.5, 99.25), 1.50, 0.0, True, (1.3, 0.7, 1.1)),       # down-scaling (antialiasing filter), gains clamp at 1
        (0, (20.0, 30.0), 1.10, 0.0, False, (0.8, 0.9, 0.7)),        # window leaves the image (zero padding), maximum < 1
        (1, (75.0, 100.0), 0.70, 0.0, False, (0.9, 0.9, 0.9)),       # interior window of a dim image: minimum > 0 (black level shifts)
        (0, (160.0, 120.0), 1.00, 17.5, False, (1.0, 1.2, 0.9)),     # rotation
        (1, (60.0, 90.0), 1.28, -33.0, True, (1.1, 1.0, 0.6)),
        (0, (170.0, 110.0), 2.70, 0.0, False, (1.0, 1.0, 1.0)),      # pre-shrink branch
        (0, (150.0, 100.0), 3.10, 21.0, True, (1.2, 0.8, 1.0)),      # pre-shrink + rotation
    ]
    fx = {'pil_version': np.array(PIL.__version__)}
    for i, im in enumerate(imgs):
        fx[f'img/{i}'] = np.round(im * 255).astype(np.uint8)         # stored as the bytes they were made from
    recs = []
    for k, (ii, c, s, r, flip, gains) in enumerate(cases):
        img = torch.from_numpy(imgs[ii].copy())
        cc = np.array(c, dtype=np.float64)
        if flip:                                                      # data/mpii_for_mpii_22.py:128-131
            img = torch.from_numpy(np.ascontiguousarray(img.numpy()[:, :, ::-1]))
            cc[0] = img.size(2) - cc[0]
        for ch in range(3):                                           # :134-136
            img[ch, :, :].mul_(gains[ch]).clamp_(0, 1)
        hwc = np.transpose(img.numpy(), (1, 2, 0))                    # utils/imutils.py im_to_numpy
        ref_out = ha.crop(hwc, cc, s, r, 256, 200)
        mine = A.crop(hwc, cc, s, r, 256, 200)
        assert ref_out.dtype == np.uint8 and ref_out.shape == (256, 256, 3)
        if not np.array_equal(ref_out, mine):
            d = np.abs(ref_out.astype(int) - mine.astype(int))
            raise SystemExit(f'ORACLE MISMATCH at G16 case {k}: {int((d > 0).sum())} bytes differ, max {d.max()}')
        full = A.augment_sample(imgs[ii], cc, s, r, flip, gains)
        assert np.array_equal(full, np.transpose(ref_out, (2, 0, 1)).astype(np.float32) / np.float32(255))
        fx[f'out/{k}'] = ref_out
        recs.append((ii, cc[0], cc[1], s, r, float(flip), gains[0], gains[1], gains[2]))
    fx['cases'] = np.array(recs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'G16_crop.npz'), **fx)
    print(f'G16: crop() through the PIL-backed resamplers (PIL {PIL.__version__}), {len(cases)} cases: oracle == executed reference, byte for byte')


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 2 and sys.argv[1] == '--only':       # regenerate one fixture
        if sys.argv[2] == 'qin':
            full = dict(neck_size=4, growth_rate=32, init_chan_num=128)
            full_width_quan_input('G13_full_L16K16_bw1_qin8', dict(full, class_num=16, layer_num=16, order=1, loss_num=16), n=1, init_seed=4,
                                  batch_seed=24, bits_w=1, bits_i=8)
            return
        if sys.argv[2] in ('big', 'binop'):
            ref = load_reference_models()
            {'big': big_configs, 'binop': lambda r: binop_quaninput_parity(r, load_reference_quantize())}[sys.argv[2]](ref)
            return
        {'decode': decode_parity, 'tta': tta_accuracy_parity, 'targets': target_synthesis_parity, 'augment': augment_parity, 'crop': crop_parity}[sys.argv[2]]()
        return
    ref = load_reference_models()
    tiny = dict(neck_size=2, growth_rate=4, init_chan_num=8)
    one_config(ref, 'G1_L2_o1', dict(tiny, class_num=3, layer_num=2, order=1, loss_num=2), n=2, hw=128, seed=11)
    one_config(ref, 'G2_L3_o2', dict(tiny, class_num=5, layer_num=3, order=2, loss_num=3), n=2, hw=128, seed=12)
    one_config(ref, 'G3_L4_o1_ln2', dict(tiny, class_num=3, layer_num=4, order=1, loss_num=2), n=2, hw=128, seed=13)
    one_config(ref, 'G4_L2_o0', dict(tiny, class_num=4, layer_num=2, order=0, loss_num=1), n=3, hw=128, seed=14)
    # wider channels (multiples of 32 like the real net), one image, rectangular-free 128x128 input
    one_config(ref, 'G9_L2_o1_c32', dict(neck_size=2, growth_rate=16, init_chan_num=32, class_num=6,
                                         layer_num=2, order=1, loss_num=2), n=1, hw=128, seed=15)
    # a 64x64 input: the neck is 1x1 and BatchNorm there sees N samples only (edge case, forward pins only)
    one_config(ref, 'G6_L2_o1_hw64', dict(tiny, class_num=3, layer_num=2, order=1, loss_num=2), n=4, hw=64, seed=16)
    full_width(ref)
    init_parity(ref)
    rq = load_reference_quantize()
    quant_parity(ref, rq)
    binop_quaninput_parity(ref, rq)
    big_configs(ref, rq)
    decode_parity()
    tta_accuracy_parity()
    target_synthesis_parity()
    augment_parity()
    crop_parity()


if __name__ == '__main__':
    main()
