"""GPU (-m gpu): BASELINE.json configs 3, 4 and 5 at FULL width through the C ABI.

  config 3  CU-Net layer_num=8 order=1 loss_num=8, K=68 (bf16 storage on one MI355X)   -> G12_full_L8K68
  config 4  the same network with K=16 (the per-rank shard of the 8-GPU data-parallel job)  -> G12_full_L8K16
  config 5  CU-Net layer_num=16 order=1 loss_num=16, K=16, QuanOp bits_w=1                  -> G13_full_L16K16[_bw1]

The fixtures were captured from the REFERENCE (tools/gen_golden.py `big_configs`: reference model + reference
QuanOp, one train step on the oracle's seeded init / batch, oracle == reference checked at generation time);
heat maps are stored sub-sampled (::4 in every dimension) with the loss, per-parameter gradient norms and
running-statistic sums.

Tolerances
  fp32        heat maps / loss: 1e-4 relative per U-Net pair of depth (north_star's bound is stated for the L=2 net;
              the error of an fp32 evaluation grows with depth: L/2 * 1e-4 of the tensor's magnitude, which is what
              torch-CPU fp32 itself shows against fp64 at these depths).  Gradient norms: +-5 % (whole-net fp32
              gradients are chaotic, tests/test_gpu_nodes.py holds the exact per-kernel checks).
  bf16        storage rounds every activation to 8 mantissa bits (relative step 2^-8 = 3.9e-3, rms 1.1e-3).  A heat
              map at U-Net i sits behind ~18 i stored tensors on its longest path; with independent roundings the
              relative-L2 error is ~1.1e-3 * sqrt(18 i) * g, g ~ 2 (BatchNorm re-normalisation gain on these nets):
              i = 8 -> 2.7e-2.  Bounds: relL2 <= 5e-2, max <= 8e-2 of the heat-map range, loss within 2e-2.
  bits_w = 1  the reference's binarised net has +-1 weights without scale (utils/quantize.py:148-149): activations
              reach 1e2..1e3 and the loss 2.8e2; fp32 heat maps are compared at 2e-3 of their magnitude and relL2 1e-3
              (a sign(W) decision on a latent that mean-centring left within rounding of 0 flips a weight: the
              quantiser test allows 1e-3 of the elements to differ for that reason).
Measured values are written to gpurun_out/parity_configs_*.txt.
"""
import os

import pytest
import torch

import cu_net_amd
from cu_net_amd.trainer import FusedTrainer
from oracle import cunet_ref as O
from tests._golden import Golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(tag, lines):
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f'parity_configs_{tag}.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    except OSError:
        pass


def _setup(tag):
    g = Golden(tag)
    spec = O.Spec(**g.cfg)
    st = O.init_state(spec, seed=int(g.z['init_seed']))
    n = int(g.z['n'])
    x, target = O.synthetic_batch(n, spec.class_num, 256, seed=int(g.z['batch_seed']))
    net = cu_net_amd.create_cu_net(**g.cfg)
    net.load_state_dict(st)
    net.cuda().train()
    return g, spec, net, x, target


def _check_step(tag, mode, quan_bits=0):
    g, spec, net, x, target = _setup(tag)
    L = spec.layer_num
    quan = None
    if quan_bits:
        from cu_net_amd.quant import QuanOp
        quan = QuanOp(net, bits_w=quan_bits, bits_i=8, bits_g=8)
    tr = FusedTrainer(net, quan_op=quan, bf16=mode != 'fp32', bf16_grads=mode == 'bf16_grads')
    # the optimiser step would move the parameters: the fixture holds gradients, so compare before it matters
    loss = tr.step(x.cuda(), target.cuda())
    outs = tr.last_outputs(x.shape)
    torch.cuda.synchronize()
    lines, bad = [], []
    ref_loss = float(g.z['loss'])
    if mode == 'fp32':
        rt_out = (2e-3 if quan_bits else 1e-4 * max(1.0, L / 2))
        rt_loss = rt_out
    else:
        rt_out, rt_loss = 8e-2, 2e-2
    rel_loss = abs(float(loss) - ref_loss) / abs(ref_loss)
    lines.append(f'loss hip={float(loss):.7g} ref={ref_loss:.7g} rel={rel_loss:.2e} (bound {rt_loss:.1e})')
    if not rel_loss <= rt_loss:
        bad.append('loss')
    assert len(outs) == spec.loss_num
    for i, o in enumerate(outs):
        ref = g.t(f'out_sub/{i}')
        got = o.cpu()[:, ::4, ::4, ::4]
        err = (got - ref).abs().max().item()
        mag = ref.abs().max().item()
        rng = (ref.max() - ref.min()).item()
        rel2 = ((got - ref).double().norm() / ref.double().norm()).item()
        if mode == 'fp32':
            ok = err <= rt_out * mag + 1e-6 and (not quan_bits or rel2 <= 1e-3)
        else:
            ok = err <= rt_out * rng and rel2 <= 5e-2
        lines.append(f'{"ok " if ok else "BAD"} head {i:2d} err={err:.3e} mag={mag:.3e} range={rng:.3e} rel={err / mag:.2e} relL2={rel2:.2e}')
        if not (ok and bool(torch.isfinite(o).all())):
            bad.append(f'head {i}')
    # gradient norms per parameter (sanity bound; quantised step: rewritten + 8-bit-rounded gradients)
    names = g.z['grad_norm_names'].tolist()
    off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
    tol = 5e-2 if mode == 'fp32' and not quan_bits else 0.15
    worst = 0.0
    nb = 0
    for k, nrm in zip(names, g.z['grad_norms']):
        o, nmel = off[k]
        got = float(net._grad_arena[o:o + nmel].double().norm())
        r = abs(got - nrm) / (nrm + 1e-12)
        worst = max(worst, r)
        if r > tol and nrm > 1e-6:
            nb += 1
            lines.append(f'BAD gradnorm {k} hip={got:.4e} ref={nrm:.4e}')
    lines.append(f'gradient norms: {len(names)} parameters, worst relative deviation {worst:.3e} (bound {tol})')
    if nb > (0 if mode == 'fp32' and not quan_bits else len(names) // 100):
        bad.append(f'{nb} gradient norms')
    if mode == 'fp32':       # running statistics after the double (checkpoint) update: sums per buffer
        sd = net.state_dict()
        rw = 0.0
        for k, s in zip(g.z['running_names'].tolist(), g.z['running_sums']):
            got = float(sd[k].double().sum())
            rw = max(rw, abs(got - s) / (abs(s) + 1e-3 * sd[k].numel()))
        lines.append(f'running statistics: worst relative deviation of a buffer sum {rw:.3e}')
        if rw > (2e-2 if quan_bits else 2e-3):
            bad.append('running stats')
    _report(f'{tag}_{mode}' + (f'_bw{quan_bits}' if quan_bits else ''), lines)
    assert not bad, (bad, lines[:12])


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16_grads'])
def test_config3_cu_net8_k68(mode):
    _check_step('G12_full_L8K68', mode)


@pytest.mark.parametrize('mode', ['fp32', 'bf16_grads'])
def test_config4_cu_net8_k16_rank_shard(mode):
    _check_step('G12_full_L8K16', mode)


def test_config5_cu_net16_k16_fp32():
    _check_step('G13_full_L16K16', 'fp32')


def test_config5_cu_net16_k16_bits_w1():
    _check_step('G13_full_L16K16_bw1', 'fp32', quan_bits=1)


def test_every_node_backward_cu_net8():
    """Node-by-node backward (tests/test_gpu_nodes.py) on the L = 8 plan at production widths: intermedia adapters
    i >= 2, FIFO pops at 128-wide channels, the 8-bucket backward order and the large workspace offsets."""
    from tests.test_gpu_nodes import _check_all_nodes
    cfg = dict(neck_size=4, growth_rate=32, init_chan_num=128, class_num=68, layer_num=8, order=1, loss_num=8)
    spec = O.Spec(**cfg)
    st = O.init_state(spec, seed=41)
    x, _ = O.synthetic_batch(1, 68, 256, seed=42)
    _check_all_nodes(cfg, st, x)
