"""CPU oracle for the CU-Net hot path -- TEST INFRASTRUCTURE ONLY.

This is a pure-PyTorch (CPU, fp32) *functional* restatement of the reference network
`models/cu_net.py` and of the train-step math of `cu-net.py:171-183`.  It exists so that
the HIP path can be checked against something that runs anywhere; it is NOT part of the
product.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; `cu_net_amd/` never does (the product fails loudly without its HIP library).

Pinning: `tools/gen_golden.py` imports the real reference from /root/reference (through an
in-memory py2->py3 shim) and checks this file against it bit-for-bit
(outputs, loss, every gradient, running statistics after one train step); the resulting
vectors are committed under `tests/golden/` and re-checked by `tests/test_oracle_golden.py`.

Structure (deliberately unlike the reference's nn.Module tree): the network is described
by a flat `Spec`, parameters live in a plain `dict name -> tensor` with the reference's
state_dict keys, and the forward is a function over that dict.

Reference citations are `file:line` relative to the reference tree.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, models/cu_net.py:22,41,45,195,301
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
NUM_BLOCKS = 4       # models/cu_net.py:232


@dataclass
class Spec:
    """Hyper-parameters of `create_cu_net` (models/cu_net.py:362-368)."""
    neck_size: int
    growth_rate: int
    init_chan_num: int
    class_num: int
    layer_num: int
    order: int
    loss_num: int
    loss_anchors: List[int] = field(default_factory=list)

    def __post_init__(self):
        # models/cu_net.py:274-287
        assert 1 <= self.loss_num <= self.layer_num
        every = float(self.layer_num) / float(self.loss_num)
        self.loss_anchors = []
        for i in range(self.loss_num):
            a = int(round(every * (i + 1)))
            if a <= self.layer_num:
                self.loss_anchors.append(a)
        assert self.layer_num in self.loss_anchors
        assert self.loss_num == len(self.loss_anchors)
        if self.order >= self.layer_num:
            raise ValueError('order is larger than the layer number.')


def _carried(i: int, order: int) -> int:
    """Number of carried feature maps seen by U-Net index i (FIFO of depth `order`)."""
    return min(i, order)


def _block_entries(prefix: str, in_num: int, spec: Spec, requires_skip: bool, is_up: bool):
    """Parameter entries of one dense block, in reference registration order
    (models/cu_net.py:75-112)."""
    g, L, K = spec.growth_rate, spec.layer_num, spec.order
    bott = spec.neck_size * g
    out = []
    for i in range(L):
        cin = in_num + _carried(i, K) * g
        p = f'{prefix}.layers.{i}'
        out += _bn_entries(f'{p}.norm1', cin)
        out.append((f'{p}.conv1.weight', (bott, cin, 1, 1), 'param'))
        out += _bn_entries(f'{p}.norm2', bott)
        out.append((f'{p}.conv2.weight', (g, bott, 3, 3), 'param'))
    adapter_out = in_num // 2 if is_up else in_num
    names = ['adapters_ahead'] + (['adapters_skip'] if requires_skip else [])
    for nm in names:
        for i in range(L):
            cin = in_num + (_carried(i, K) + 1) * g
            p = f'{prefix}.{nm}.{i}'
            out += _bn_entries(f'{p}.adapter_norm', cin)
            out.append((f'{p}.adapter_conv.weight', (adapter_out, cin, 1, 1), 'param'))
    return out


def _bn_entries(prefix: str, c: int):
    return [(f'{prefix}.weight', (c,), 'param'), (f'{prefix}.bias', (c,), 'param'),
            (f'{prefix}.running_mean', (c,), 'buffer'), (f'{prefix}.running_var', (c,), 'buffer'),
            (f'{prefix}.num_batches_tracked', (), 'counter')]


def state_entries(spec: Spec) -> List[Tuple[str, tuple, str]]:
    """(name, shape, kind) for every state_dict entry in the reference's order
    (models/cu_net.py:299-320: features, hg{down,up,neck}, linears, intermedia)."""
    c0 = spec.init_chan_num
    e = [('features.conv0.weight', (c0, 3, 7, 7), 'param')]
    e += _bn_entries('features.norm0', c0)
    for j in range(NUM_BLOCKS):
        e += _block_entries(f'hg.down_blocks.{j}', c0, spec, True, False)
    for j in range(NUM_BLOCKS):
        e += _block_entries(f'hg.up_blocks.{j}', 2 * c0, spec, False, True)
    e += _block_entries('hg.neck_block', c0, spec, False, False)
    for i in range(spec.layer_num):
        e += _bn_entries(f'linears.{i}.norm', c0)
        e.append((f'linears.{i}.conv.weight', (spec.class_num, c0, 1, 1), 'param'))
    for i in range(spec.layer_num - 1):
        # models/cu_net.py:156-162: in_num + (i+1)*out_num for i < max_link else in_num + max_link*out_num
        cin = c0 + (i + 1) * c0 if i < spec.order else c0 + spec.order * c0
        e += _bn_entries(f'intermedia.adapters.{i}.adapter_norm', cin)
        e.append((f'intermedia.adapters.{i}.adapter_conv.weight', (c0, cin, 1, 1), 'param'))
    return e


def init_state(spec: Spec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Random state with the reference's init *distributions* (models/cu_net.py:322-334):
    conv U(-1/sqrt(k*k*Cin), +), BN gamma U(0,1), beta 0, running stats (0, 1)."""
    gen = torch.Generator().manual_seed(seed)
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape, kind in state_entries(spec):
        if kind == 'counter':
            st[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith('running_mean'):
            st[name] = torch.zeros(shape)
        elif name.endswith('running_var'):
            st[name] = torch.ones(shape)
        elif len(shape) == 4:
            stdv = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
            st[name] = (torch.rand(shape, generator=gen) * 2 - 1) * stdv
        elif name.endswith('.bias'):
            st[name] = torch.zeros(shape)
        else:
            st[name] = torch.rand(shape, generator=gen)
    return st


class _Ctx:
    """Per-forward bookkeeping: which BNs the reference would re-run in backward."""
    def __init__(self, state, training, quan_input_bits=0, storage='fp32'):
        self.state = state
        self.training = training
        self.recomputed: List[Tuple[str, torch.Tensor]] = []
        self.quan_input_bits = quan_input_bits
        assert storage in ('fp32', 'bf16', 'bf16_grads')
        self.storage = storage


# ---- bf16 storage modes (BASELINE config 3; NOT a reference feature -- the reference is fp32).  The HIP path with bf16
# storage (cunet_forward_bf16, include/cunet.h) rounds to bf16, round-to-nearest-even, at these points and computes in fp32
# everywhere else; `storage='bf16'` / `'bf16_grads'` restates exactly those points so that a whole-step comparison has an
# oracle that is rounded where the kernels round:
#   * every tensor a node STORES (stem BN-ReLU-pool output, every conv output except the heads, pooled tensors -- a max of
#     bf16 values is exact) -- and therefore the batch statistics every consumer BatchNorm derives from it;
#   * the activated operand relu(bn(x)) on its way into the contraction, and the forward conv weights (bf16 MFMA operands);
#   * 'bf16_grads' only: every gradient TENSOR of backward -- d(loss)/d(stored tensor) (heads included) and dz, the
#     gradient at a BatchNorm output -- as tensor hooks that round the accumulated gradient once, as the per-tensor
#     gather / data-gradient kernels store it; parameter gradients stay fp32; d(loss)/d(stem conv output) stays fp32.
# Rounding is straight-through for autograd (the kernels differentiate the fp32 expression at the rounded values).
def _round_bf16(t: torch.Tensor) -> torch.Tensor:
    return t + (t.detach().bfloat16().float() - t.detach())


def _grad_bf16(t: torch.Tensor) -> torch.Tensor:
    if t.requires_grad:
        t.register_hook(lambda g: g.bfloat16().float())
    return t


def _stored(ctx: "_Ctx", t: torch.Tensor, is_head: bool = False) -> torch.Tensor:
    """What a node writes to memory in the context's storage mode."""
    if ctx.storage == 'fp32':
        return t
    if not is_head:
        t = _round_bf16(t)
    if ctx.storage == 'bf16_grads':
        t = _grad_bf16(t)
    return t


class _QuanInputFn(torch.autograd.Function):
    """utils/quantize.py:47-63 as a modern autograd Function: forward Q(C(x, bits), bits), backward straight-through
    with no gradient where |x| >= 1.  Forward / backward are oracle/quant_ref.py's quan_input / quan_input_backward,
    which tools/gen_golden.py pins to the EXECUTED reference class (G14)."""

    @staticmethod
    def forward(ctx, x, bits):
        from oracle import quant_ref as QR
        ctx.save_for_backward(x)
        return QR.quan_input(x, bits)

    @staticmethod
    def backward(ctx, g):
        from oracle import quant_ref as QR
        (x,) = ctx.saved_tensors
        return QR.quan_input_backward(x, g), None


def _bn_relu_conv(ctx: _Ctx, inputs: List[torch.Tensor], bn: str, conv: str, pad: int,
                  checkpointed: bool, quan_site: bool = False, is_head: bool = False) -> torch.Tensor:
    """cat -> BN -> ReLU -> conv  (models/cu_net.py:11-17).  `quan_site`: one of the places where the reference's
    quantised model puts a QuanInput2d between the ReLU and the conv (models/cu_net_prev_version_wig.py:96-98 the 3x3
    convs, :277-279 the heads); active when the forward was asked for quantised inputs."""
    st = ctx.state
    x = torch.cat(inputs, 1) if len(inputs) > 1 else inputs[0]
    if ctx.training:
        if checkpointed and torch.is_grad_enabled():
            ctx.recomputed.append((bn, x.detach()))
        st[bn + '.num_batches_tracked'] += 1
    y = F.batch_norm(x, st[bn + '.running_mean'], st[bn + '.running_var'],
                     st[bn + '.weight'], st[bn + '.bias'], ctx.training, BN_MOMENTUM, BN_EPS)
    w = st[conv + '.weight']
    if ctx.storage != 'fp32':
        if ctx.storage == 'bf16_grads':
            y = _grad_bf16(y)                    # dz is stored as bf16
        y = _round_bf16(F.relu(y))               # the activated bf16 MFMA operand
        w = _round_bf16(w)                       # the bf16 weight operand
        return _stored(ctx, F.conv2d(y, w, None, 1, pad), is_head)
    y = F.relu(y)
    if quan_site and ctx.quan_input_bits:
        y = _QuanInputFn.apply(y, ctx.quan_input_bits)
    return F.conv2d(y, w, None, 1, pad)


def _dense_block(ctx: _Ctx, prefix: str, xs: List[torch.Tensor], i: int, saved: List[torch.Tensor],
                 order: int, requires_skip: bool):
    """models/cu_net.py:115-144 with the FIFO passed in explicitly."""
    xs = list(xs) + list(saved)
    p = f'{prefix}.layers.{i}'
    z = _bn_relu_conv(ctx, xs, f'{p}.norm1', f'{p}.conv1', 0, True)       # :53-61 (checkpointed)
    out = _bn_relu_conv(ctx, [z], f'{p}.norm2', f'{p}.conv2', 1, False, quan_site=True)   # :62
    if i < order:                                                          # :133-137
        saved.append(out)
    elif len(saved) != 0:
        saved.pop(0)
        saved.append(out)
    xs = xs + [out]                                                        # :138
    pa = f'{prefix}.adapters_ahead.{i}'
    ahead = _bn_relu_conv(ctx, xs, f'{pa}.adapter_norm', f'{pa}.adapter_conv', 0, True)
    if requires_skip:
        ps = f'{prefix}.adapters_skip.{i}'
        skip = _bn_relu_conv(ctx, xs, f'{ps}.adapter_norm', f'{ps}.adapter_conv', 0, True)
        return ahead, skip
    return ahead, None


def forward(spec: Spec, state: Dict[str, torch.Tensor], x: torch.Tensor,
            training: bool = True, ctx_out: list | None = None, quan_input_bits: int = 0,
            storage: str = 'fp32') -> List[torch.Tensor]:
    """`_CU_Net_Wrapper.forward` (models/cu_net.py:336-360).

    In training mode BN running statistics in `state` are updated in place exactly once per
    BN (the effect of the reference's *forward*); `finish_backward_stat_updates` applies the
    extra update the reference's checkpoint recompute performs during `backward()`.
    """
    ctx = _Ctx(state, training, quan_input_bits, storage)
    if storage != 'fp32' and quan_input_bits:
        raise ValueError('the quantised-input mode is fp32 only')
    st = state
    # stem, :299-304
    y = F.conv2d(x, st['features.conv0.weight'], None, 2, 3)
    if training:
        st['features.norm0.num_batches_tracked'] += 1
    y = F.batch_norm(y, st['features.norm0.running_mean'], st['features.norm0.running_var'],
                     st['features.norm0.weight'], st['features.norm0.bias'], training,
                     BN_MOMENTUM, BN_EPS)
    y = _stored(ctx, F.max_pool2d(F.relu(y), 2, 2))      # (the stem itself runs in fp32 in every storage mode)

    K = spec.order
    saved_blocks: Dict[str, List[torch.Tensor]] = {}
    inter_saved: List[torch.Tensor] = []
    outs = []
    cur = y
    for i in range(spec.layer_num):
        # intermedia, :166-190
        if i == 0:
            inter_saved = [cur] if K != 0 else []
        else:
            xs = [cur] + inter_saved
            p = f'intermedia.adapters.{i - 1}'
            cur = _bn_relu_conv(ctx, xs, f'{p}.adapter_norm', f'{p}.adapter_conv', 0, True)
            if i < K:
                inter_saved.append(cur)
            elif len(inter_saved) != 0:
                inter_saved.pop(0)
                inter_saved.append(cur)
        if i == 0:
            saved_blocks = {}
        # hourglass, :252-269
        h = cur
        skips = [None] * NUM_BLOCKS
        for j in range(NUM_BLOCKS):
            name = f'hg.down_blocks.{j}'
            h, skips[j] = _dense_block(ctx, name, [h], i, saved_blocks.setdefault(name, []), K, True)
            h = F.max_pool2d(h, 2, 2)
            if ctx.storage == 'bf16_grads':
                h = _grad_bf16(h)                # (the pooled values are exact; its gradient tensor is stored as bf16)
        name = 'hg.neck_block'
        h, _ = _dense_block(ctx, name, [h], i, saved_blocks.setdefault(name, []), K, False)
        for j in reversed(range(NUM_BLOCKS)):
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            name = f'hg.up_blocks.{j}'
            h, _ = _dense_block(ctx, name, [h, skips[j]], i, saved_blocks.setdefault(name, []), K, False)
        cur = h
        if (i + 1) in spec.loss_anchors:                                   # :353-356
            p = f'linears.{i}'
            outs.append(_bn_relu_conv(ctx, [cur], f'{p}.norm', f'{p}.conv', 0, False, quan_site=True, is_head=True))
    if ctx_out is not None:
        ctx_out.append(ctx)
    return outs


def finish_backward_stat_updates(ctx: _Ctx) -> None:
    """Second running-stat update of every checkpointed BN, as performed by the reference's
    `cp.checkpoint` recompute during backward (models/cu_net.py:30-31,58-59)."""
    st = ctx.state
    with torch.no_grad():
        for bn, x in ctx.recomputed:
            # re-running train-mode BN on the same input is literally what the recompute does
            F.batch_norm(x, st[bn + '.running_mean'], st[bn + '.running_var'],
                         st[bn + '.weight'], st[bn + '.bias'], True, BN_MOMENTUM, BN_EPS)
            st[bn + '.num_batches_tracked'] += 1
        ctx.recomputed = []


def mse_loss(outputs: List[torch.Tensor], target: torch.Tensor) -> torch.Tensor:
    """cu-net.py:175-178: sum over heads of mean squared error."""
    loss = 0
    for o in outputs:
        d = (o - target) ** 2
        loss = loss + d.sum() / d.numel()
    return loss


def param_names(spec: Spec) -> List[str]:
    return [n for n, _, k in state_entries(spec) if k == 'param']


def conv_weight_names(spec: Spec) -> List[str]:
    """nn.Conv2d weights in the reference's modules() order (== state_dict order)."""
    return [n for n, shape, k in state_entries(spec) if k == 'param' and len(shape) == 4]


def train_step(spec: Spec, state: Dict[str, torch.Tensor], x: torch.Tensor, target: torch.Tensor,
               opt_state: Dict[str, torch.Tensor] | None = None, lr: float = 2.5e-4,
               alpha: float = 0.99, eps: float = 1e-8, apply_update: bool = True, quant=None, quan_input_bits: int = 0,
               storage: str = 'fp32'):
    """One optimisation step (cu-net.py:171-183) with RMSprop(lr, alpha, eps) (cu-net.py:60-61).

    `quant=(bits_w, bits_g)` wraps the step in QuanOp.quantization / restore / updateQuanGradWeight
    (utils/quantize.py:104-175, loop placement cu-net-prev-version-wig.py:163-190).
    `storage` = 'bf16' / 'bf16_grads': the bf16 storage points of the HIP path (see `_stored`).
    Returns (loss, outputs, grads dict).  `state` is updated in place (running stats always,
    parameters when `apply_update`).  Parameters whose gradient is None (non-anchor heads)
    are skipped by the optimiser, as torch.optim.RMSprop does.
    """
    names = param_names(spec)
    qnames = []
    if quant is not None:       # (bits_w, bits_g): cu-net-prev-version-wig.py:163-190 around the same step
        from oracle import quant_ref as QR
        convs = conv_weight_names(spec)
        qnames = [convs[i] for i in QR.target_indices(len(convs))]
        latents = {}
        with torch.no_grad():
            for n in qnames:
                wq, saved = QR.quantization(state[n].detach(), quant[0], quant[1])
                latents[n] = saved
                state[n].copy_(wq)
    for n in names:
        state[n].requires_grad_(True)
        state[n].grad = None
    ctxs: list = []
    outs = forward(spec, state, x, True, ctxs, quan_input_bits=quan_input_bits, storage=storage)
    loss = mse_loss(outs, target)
    loss.backward()
    finish_backward_stat_updates(ctxs[0])
    grads = {n: (state[n].grad.detach().clone() if state[n].grad is not None else None) for n in names}
    if quant is not None:
        with torch.no_grad():
            for n in qnames:
                state[n].copy_(latents[n])                                   # restore()
                grads[n] = QR.grad_rewrite(latents[n], grads[n], quant[0], quant[1])   # updateQuanGradWeight()
    if apply_update:
        if opt_state is None:
            opt_state = {}
        with torch.no_grad():
            for n in names:
                g = grads[n]
                if g is None:
                    continue
                v = opt_state.setdefault(n, torch.zeros_like(state[n]))
                v.mul_(alpha).addcmul_(g, g, value=1 - alpha)
                state[n].addcdiv_(g, v.sqrt().add_(eps), value=-lr)
    for n in names:
        state[n].requires_grad_(False)
        state[n].grad = None
    return loss.detach(), [o.detach() for o in outs], grads


def synthetic_batch(n: int, class_num: int, hw: int = 256, seed: int = 0):
    """BASELINE.md section 3 inputs: x ~ U[0,1) (seed), targets = one 7x7 Gaussian blob
    exp(-(dx^2+dy^2)/9) per landmark at integer centres U{3..res-4}^2 (seed+1)
    (blob as pylib/HumanPts.py:49-76 with sigma=1: tmp_size=3, g=exp(-d^2/tmp_size^2))."""
    g0 = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, hw, hw, generator=g0)
    res = hw // 4
    g1 = torch.Generator().manual_seed(seed + 1)
    cx = torch.randint(3, res - 3, (n, class_num), generator=g1)
    cy = torch.randint(3, res - 3, (n, class_num), generator=g1)
    ax = torch.arange(7, dtype=torch.float32) - 3
    blob = torch.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / 9.0)
    t = torch.zeros(n, class_num, res, res)
    for a in range(n):
        for k in range(class_num):
            yy, xx = int(cy[a, k]), int(cx[a, k])
            t[a, k, yy - 3:yy + 4, xx - 3:xx + 4] = blob
    return x, t
