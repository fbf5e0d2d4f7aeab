"""Weight / gradient quantisation of the CU-Net on the HIP path.

`QuanOp(model)` mirrors the reference's `utils/quantize.py:77-175` (same constructor rule: every
`nn.Conv2d` of `model.modules()` except the first and the last; same three phases, same call order as
`cu-net-prev-version-wig.py:163-190`):

    quan_op.quantization()            # before forward: W <- quantised(W), latent saved
    ... forward / backward ...
    quan_op.restore()                 # W <- latent (8-bit rounded)
    quan_op.updateQuanGradWeight()    # XNOR-Net style gradient rewrite + 8-bit gradient rounding
    optimizer.step()

but each phase is ONE HIP launch over the flat parameter / gradient arenas instead of ~10 tiny
torch kernels per conv.  `BinOp` (models/cu_net_prev_version.py:17-92) is the same machinery with
the per-filter scale kept and no 8-bit rounding.  `ternary_conv` is the multiplier-free AND-popcount
convolution for weights in {-1,0,+1} on `bits_i`-bit activations.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ._lib import CUNetError, check, lib
from .module import CUNet, _ptr, _stream_ptr

_ENTRY = np.dtype([('off', np.int64), ('O', np.int32), ('I', np.int32), ('KK', np.int32), ('pad', np.int32)])


class QuanOp:
    def __init__(self, model, bits_w: int = 1, bits_i: int = 8, bits_g: int = 8, keep_scale: bool = False):
        net = model.module if hasattr(model, 'module') and isinstance(model.module, CUNet) else model
        if not isinstance(net, CUNet):
            raise CUNetError('QuanOp needs a cu_net_amd.CUNet (flat parameter arena)')
        self.net = net
        self.bits_w, self.bits_i, self.bits_g, self.keep_scale = int(bits_w), int(bits_i), int(bits_g), bool(keep_scale)
        convs = [(n, m) for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
        self.target_names = [n for n, _ in convs[1:-1]]          # utils/quantize.py:85-102
        self.target_modules = [m.weight for _, m in convs[1:-1]]
        self.num_of_params = len(self.target_modules)
        off = {name: (o, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
        tab = np.zeros(self.num_of_params, dtype=_ENTRY)
        for i, n in enumerate(self.target_names):
            o, shape = off[n + '.weight']
            tab[i] = (o, shape[0], shape[1], shape[2] * shape[3], 0)
        self._tab_host = tab
        self.max_o = int(tab['O'].max())
        self.max_n = int((tab['I'] * tab['KK']).max())
        self._tab = None
        self.saved = None

    def _device_state(self):
        net = self.net
        dev = net._param_arena.device
        if dev.type != 'cuda':
            raise CUNetError('QuanOp runs on the GPU arenas: call net.cuda() first (the CPU oracle is oracle/quant_ref.py)')
        net._check_aliasing()
        if self._tab is None or self._tab.device != dev:
            self._tab = torch.from_numpy(self._tab_host.view(np.uint8).copy()).to(dev)
        if self.saved is None or self.saved.device != dev or self.saved.data_ptr() == 0:
            self.saved = torch.zeros_like(net._param_arena)
        return net, dev

    def quantization(self):
        net, dev = self._device_state()
        check(lib().cunet_quant_prepare(_ptr(net._param_arena), _ptr(self.saved), _ptr(self._tab), self.num_of_params,
                                        self.max_o, self.max_n, self.bits_w, self.bits_g, 1 if self.keep_scale else 0,
                                        _stream_ptr(dev)), 'cunet_quant_prepare')
        # from here to restore() the target convs hold {-1, 0, +1}: the plan may run them on AND-popcount (module.py)
        net._weights_ternary = self.bits_w in (1, 2) and not self.keep_scale

    def restore(self):
        net, dev = self._device_state()
        net._weights_ternary = False
        check(lib().cunet_quant_restore(_ptr(net._param_arena), _ptr(self.saved), _ptr(self._tab), self.num_of_params,
                                        _stream_ptr(dev)), 'cunet_quant_restore')

    def updateQuanGradWeight(self, grad_arena: torch.Tensor = None):
        """Rewrites the gradients of the target convs in the flat gradient arena (default: the module's own,
        which is what FusedTrainer and `loss.backward()` through this module fill)."""
        net, dev = self._device_state()
        g = net._grad_arena if grad_arena is None else grad_arena
        check(lib().cunet_quant_grad(_ptr(net._param_arena), _ptr(g), _ptr(self._tab), self.num_of_params, self.max_o,
                                     self.bits_w, self.bits_g, 1 if self.keep_scale else 0, _stream_ptr(dev)),
              'cunet_quant_grad')


class BinOp(QuanOp):
    """models/cu_net_prev_version.py:17-92: sign(W) * mean|W| with the scale kept, no 8-bit rounding."""

    def __init__(self, model):
        super().__init__(model, bits_w=1, bits_i=32, bits_g=32, keep_scale=True)

    binarization = QuanOp.quantization
    updateBinaryGradWeight = QuanOp.updateQuanGradWeight


def ternary_conv_planes(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, weight: torch.Tensor, bits_i: int = 8, variant: int = 1):
    """The same operator as `ternary_conv` through the network's own two-kernel path (include/cunet.h cunet_ternary_conv_ex): bit-plane
    records once per tensor, then AND + popcount -- variant 1: lane = pixel, weight masks as scalar operands; 0: wave = pixel.
    Returns (y NCHW, per-channel [2][O] fp64 sums of y and y^2 as the consumer BatchNorms receive them)."""
    if not x.is_cuda:
        raise CUNetError('ternary_conv_planes: GPU tensor required (the CPU oracle is oracle/quant_ref.py)')
    n, c, h, w = x.shape
    o, ci, kh, kw = weight.shape
    assert ci == c and kh == kw and kh in (1, 3)
    taps = kh * kw
    dev = x.device
    xn = x.permute(0, 2, 3, 1).contiguous().float()
    wt = weight.contiguous().float().to(dev)
    opad = (o + 63) // 64 * 64
    words = taps * ((c + 63) // 64) * opad
    wpos = torch.zeros(words, dtype=torch.int64, device=dev)
    wneg = torch.zeros(words, dtype=torch.int64, device=dev)
    st = _stream_ptr(dev)
    check(lib().cunet_ternary_pack(_ptr(wt), _ptr(wpos), _ptr(wneg), o, c, taps, st), 'cunet_ternary_pack')
    planes = torch.empty((n * h * w + 1) * 16, dtype=torch.int64, device=dev)
    y = torch.empty((n, h, w, o), dtype=torch.float32, device=dev)
    stats = torch.zeros((2, o), dtype=torch.float64, device=dev)
    sc = scale.contiguous().float().to(dev)
    sh = shift.contiguous().float().to(dev)
    check(lib().cunet_ternary_conv_ex(_ptr(xn), _ptr(sc), _ptr(sh), _ptr(wpos), _ptr(wneg), _ptr(planes), _ptr(y), _ptr(stats),
                                      n, h, w, c, o, taps, int(bits_i), int(variant), st), 'cunet_ternary_conv_ex')
    return y.permute(0, 3, 1, 2).contiguous(), stats


def ternary_conv(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, weight: torch.Tensor, bits_i: int = 8):
    """y = conv2d(QuanInput_bits_i(relu(x * scale + shift)), weight) for weight in {-1,0,+1}, kernel 1x1 or 3x3 (pad 1),
    computed with AND + popcount over activation bit-planes (no multiplier, no MFMA).  x: N x C x H x W (NCHW, GPU)."""
    if not x.is_cuda:
        raise CUNetError('ternary_conv: GPU tensor required (the CPU oracle is oracle/quant_ref.py)')
    n, c, h, w = x.shape
    o, ci, kh, kw = weight.shape
    assert ci == c and kh == kw and kh in (1, 3)
    taps = kh * kw
    dev = x.device
    xn = x.permute(0, 2, 3, 1).contiguous().float()                     # NHWC
    wt = weight.contiguous().float().to(dev)
    opad = (o + 63) // 64 * 64
    words = taps * ((c + 63) // 64) * opad
    wpos = torch.zeros(words, dtype=torch.int64, device=dev)
    wneg = torch.zeros(words, dtype=torch.int64, device=dev)
    st = _stream_ptr(dev)
    check(lib().cunet_ternary_pack(_ptr(wt), _ptr(wpos), _ptr(wneg), o, c, taps, st), 'cunet_ternary_pack')
    y = torch.empty((n, h, w, o), dtype=torch.float32, device=dev)
    sc = scale.contiguous().float().to(dev)       # keep every device temporary alive in a local: a tensor that dies
    sh = shift.contiguous().float().to(dev)       # inside the argument list hands its block to the next allocation
    check(lib().cunet_ternary_conv(_ptr(xn), _ptr(sc), _ptr(sh), _ptr(wpos), _ptr(wneg), _ptr(y), n, h, w, c, o, taps,
                                   int(bits_i), st), 'cunet_ternary_conv')
    return y.permute(0, 3, 1, 2).contiguous()
