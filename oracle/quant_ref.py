"""CPU oracle for the weight / activation / gradient quantisers -- TEST INFRASTRUCTURE ONLY.

Restates, as pure functions over tensors, the arithmetic of the reference's
    utils/quantize.py:15-42    S, C, Q
    utils/quantize.py:47-63    QuanInput (forward quantisation + straight-through backward)
    utils/quantize.py:77-175   QuanOp.quantization / restore / updateQuanGradWeight
    models/cu_net_prev_version.py:17-92   BinOp (XNOR-Net style, scale kept, no 8-bit rounding)

Pinning: `tools/gen_golden.py` executes the reference's utils/quantize.py (with a stub for its
import-time option parsing) on a reference model and checks these functions against it; the vectors
are committed as tests/golden/G7_quant.npz.  `QuanInput` is a legacy autograd Function (instantiating
it raises on torch >= 1.3) and BinOp relies on torch-0.1.12 keepdim semantics (mean(1) keeps the
dimension) inside a file that no longer imports: both are nevertheless pinned BY EXECUTION --
gen_golden.py compiles the BinOp class and QuanInput.forward / .backward from the reference files' AST
and runs them (BinOp under a context that restores the 0.1.12 reduction semantics on torch.Tensor),
checks this file against them bit-for-bit and commits the vectors as tests/golden/G14_binop_quaninput.npz.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def S(bits: int) -> float:                      # utils/quantize.py:15-16
    return 2.0 ** (bits - 1)


def C(x: torch.Tensor, bits: int = 32) -> torch.Tensor:   # utils/quantize.py:20-28
    delta = 0.0 if (bits > 15 or bits == 1 or bits == 2) else 1.0 / S(bits)
    return torch.clamp(x, -1 + delta, +1 - delta)


def Q(x: torch.Tensor, bits: int) -> torch.Tensor:        # utils/quantize.py:33-42
    if bits > 15:
        return x
    if bits == 1:
        return torch.sign(x)
    if bits == 2:
        return torch.round(x)
    sc = S(bits)
    return torch.round(x * sc) / sc


def quan_input(x: torch.Tensor, bits_i: int = 8) -> torch.Tensor:
    """QuanInput.forward (utils/quantize.py:52-55): Q(C(x, bits_i), bits_i)."""
    return Q(C(x, bits_i), bits_i)


def quan_input_backward(x: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """QuanInput.backward (utils/quantize.py:58-63): straight-through, zero where |x| >= 1."""
    g = grad_out.clone()
    g[x.ge(1)] = 0
    g[x.le(-1)] = 0
    return g


def target_indices(num_convs: int) -> List[int]:
    """QuanOp/BinOp.__init__ (utils/quantize.py:85-90): every nn.Conv2d in modules() order except the
    first and the last."""
    return list(range(1, num_convs - 1))


def quantization(w: torch.Tensor, bits_w: int = 1, bits_g: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    """One conv weight [O, I, kh, kw] through QuanOp.quantization (utils/quantize.py:104-149).
    Returns (quantised weight used in forward/backward, saved latent restored afterwards)."""
    w = w + w.mean(1, True).mul(-1).expand_as(w)            # meancenterConvParams :110-115
    w = C(w, bits_g)                                        # clampConvParams :117-119
    saved = Q(w, bits_g)                                    # save_params :121-123
    if bits_w == 1:                                         # quantizeConvParams :125-149
        n = w[0].nelement()
        m = w.norm(1, 3, True).sum(2, True).sum(1, True).div(n).expand(w.size())
        m = Q(m, bits_g)
        w = w.sign().mul(m)
    if bits_w == 2:
        n = w[0].nelement()
        d = w.norm(1, 3, True).sum(2, True).sum(1, True).div(n).mul(0.7)
        wt = torch.empty_like(w)
        for col in range(w.shape[0]):
            dc = d[col, 0, 0, 0]
            wt[col] = w[col].gt(1.0 * dc).float().add(w[col].lt(-1.0 * dc).float().mul(-1))
        w = wt
    else:
        # NB: also taken for bits_w == 1 (an `if`, not `elif`, at :135): sign(clamp(sign(W)*m)) drops the scale
        w = Q(C(w, bits_w), bits_w)
    return w, saved


def grad_rewrite(w: torch.Tensor, g: torch.Tensor, bits_w: int = 1, bits_g: int = 8) -> torch.Tensor:
    """QuanOp.updateQuanGradWeight for one conv (utils/quantize.py:156-175); `w` is the RESTORED latent."""
    if bits_w == 1:
        n = w[0].nelement()
        s = w.size()
        m = w.norm(1, 3, True).sum(2, True).sum(1, True).div(n).expand(s).clone()
        m[w.lt(-1.0)] = 0
        m[w.gt(1.0)] = 0
        m = Q(m, bits_g)
        m = m.mul(g)
        m_add = w.sign().mul(g)
        m_add = m_add.sum(3, True).sum(2, True).sum(1, True).div(n).expand(s)
        m_add = m_add.mul(w.sign())
        g = m.add(m_add).mul(1.0 - 1.0 / s[1]).mul(n)
    return Q(C(g, bits_g), bits_g)


# ---- BinOp (models/cu_net_prev_version.py:17-92), torch-0.1.12 keepdim semantics restated ----------
def binop_binarization(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    w = w - w.mean(1, True)                       # :50-55
    w = w.clamp(-1.0, 1.0)                        # :57-60
    saved = w.clone()                             # :62-64 (no rounding)
    n = w[0].nelement()
    m = w.norm(1, 3, True).sum(2, True).sum(1, True).div(n)       # :66-72 (same reduction order as the reference)
    return w.sign().mul(m.expand(w.size())), saved


def binop_grad(w: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    n = w[0].nelement()                           # :78-92
    s = w.size()
    m = w.norm(1, 3, True).sum(2, True).sum(1, True).div(n).expand(s).clone()
    m[w.lt(-1.0)] = 0         # (never true in the reference's call order: `w` is the restored, clamped latent)
    m[w.gt(1.0)] = 0
    m = m.mul(g)
    m_add = w.sign().mul(g).sum(3, True).sum(2, True).sum(1, True).div(n).expand(s).mul(w.sign())
    return m.add(m_add).mul(1.0 - 1.0 / s[1]).mul(n)


def ternary_conv_reference(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, w: torch.Tensor,
                           bits_i: int = 8, pad: int = 0) -> torch.Tensor:
    """BN(eval, folded to scale/shift) -> ReLU -> QuanInput(bits_i) -> conv with weights in {-1,0,+1}:
    what the XNOR/AND-popcount kernel computes (placement of QuanInput2d before the 3x3 convs and heads,
    models/cu_net_prev_version_wig.py:96-98,277-279).  Every product and partial sum is a multiple of
    2^-(bits_i-1) below 2^16, i.e. exact in fp32, so this is a bit-exact target."""
    a = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    a = quan_input(a, bits_i)
    return torch.nn.functional.conv2d(a, w, None, 1, pad)
