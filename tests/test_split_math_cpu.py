"""CPU (no GPU): the arithmetic identity behind planner option f32_split (cu_net_amd/csrc/conv_common.h split_bf16x3).

An fp32 value is cut into three bf16 pieces, each rounded to nearest even from what the pieces before it left; the kernels contract
six of the nine piece products on the bf16 matrix pipe.  torch's CPU bfloat16 conversion is the same rounding as v_cvt_pk_bf16_f32, so
the claims can be checked here without hardware:
  1. h + m + l == x exactly (8 + 8 + 8 significand bits cover fp32's 24) wherever l does not underflow;
  2. the three dropped products (m l', l m', l l') are below 2^-23 of |x y|;
  3. a K-long dot product of six-piece products, accumulated in fp32, is as close to the float64 result as the plain fp32 dot product is."""
import torch


def split3(x):
    h = x.bfloat16().float()
    r1 = x - h
    m = r1.bfloat16().float()
    r2 = r1 - m
    l = r2.bfloat16().float()
    return h, m, l


def test_three_pieces_are_the_value():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-20, 20, (1 << 16,), generator=g).float())
    h, m, l = split3(x)
    assert torch.equal((h.double() + m.double() + l.double()).float(), x)
    assert torch.equal(h + (m + l), x)
    # each piece is a bf16 value, and each residue is exact in fp32 (the subtraction of a value's own leading bits)
    for p in (h, m, l):
        assert torch.equal(p.bfloat16().float(), p)
    assert float((m.abs() / x.abs().clamp_min(1e-38)).max()) <= 2.0 ** -8 and float((l.abs() / x.abs().clamp_min(1e-38)).max()) <= 2.0 ** -16


def test_dropped_products_are_below_fp32_resolution():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1 << 16, generator=g).double()
    y = torch.randn(1 << 16, generator=g).double()
    xh, xm, xl = (p.double() for p in split3(x.float()))
    yh, ym, yl = (p.double() for p in split3(y.float()))
    kept = xh * yh + xh * ym + xm * yh + xm * ym + xh * yl + xl * yh
    exact = x.float().double() * y.float().double()
    rel = ((kept - exact).abs() / exact.abs().clamp_min(1e-300)).max()
    assert float(rel) <= 2.0 ** -23, float(rel)


def test_six_product_dot_is_as_accurate_as_the_fp32_dot():
    g = torch.Generator().manual_seed(3)
    K = 512
    a = torch.relu(torch.randn(256, K, generator=g) + 0.3)          # post-ReLU-like activations
    b = 0.1 * torch.randn(K, 64, generator=g)
    ref = a.double() @ b.double()
    scale = (a.double().abs() @ b.double().abs())
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    six = torch.zeros(256, 64)
    for p, q in ((ah, bl), (al, bh), (am, bm), (ah, bm), (am, bh), (ah, bh)):     # the kernels' order: small terms first
        six = six + p @ q                                                         # fp32 accumulation
    plain = a @ b
    e_six = float(((six.double() - ref).abs() / scale).max())
    e_plain = float(((plain.double() - ref).abs() / scale).max())
    assert e_six <= 2.0 * e_plain + 1e-8 and e_six <= 5e-7, (e_six, e_plain)
