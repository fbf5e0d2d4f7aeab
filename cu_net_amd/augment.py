"""Training-sample preparation on the GPU: the image half of the reference's data loader
(data/mpii_for_mpii_22.py:120-145 -- scale / rotation jitter, horizontal flip, per-channel colour gain, `HumanAug.crop`
to the 256 x 256 network input, `TransformPts` / `shufflelr` for the target points) for a whole batch.

The per-sample geometry (crop window, rotation padding, pre-shrink size) is computed on the host exactly as
pylib/HumanAug.py:10-42,118-142 computes it (float64, the same integer truncation).  The pixels follow the reference's own
route: its resamplers scipy.misc.imresize / imrotate were PIL's 8-bit `Image.resize` / `Image.rotate(BILINEAR)` behind scipy's
byte-scale, so the library keeps crop()'s stages (byte-scale -> [shrink] -> canvas -> [rotate] -> resize -> / 255) with uint8
intermediates and PIL's arithmetic restated exactly; the result equals the reference function executed over PIL bit for bit
(tests/golden/G16_crop.npz; oracle/augment_ref.py is the numpy restatement the GPU tests compare with).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import CUNetError, check, lib
from .module import _ptr, _stream_ptr

MPII_PAIRS = ([0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13])      # pylib/HumanAug.py:238-242

_REC = np.dtype([('src', np.uint64), ('mm', np.uint64), ('i8', np.uint64), ('t1', np.uint64), ('i1', np.uint64), ('c8', np.uint64),
                 ('r8', np.uint64), ('t2', np.uint64), ('o8', np.uint64), ('rm', np.float64, (6,)),
                 ('sh', np.int32), ('sw', np.int32), ('sh1', np.int32), ('sw1', np.int32), ('ulx', np.int32), ('uly', np.int32),
                 ('cw', np.int32), ('ch', np.int32), ('win_w', np.int32), ('win_h', np.int32), ('pad', np.int32),
                 ('flip', np.int32), ('rotated', np.int32), ('pre', np.int32), ('gain', np.float32, (3,)), ('pad_', np.int32)])
assert _REC.itemsize == 192        # == struct AugSample of csrc/common.h


def _translation(dx, dy):
    m = np.eye(3)
    m[0, 2], m[1, 2] = dx, dy
    return m


def get_transform(center, scale, rot, res, size):
    """Image -> crop coordinates as a 3 x 3 homogeneous matrix (pylib/HumanAug.py:10-34): the box of `size * scale` pixels
    around `center` is mapped onto [0, res)^2; a rotation by -rot degrees is taken about the centre of the crop.  The
    factors are multiplied in the reference's order (right to left: map, shift the crop centre to the origin, rotate,
    shift back), which keeps the entries -- and so the truncated point coordinates -- identical to its."""
    box = size * scale
    k = float(res) / box
    m = np.array([[k, 0., res * (0.5 - float(center[0]) / box)],
                  [0., k, res * (0.5 - float(center[1]) / box)],
                  [0., 0., 1.]])
    if rot == 0:
        return m
    phi = -rot * np.pi / 180
    s_, c_ = np.sin(phi), np.cos(phi)
    turn = np.array([[c_, -s_, 0.], [s_, c_, 0.], [0., 0., 1.]])
    return _translation(res / 2, res / 2) @ (turn @ (_translation(-res / 2, -res / 2) @ m))


def transform_pts(pts, center, scale, rot, res, size=200, invert=0):
    """pylib/HumanAug.py:44-52 for a K x 2 array (truncation toward zero, as `.astype(int)`)."""
    pts = np.asarray(pts, dtype=np.float64)
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.dot(t, np.concatenate((pts, np.ones((pts.shape[0], 1))), axis=1).T)
    return new_pt[0:2, :].T.astype(int)


def shufflelr(pts, width, pairs=MPII_PAIRS):
    """pylib/HumanAug.py:234-265: mirror the x coordinates and swap the left / right joints."""
    x = np.array(pts, dtype=np.float64, copy=True)
    x[:, 0] = width - x[:, 0]
    for a, b in pairs:
        x[[a, b]] = x[[b, a]]
    return x


def _geometry(center, scale, rot, res, size, sh, sw):
    """pylib/HumanAug.py:118-142 for an sh x sw image: pre-shrunk size (or None), canvas corners ul / br incl. the rotation
    padding, pad."""
    sf = float(scale * size) / float(res)
    pre = None
    if sf < 2:
        sf = 1
    else:
        if np.floor(max(sh, sw) / sf) < 2:
            return None                                         # HumanAug.crop returns the image unchanged (:124-125)
        frac = 1 / sf                                           # imresize(img, size=1/scale_factor): int(W * frac) x int(H * frac)
        pre = (int(sh * frac), int(sw * frac))
    c = np.asarray(center, dtype=np.float64) / sf
    s = scale / sf

    def single(pt):
        t = np.linalg.inv(get_transform(c, s, 0, res, size))
        return np.dot(t, np.array([pt[0], pt[1], 1.]))[:2].astype(int)
    ul, br = single([0, 0]), single([res, res])
    if sf >= 2:
        br = br - (br - ul - res)
    pad = int(np.ceil(np.linalg.norm(br - ul) / 2 - float(br[1] - ul[1]) / 2))
    if not rot == 0:
        ul = ul - pad
        br = br + pad
    return ul, br, pad, pre


def _pil_rotate_matrix(w, h, angle_deg):
    """Image.rotate(angle) of a w x h image (expand off, centre = image centre): the destination -> source affine map,
    computed the way PIL computes it (entries rounded to 15 decimals)."""
    import math
    angle = angle_deg % 360.0
    cx, cy = w / 2.0, h / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def augment_batch(images, centers, scales, rots=None, flips=None, gains=None, res: int = 256, size: float = 200.0):
    """images: list of C x H x W fp32 GPU tensors in [0, 1] (any sizes); centers N x 2 (x, y) -- already mirrored for flipped
    samples, as the reference does (`c[0] = W - c[0]`); scales N; rots N degrees (0 = none); flips N bool; gains N x 3.
    Returns N x 3 x res x res fp32 (values k / 255: the reference's crop ends in uint8)."""
    n = len(images)
    if n == 0:
        raise CUNetError('augment_batch: empty batch')
    dev = images[0].device
    if dev.type != 'cuda':
        raise CUNetError('augment_batch: GPU tensors required (the CPU oracle is oracle/augment_ref.py)')
    rots = np.zeros(n) if rots is None else np.asarray(rots, dtype=np.float64)
    flips = np.zeros(n, dtype=bool) if flips is None else np.asarray(flips, dtype=bool)
    gains = np.ones((n, 3)) if gains is None else np.asarray(gains, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64).reshape(-1)
    rec = np.zeros(n, dtype=_REC)
    keep = []
    passthrough = []
    need = 0

    def take(nbytes):                     # scratch offsets (16-byte aligned) inside one caller-owned buffer
        nonlocal need
        off = need
        need += (int(nbytes) + 15) // 16 * 16
        return off
    offs = []
    for i, img in enumerate(images):
        if img.dim() != 3 or img.shape[0] != 3 or img.dtype != torch.float32 or img.device != dev:
            raise CUNetError('augment_batch: every image must be a 3 x H x W fp32 tensor on the same GPU')
        img = img.contiguous()
        keep.append(img)
        rot = float(rots[i])
        sh, sw = int(img.shape[1]), int(img.shape[2])
        geo = _geometry(centers[i], float(scales[i]), rot, res, size, sh, sw)
        if geo is None:
            # The box is so large that the pre-shrink would leave fewer than 2 pixels: the reference's crop() hands the image back
            # UNCHANGED (pylib/HumanAug.py:124-125) and the loader goes on with it.  That only yields a sample when the image
            # already has the network's size (anything else fails in the reference's collate): passed through here the same way --
            # flip, colour gains, clamp, no resampling -- after the batch's kernels; any other size is an error for THIS sample.
            if (sh, sw) != (res, res):
                raise CUNetError(f'augment_batch: sample {i}: person box too large to pre-shrink (HumanAug.crop returns the {sh} x {sw} '
                                 f'image unchanged there, :124-125, which is not a {res} x {res} network input)')
            passthrough.append(i)
            geo = _geometry(centers[i], float(res) / size, 0.0, res, size, sh, sw)      # a valid dummy record (a res x res window); its output is overwritten below
            rot = 0.0
        ul, br, pad, pre = geo
        cw, ch = int(br[0] - ul[0]), int(br[1] - ul[1])
        rotated = 1 if rot != 0 else 0
        win_w, win_h = cw - 2 * pad * rotated, ch - 2 * pad * rotated
        if cw < 1 or ch < 1 or win_w < 1 or win_h < 1:
            raise CUNetError('augment_batch: empty crop window')
        r = rec[i]
        r['src'] = img.data_ptr()
        r['sh'], r['sw'] = sh, sw
        r['pre'] = 1 if pre is not None else 0
        if pre is not None:
            r['sh1'], r['sw1'] = pre
        r['ulx'], r['uly'], r['cw'], r['ch'], r['win_w'], r['win_h'], r['pad'] = int(ul[0]), int(ul[1]), cw, ch, win_w, win_h, pad
        r['flip'], r['rotated'] = int(bool(flips[i])), rotated
        r['gain'] = gains[i].astype(np.float32)
        if rotated:
            r['rm'] = _pil_rotate_matrix(cw, ch, rot)
        o = {'mm': take(32), 'c8': take(ch * cw * 3), 't2': take(win_h * res * 3), 'o8': take(res * res * 3)}
        if rotated:
            o['r8'] = take(win_h * win_w * 3)
        if pre is not None:
            o['i8'], o['t1'], o['i1'] = take(sh * sw * 3), take(sh * pre[1] * 3), take(pre[0] * pre[1] * 3)
        offs.append(o)
    scratch = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    base = scratch.data_ptr()
    for i, o in enumerate(offs):
        for k, v in o.items():
            rec[i][k] = base + v
    host = np.ascontiguousarray(rec.view(np.uint8))
    tab = torch.from_numpy(host.copy()).to(dev)
    out = torch.empty((n, 3, res, res), dtype=torch.float32, device=dev)
    check(lib().cunet_augment_batch(_ptr(tab), C.c_void_p(host.ctypes.data), n, _ptr(out), int(res), _stream_ptr(dev)), 'cunet_augment_batch')
    for i in passthrough:
        im = keep[i].flip(2) if flips[i] else keep[i]
        out[i] = (im * torch.as_tensor(gains[i], dtype=torch.float32, device=dev).view(3, 1, 1)).clamp_(0, 1)
    out._cunet_keepalive = (keep, tab, scratch)          # the launches are asynchronous: inputs and scratch must outlive them
    return out


# ---- the loader's per-sample recipe (data/mpii_for_mpii_22.py:86-145) for a whole batch -----------------------------------
def sample_from_bounded_gaussian(x, rng=np.random):
    """data/mpii_for_mpii_22.py:12-13."""
    return max(-2 * x, min(2 * x, rng.randn() * x))


def draw_train_params(scale_factor: float = 0.25, rot_factor: float = 30.0, rng=np.random):
    """The random draws of one training sample in the reference's ORDER (data/mpii_for_mpii_22.py:122-136), so that the same
    numpy seed gives the same augmentation: scale jitter 2**N(0, sf) (bounded at 2 sf), rotation N(0, rf) (bounded) zeroed
    with probability 0.6, flip with probability 0.5, three colour gains U(0.6, 1.4)."""
    s_mul = 2 ** sample_from_bounded_gaussian(scale_factor, rng)
    r = sample_from_bounded_gaussian(rot_factor, rng)
    if rng.uniform(0, 1, 1) <= 0.6:
        r = 0
    flip = bool(rng.random() <= 0.5) if hasattr(rng, 'random') else bool(rng.random_sample() <= 0.5)
    gains = [rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4)]
    return s_mul, r, flip, gains


def mpii_center_scale(objpos, scale_provided, dataset: str = 'MPII'):
    """data/mpii_for_mpii_22.py:99-110: person box -> crop centre / scale."""
    c = np.array(objpos, dtype=np.float64, copy=True)
    s = float(scale_provided)
    if dataset == 'MPII':
        c[1] = c[1] + 15 * s
        s = s * 1.25
    elif dataset == 'LEEDS':
        s = s * 1.4375
    else:
        raise ValueError('no such dataset %s' % dataset)
    return c, s


def prepare_batch(samples, is_train: bool = True, inp_res: int = 256, out_res: int = 64, sigma: float = 1, scale_factor: float = 0.25,
                  rot_factor: float = 30.0, std_size: float = 200.0, rng=np.random, params=None):
    """`MPII.__getitem__` (data/mpii_for_mpii_22.py:86-145) for a batch, everything after the JPEG decode on the GPU.
    samples: dicts with 'img' (3 x H x W fp32 GPU tensor in [0, 1]), 'joint_self' (K x >=2), 'objpos' (x, y), 'scale_provided',
    optionally 'dataset'.  `params` (a list of (s_mul, r, flip, gains)) overrides the random draws.
    Returns (inp N x 3 x inp_res^2, heatmap N x K x out_res^2, meta) with meta = dict(center, scale, rot, pts) as the reference
    returns them per sample (pts AFTER the flip shuffle, centre / scale after jitter)."""
    from .trainer import pts2heatmap
    n = len(samples)
    imgs, cs, ss, rs, fl, gs, pts_all, pts_aug = [], [], [], [], [], [], [], []
    for i, a in enumerate(samples):
        img = a['img']
        pts = np.asarray(a['joint_self'], dtype=np.float64)[:, 0:2].copy()
        c, s = mpii_center_scale(a['objpos'], a['scale_provided'], a.get('dataset', 'MPII'))
        r, flip, gains = 0.0, False, [1.0, 1.0, 1.0]
        if is_train:
            s_mul, r, flip, gains = params[i] if params is not None else draw_train_params(scale_factor, rot_factor, rng)
            s = s * s_mul
            if flip:
                width = img.shape[2]
                pts = shufflelr(pts, width)
                c[0] = width - c[0]
        imgs.append(img); cs.append(c); ss.append(s); rs.append(float(r)); fl.append(bool(flip)); gs.append(gains)
        pts_all.append(pts)
        pts_aug.append(transform_pts(pts, c, s, r, out_res, std_size))
    inp = augment_batch(imgs, np.stack(cs), np.asarray(ss), rs, fl, np.asarray(gs), res=inp_res, size=std_size)
    pa = torch.from_numpy(np.stack(pts_aug).astype(np.float64)).to(inp.device)
    heatmap = pts2heatmap(pa, (out_res, out_res), sigma)
    meta = {'center': np.stack(cs), 'scale': np.asarray(ss), 'rot': np.asarray(rs), 'pts': np.stack(pts_all), 'pts_aug': np.stack(pts_aug)}
    return inp, heatmap, meta
