"""Training-sample preparation on the GPU: the image half of the reference's data loader
(data/mpii_for_mpii_22.py:120-145 -- scale / rotation jitter, horizontal flip, per-channel colour gain, `HumanAug.crop`
to the 256 x 256 network input, `TransformPts` / `shufflelr` for the target points) for a whole batch in one launch.

The per-sample geometry (crop window, rotation padding, pre-shrink factor) is computed on the host exactly as
pylib/HumanAug.py:10-42,118-142 computes it (float64, the same integer truncation); the kernel then takes ONE bilinear
sample per output pixel (and per k x k sub-sample when the reference would shrink the image first) at the composition of
the resize, the rotation and the window offset.  The reference's resamplers (scipy.misc.imresize / imrotate: 8-bit PIL
images with a data-dependent contrast stretch) no longer exist; pixel values are therefore defined by
oracle/augment_ref.py (same geometry, plain bilinear), not by the reference -- see DESIGN.md.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import CUNetError, check, lib
from .module import _ptr, _stream_ptr

MPII_PAIRS = ([0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13])      # pylib/HumanAug.py:238-242

_REC = np.dtype([('src', np.uint64), ('sh', np.int32), ('sw', np.int32), ('ulx', np.int32), ('uly', np.int32),
                 ('win_w', np.int32), ('win_h', np.int32), ('pad', np.int32), ('k', np.int32),
                 ('cw', np.int32), ('ch', np.int32), ('flip', np.int32), ('rotated', np.int32),
                 ('sf', np.float64), ('cs', np.float64), ('sn', np.float64), ('g0', np.float32), ('g1', np.float32), ('g2', np.float32),
                 ('pad_', np.float32)])        # == struct AugSample of csrc/common.h (96 bytes)


def get_transform(center, scale, rot, res, size):
    """pylib/HumanAug.py:10-34."""
    h = size * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res) / h
    t[1, 1] = float(res) / h
    t[0, 2] = res * (-float(center[0]) / h + .5)
    t[1, 2] = res * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    if not rot == 0:
        rot = -rot
        rot_rad = rot * np.pi / 180
        sn, cs = np.sin(rot_rad), np.cos(rot_rad)
        rot_mat = np.array([[cs, -sn, 0.], [sn, cs, 0.], [0., 0., 1.]])
        t_mat = np.eye(3)
        t_mat[0, 2] = -res / 2
        t_mat[1, 2] = -res / 2
        t_inv = t_mat.copy()
        t_inv[:2, 2] *= -1
        t = np.dot(t_inv, np.dot(rot_mat, np.dot(t_mat, t)))
    return t


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


def _geometry(center, scale, rot, res, size):
    """pylib/HumanAug.py:118-142 (window of the source image, rotation padding, pre-shrink factor)."""
    sf_full = float(scale * size) / float(res)
    sf = sf_full if sf_full >= 2 else 1.0
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
    k = int(np.floor(sf_full)) if sf_full >= 2 else 1
    return ul, br, pad, sf, k


def augment_batch(images, centers, scales, rots=None, flips=None, gains=None, res: int = 256, size: float = 200.0):
    """images: list of C x H x W fp32 GPU tensors in [0, 1] (any sizes); centers N x 2 (x, y) -- already mirrored for flipped
    samples, as the reference does (`c[0] = W - c[0]`); scales N; rots N degrees (0 = none); flips N bool; gains N x 3.
    Returns N x 3 x res x res fp32.  One launch for the batch."""
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
    for i, img in enumerate(images):
        if img.dim() != 3 or img.shape[0] != 3 or img.dtype != torch.float32 or img.device != dev:
            raise CUNetError('augment_batch: every image must be a 3 x H x W fp32 tensor on the same GPU')
        img = img.contiguous()
        keep.append(img)
        rot = float(rots[i])
        ul, br, pad, sf, k = _geometry(centers[i], float(scales[i]), rot, res, size)
        cw, ch = int(br[0] - ul[0]), int(br[1] - ul[1])
        rotated = 1 if rot != 0 else 0
        phi = -np.deg2rad(rot)
        rec[i] = (img.data_ptr(), img.shape[1], img.shape[2], int(ul[0]), int(ul[1]), cw - 2 * pad * rotated, ch - 2 * pad * rotated,
                  pad, k, cw, ch, int(bool(flips[i])), rotated, sf, np.cos(phi), np.sin(phi), gains[i, 0], gains[i, 1], gains[i, 2], 0.0)
    tab = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    out = torch.empty((n, 3, res, res), dtype=torch.float32, device=dev)
    check(lib().cunet_augment_batch(_ptr(tab), n, _ptr(out), int(res), _stream_ptr(dev)), 'cunet_augment_batch')
    out._cunet_keepalive = (keep, tab)          # the launch is asynchronous: inputs must outlive it
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
