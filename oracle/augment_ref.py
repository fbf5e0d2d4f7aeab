"""CPU oracle for the training-sample preparation (image half of SURVEY 8f rank 4) -- TEST INFRASTRUCTURE ONLY.

Restates, in numpy, the per-sample work of the reference's data loader between "decoded image" and "network input":
    data/mpii_for_mpii_22.py:120-145   scale / rotation jitter, horizontal flip, per-channel colour gain, crop, target points
    pylib/HumanAug.py:10-57            GetTransform / TransformSinglePts / TransformPts
    pylib/HumanAug.py:115-172          crop (window of the source image on a zero canvas, rotation, resize to res x res)
    pylib/HumanAug.py:234-271          shufflelr / fliplr

Pinning (tools/gen_golden.py --only augment, tests/golden/G15_augment.npz): the transform functions, shufflelr / fliplr and
the WINDOW EXTRACTION of crop (ul, br, pad and the zero-padded canvas handed to the resampler) are checked bit-for-bit
against the reference's own functions, compiled from pylib/HumanAug.py's AST and executed.  The two resamplers the
reference calls -- scipy.misc.imrotate and scipy.misc.imresize -- no longer exist (scipy >= 1.3) and went through an
8-bit PIL image with a data-dependent contrast stretch (scipy.misc.bytescale): that part is PARITY-UNPINNED.  Here, and
in the HIP kernel, the canvas -> network-input map is ONE bilinear sample (pixel centres at half-integers, the PIL
convention) at the composition of the resize and the rotation about the canvas centre; no 8-bit round trip.
"""
from __future__ import annotations

import numpy as np


def get_transform(center, scale, rot, res, size):                      # pylib/HumanAug.py:10-34
    h = size * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res) / h
    t[1, 1] = float(res) / h
    t[0, 2] = res * (-float(center[0]) / h + .5)
    t[1, 2] = res * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    if not rot == 0:
        rot = -rot
        rot_mat = np.zeros((3, 3))
        rot_rad = rot * np.pi / 180
        sn, cs = np.sin(rot_rad), np.cos(rot_rad)
        rot_mat[0, :2] = [cs, -sn]
        rot_mat[1, :2] = [sn, cs]
        rot_mat[2, 2] = 1
        t_mat = np.eye(3)
        t_mat[0, 2] = -res / 2
        t_mat[1, 2] = -res / 2
        t_inv = t_mat.copy()
        t_inv[:2, 2] *= -1
        t = np.dot(t_inv, np.dot(rot_mat, np.dot(t_mat, t)))
    return t


def transform_single_pts(pt, center, scale, rot, res, size, invert=0):  # :36-42
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.dot(t, np.array([pt[0], pt[1], 1.]).T)
    return new_pt[:2].astype(int)


def transform_pts(pts, center, scale, rot, res, size, invert=0):        # :44-52
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.concatenate((pts, np.ones((pts.shape[0], 1))), axis=1).T
    new_pt = np.dot(t, new_pt)
    return new_pt[0:2, :].T.astype(int)


MPII_PAIRS = ([0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13])


def shufflelr(x, width, pairs=MPII_PAIRS):                              # :234-265 (dataset 'mpii')
    x = x.copy()
    x[:, 0] = width - x[:, 0]
    for a, b in pairs:
        tmp = x[a, :].copy()
        x[a, :] = x[b, :]
        x[b, :] = tmp
    return x


def fliplr(img_chw):                                                    # :267-271 on a C x H x W array
    return img_chw[:, :, ::-1].astype(float)


def crop_geometry(center, scale, rot, res, size):
    """Window of pylib/HumanAug.py:118-142: (ul, br, pad, scale_factor) -- ul / br already include the rotation padding."""
    scale_factor = float(scale * size) / float(res)
    if scale_factor < 2:
        scale_factor = 1
    center = np.asarray(center, dtype=float) / scale_factor
    scale = scale / scale_factor
    ul = np.array(transform_single_pts([0, 0], center, scale, 0, res, size, invert=1))
    br = np.array(transform_single_pts([res, res], center, scale, 0, res, size, invert=1))
    if scale_factor >= 2:
        br = br - (br - ul - res)
    pad = np.ceil(np.linalg.norm(br - ul) / 2 - float(br[1] - ul[1]) / 2).astype(int)
    if not rot == 0:
        ul = ul - pad
        br = br + pad
    return ul, br, int(pad), scale_factor


def crop_canvas(img_hwc, center, scale, rot, res, size):
    """The zero-padded window the reference hands to its resamplers (pylib/HumanAug.py:144-159), scale_factor < 2."""
    ul, br, pad, sf = crop_geometry(center, scale, rot, res, size)
    assert sf == 1, 'the pre-shrink branch (scale * size / res >= 2) resamples before the window is cut'
    new_shape = [br[1] - ul[1], br[0] - ul[0]]
    if img_hwc.ndim > 2:
        new_shape += [img_hwc.shape[2]]
    new_img = np.zeros(new_shape)
    ht, wd = img_hwc.shape[0], img_hwc.shape[1]
    new_x = max(0, -ul[0]), min(br[0], wd) - ul[0]
    new_y = max(0, -ul[1]), min(br[1], ht) - ul[1]
    old_x = max(0, ul[0]), min(wd, br[0])
    old_y = max(0, ul[1]), min(ht, br[1])
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img_hwc[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    return new_img


def augment_sample(img_chw, center, scale, rot=0.0, flip=False, gain=(1.0, 1.0, 1.0), res=256, size=200):
    """C x H x W float image in [0, 1] -> C x res x res float32 network input.

    Order of the reference (data/mpii_for_mpii_22.py:127-141): flip (the CALLER also flips center[0] and the points, as the
    reference does), per-channel gain + clamp, crop.  The crop is one bilinear sample per output pixel: output pixel centre
    -> window coordinates by the resize (u = (ox + 0.5) * w / res - 0.5), rotation by `rot` degrees about the centre of the
    padded canvas (PIL's rotate: destination -> source with angle -rot), + ul -> source image; zero outside the image.
    With scale * size / res >= 2 the reference shrinks the image first: here the footprint is then averaged over k x k
    sub-samples, k = floor(scale * size / res)."""
    img = np.asarray(img_chw, dtype=np.float64)
    c, ht, wd = img.shape
    if flip:
        img = img[:, :, ::-1]
    img = np.clip(img * np.asarray(gain, dtype=np.float64)[:, None, None], 0.0, 1.0)
    sf_full = float(scale * size) / float(res)
    ul, br, pad, sf = crop_geometry(center, scale, rot, res, size)
    k = int(np.floor(sf_full)) if sf_full >= 2 else 1
    cw, chh = br[0] - ul[0], br[1] - ul[1]                # padded canvas size (in pre-shrunk pixels)
    win_w, win_h = cw - 2 * pad * (rot != 0), chh - 2 * pad * (rot != 0)
    phi = -np.deg2rad(rot)                                # PIL rotate(): destination -> source with angle = -radians(rot)
    cs, sn = np.cos(phi), np.sin(phi)
    ccx, ccy = cw / 2.0, chh / 2.0
    out = np.zeros((c, res, res), dtype=np.float64)
    sub = (np.arange(k) + 0.5) / k - 0.5                  # sub-sample offsets in output-pixel units
    for oy in range(res):
        for ox in range(res):
            acc = np.zeros(c)
            for sy in sub:
                for sx in sub:
                    u = (ox + sx + 0.5) * win_w / res - 0.5 + (pad if rot != 0 else 0)
                    v = (oy + sy + 0.5) * win_h / res - 0.5 + (pad if rot != 0 else 0)
                    if rot != 0:                          # pixel-centre coordinates -> rotate about the canvas centre
                        px, py = u + 0.5 - ccx, v + 0.5 - ccy
                        u = cs * px + sn * py + ccx - 0.5
                        v = -sn * px + cs * py + ccy - 0.5
                    # canvas pixel centres -> pixel centres of the (un-shrunk) source image
                    xs = (u + ul[0] + 0.5) * sf - 0.5
                    ys = (v + ul[1] + 0.5) * sf - 0.5
                    x0, y0 = int(np.floor(xs)), int(np.floor(ys))
                    fx, fy = xs - x0, ys - y0
                    for (yy, wy) in ((y0, 1 - fy), (y0 + 1, fy)):
                        for (xx, wx) in ((x0, 1 - fx), (x0 + 1, fx)):
                            if 0 <= yy < ht and 0 <= xx < wd:
                                acc += wy * wx * img[:, yy, xx]
            out[:, oy, ox] = acc / (k * k)
    return out.astype(np.float32)


# ---- the loader's per-sample recipe (data/mpii_for_mpii_22.py:86-145), one sample at a time on the CPU -------------------
def sample_from_bounded_gaussian(x, rng):                               # data/mpii_for_mpii_22.py:12-13
    return max(-2 * x, min(2 * x, rng.randn() * x))


def getitem_train(img_chw, joint_self, objpos, scale_provided, rng, inp_res=256, out_res=64, scale_factor=0.25, rot_factor=30,
                  std_size=200, is_train=True):
    """Restates MPII.__getitem__ for dataset 'MPII' (data/mpii_for_mpii_22.py:86-145) with the draws taken from `rng` (a
    numpy RandomState standing in for the module-level np.random) in the reference's order.  Returns
    (inp 3 x res x res, pts_aug K x 2 int, c, s, r, pts)."""
    pts = np.asarray(joint_self, dtype=np.float64)[:, 0:2].copy()      # :93-95
    c = np.array(objpos, dtype=np.float64)                              # :98
    s = float(scale_provided)                                           # :100
    c[1] = c[1] + 15 * s                                                # :104
    s = s * 1.25                                                        # :105
    img = np.array(img_chw, dtype=np.float32, copy=True)
    r, flip, gain = 0, False, (1.0, 1.0, 1.0)
    if is_train:
        s = s * (2 ** sample_from_bounded_gaussian(scale_factor, rng))  # :122
        r = sample_from_bounded_gaussian(rot_factor, rng)               # :123
        if rng.uniform(0, 1, 1) <= 0.6:                                 # :124-125
            r = 0
        if rng.random_sample() <= 0.5:                                  # :128-131
            flip = True
            pts = shufflelr(pts, width=img.shape[2])
            c[0] = img.shape[2] - c[0]
        gain = (rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4))      # :134-136
    inp = augment_sample(img, c, s, r, flip, gain, res=inp_res, size=std_size)            # :139-141
    pts_aug = transform_pts(pts, c, s, r, out_res, std_size)            # :143-144
    return inp, pts_aug, c, s, r, pts
