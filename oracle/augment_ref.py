"""CPU oracle for the training-sample preparation (image half of SURVEY 8f rank 4) -- TEST INFRASTRUCTURE ONLY.

Restates, in numpy, the per-sample work of the reference's data loader between "decoded image" and "network input":
    data/mpii_for_mpii_22.py:120-145   scale / rotation jitter, horizontal flip, per-channel colour gain, crop, target points
    pylib/HumanAug.py:10-57            GetTransform / TransformSinglePts / TransformPts
    pylib/HumanAug.py:115-172          crop (window of the source image on a zero canvas, rotation, resize to res x res)
    pylib/HumanAug.py:234-271          shufflelr / fliplr

Pinning (tools/gen_golden.py --only augment): G15_augment.npz holds the transform functions, shufflelr / fliplr and the WINDOW
EXTRACTION of crop (ul, br, pad, the zero-padded canvas) checked bit-for-bit against the reference's own functions compiled from
pylib/HumanAug.py's AST; G16_crop.npz holds whole crop() outputs of the EXECUTED reference function with its two removed
resamplers (scipy.misc.imresize / imrotate) rebuilt over the installed PIL exactly as scipy <= 1.2 wrapped it (byte-scale ->
Image.resize / Image.rotate(BILINEAR) -> uint8 array), and `crop` below -- a restatement of the PIL algorithms, no PIL import --
equals it bit for bit.  What "the reference" is for these pixels is therefore: reference code + scipy 1.x wrapper semantics + the
PIL of this image (12.2); Pillow's 8-bit resize / rotate arithmetic has been stable since 4.x but is not versioned against here.
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


# ---- the reference's resamplers, restated ---------------------------------------------------------------------------------
# crop() ends in scipy.misc.imresize and, for rot != 0, scipy.misc.imrotate (pylib/HumanAug.py:128,167,173).  Both were thin
# wrappers (scipy <= 1.2, scipy/misc/pilutil.py): toimage(arr) -- which BYTE-SCALES a float array with a data-dependent contrast
# stretch, (x - min) * 255 / (max - min) + 0.5 truncated to uint8 over the WHOLE array -- then PIL's Image.resize / Image.rotate
# with resample = BILINEAR, then back to a uint8 array.  scipy.misc is gone, PIL is not: tools/gen_golden.py rebuilds the two
# wrappers over the installed PIL, EXECUTES the reference's crop() with them and pins the functions below bit-for-bit (G16).
# The PIL algorithms restated here (Pillow src/libImaging/Resample.c, Geometry.c):
#   resize BILINEAR on 8-bit images: separable triangle filter, support = max(1, in/out) input pixels around the output pixel's
#     centre (an antialiasing filter when shrinking), weights normalised, converted to 22-bit fixed point, horizontal pass then
#     vertical pass with a uint8 intermediate image, each output = clip8((2^21 + sum coeff * pixel) >> 22);
#   rotate BILINEAR: inverse affine map about the image centre (matrix entries rounded to 15 decimals), 2 x 2 bilinear
#     interpolation in double with edge clamping, TRUNCATED to uint8, zero where the source position is outside the image.
PIL_PRECISION_BITS = 32 - 8 - 2


def bytescale(data):
    """scipy.misc.bytescale(data) with its defaults (cmin = data.min(), cmax = data.max(), low = 0, high = 255)."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(255) / cscale
    bytedata = (data - cmin) * scale + 0
    return (bytedata.clip(0, 255) + 0.5).astype(np.uint8)


def _pil_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter over the whole input range."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    bounds, coeffs = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds.append((xmin, xmax))
        coeffs.append(np.array([int(0.5 + v * (1 << PIL_PRECISION_BITS)) for v in w], dtype=np.int64))
    return bounds, coeffs


def _pil_resample_axis1(img, out_size):
    """One pass along axis 1 of an H x W x C uint8 image."""
    h, w, c = img.shape
    bounds, coeffs = _pil_coeffs(w, out_size)
    out = np.zeros((h, out_size, c), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        ss = (1 << (PIL_PRECISION_BITS - 1)) + (src[:, xmin:xmin + n, :] * coeffs[xx][None, :, None]).sum(1)
        out[:, xx, :] = np.clip(ss >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear(img_u8, out_w, out_h):
    """Image.resize((out_w, out_h), BILINEAR) of an H x W x C uint8 image (identity size: a copy)."""
    img = np.asarray(img_u8)
    if out_w != img.shape[1]:
        img = _pil_resample_axis1(img, out_w)
    if out_h != img.shape[0]:
        img = _pil_resample_axis1(img.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def pil_rotate_matrix(w, h, angle_deg):
    """The inverse affine map of Image.rotate(angle) (expand = False, centre = image centre): destination pixel centre
    -> source position, [a, b, c, d, e, f] with x_src = a x + b y + c, y_src = d x + e y + f."""
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


def pil_rotate_bilinear(img_u8, angle_deg):
    """Image.rotate(angle, resample=BILINEAR) of an H x W x C uint8 image.  Multiples of 90 degrees: PIL takes its transpose
    shortcuts there (180 always, 90 / 270 on square images); with the matrix entries rounded to 15 decimals (exactly 0 / +-1)
    the affine path below lands on pixel centres and gives the same bytes (tests/test_oracle_aux.py checks that against PIL)."""
    img = np.asarray(img_u8)
    h, w = img.shape[:2]
    if angle_deg % 360.0 == 0:
        return img.copy()
    m = pil_rotate_matrix(w, h, angle_deg)
    ys, xs = np.mgrid[0:h, 0:w]
    xin = m[0] * (xs + 0.5) + m[1] * (ys + 0.5) + m[2]
    yin = m[3] * (xs + 0.5) + m[4] * (ys + 0.5) + m[5]
    inside = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    xi, yi = xin - 0.5, yin - 0.5
    x, y = np.floor(xi).astype(np.int64), np.floor(yi).astype(np.int64)
    dx, dy = xi - x, yi - y
    x0, x1 = np.clip(x, 0, w - 1), np.clip(x + 1, 0, w - 1)
    r0 = np.clip(y, 0, h - 1)
    has2 = (y + 1 >= 0) & (y + 1 < h)
    r1 = np.clip(y + 1, 0, h - 1)
    f = img.astype(np.float64)
    out = np.zeros_like(img)
    for c in range(img.shape[2]):
        ch = f[:, :, c]
        a, b = ch[r0, x0], ch[r0, x1]
        v1 = a + (b - a) * dx
        a2, b2 = ch[r1, x0], ch[r1, x1]
        v2 = np.where(has2, a2 + (b2 - a2) * dx, v1)
        v = v1 + (v2 - v1) * dy
        out[:, :, c] = np.where(inside, v, 0.0).astype(np.uint8)          # (UINT8) v: truncation
    return out


def imresize(arr, size):
    """scipy.misc.imresize(arr, size, interp='bilinear') for H x W x 3 arrays; size: float fraction or (rows, cols)."""
    u8 = bytescale(arr)
    h, w = u8.shape[:2]
    if isinstance(size, float):
        out_w, out_h = int(w * size), int(h * size)
    else:
        out_h, out_w = size
    return pil_resize_bilinear(u8, out_w, out_h)


def imrotate(arr, angle):
    """scipy.misc.imrotate(arr, angle, interp='bilinear') for H x W x 3 arrays."""
    return pil_rotate_bilinear(bytescale(arr), angle)


def crop(img_hwc, center, scale, rot, res, size):
    """pylib/HumanAug.py:115-172 with the two resamplers above: H x W x 3 float image -> res x res x 3 uint8."""
    img = np.asarray(img_hwc)
    scale_factor = float(scale * size) / float(res)
    if scale_factor < 2:
        scale_factor = 1
    else:
        new_img_size = np.floor(max(img.shape[0], img.shape[1]) / scale_factor)
        if new_img_size < 2:
            return img
        img = imresize(img, 1 / scale_factor)
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
    new_img = np.zeros([br[1] - ul[1], br[0] - ul[0], img.shape[2]])
    ht, wd = img.shape[0], img.shape[1]
    new_x = max(0, -ul[0]), min(br[0], wd) - ul[0]
    new_y = max(0, -ul[1]), min(br[1], ht) - ul[1]
    old_x = max(0, ul[0]), min(wd, br[0])
    old_y = max(0, ul[1]), min(ht, br[1])
    new_img[new_y[0]:new_y[1], new_x[0]:new_x[1]] = img[old_y[0]:old_y[1], old_x[0]:old_x[1]]
    if not rot == 0:
        new_img = imrotate(new_img, rot)
        new_img = new_img[pad:-pad, pad:-pad]
    return imresize(new_img, (res, res))


def augment_sample(img_chw, center, scale, rot=0.0, flip=False, gain=(1.0, 1.0, 1.0), res=256, size=200):
    """C x H x W float32 image in [0, 1] -> C x res x res float32 network input, the reference's way
    (data/mpii_for_mpii_22.py:127-141): flip (the CALLER also flips center[0] and the points), per-channel gain + clamp in
    float32 (`img[c].mul_(gain).clamp_(0, 1)` on the torch tensor), crop() through the 8-bit resamplers, and
    utils/imutils.py:31-36 `im_to_torch`: uint8 -> float32 / 255 (when the crop's maximum exceeds 1, i.e. always but for an
    all-black / all-{0,1} crop, which the reference leaves unscaled and so does this)."""
    img = np.array(img_chw, dtype=np.float32, copy=True)
    if flip:
        img = img[:, :, ::-1]
    g = np.asarray(gain, dtype=np.float64).astype(np.float32)
    img = np.clip(img * g[:, None, None], np.float32(0), np.float32(1)).astype(np.float32)
    out = crop(np.transpose(img, (1, 2, 0)), center, scale, rot, res, size)
    out = np.transpose(out, (2, 0, 1)).astype(np.float32)
    if out.max() > 1:
        out = out / np.float32(255)
    return out


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
