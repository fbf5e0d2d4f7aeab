"""GPU (-m gpu): training-sample preparation on the device (cunet_augment_batch) -- BIT-EXACT against
  * tests/golden/G16_crop.npz: outputs of the reference's crop() (pylib/HumanAug.py:115-172) EXECUTED with its removed
    resamplers scipy.misc.imresize / imrotate rebuilt over PIL (tools/gen_golden.py crop_parity), and
  * oracle/augment_ref.py, the numpy restatement pinned to those outputs byte for byte,
on samples of different sizes in one call: flip, colour gain with clamp, windows hanging over the image border, rotation, the
pre-shrink branch (scale * 200 / res >= 2), the byte-scale contrast stretch of dim images.  The uint8 round trips of the
reference are reproduced, not approximated: tolerance 0."""
import os

import numpy as np
import pytest
import torch

import cu_net_amd
from oracle import augment_ref as A
from tests._golden import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_crop_matches_the_executed_reference_fixture():
    z = np.load(os.path.join(GOLDEN_DIR, 'G16_crop.npz'))
    imgs = [(z[f'img/{i}'].astype(np.float32) / np.float32(255)) for i in range(2)]
    cases = z['cases']
    dev_imgs = [torch.from_numpy(imgs[int(c[0])]).cuda() for c in cases]
    out = cu_net_amd.augment_batch(dev_imgs, cases[:, 1:3], cases[:, 3], cases[:, 4], cases[:, 5] != 0, cases[:, 6:9], res=256).cpu().numpy()
    for k in range(len(cases)):
        ref = np.transpose(z[f'out/{k}'], (2, 0, 1)).astype(np.float32) / np.float32(255)
        assert np.array_equal(out[k], ref), (k, int((out[k] != ref).sum()), float(np.abs(out[k] - ref).max()) * 255)


def test_augment_batch_matches_oracle():
    rng = np.random.RandomState(3)
    res = 48
    shapes = [(3, 60, 80), (3, 75, 64), (3, 120, 90), (3, 64, 64), (3, 50, 70), (3, 90, 110)]
    imgs = [(rng.randint(0, 256, size=s) / 255.0).astype(np.float32) for s in shapes]
    imgs[4] = (imgs[4] * 0.6 + 0.1).astype(np.float32)          # a dim image with a raised black level: byte-scale stretches it
    centers = np.array([[40.0, 30.0], [5.0, 70.0], [45.0, 60.0], [32.0, 32.0], [35.0, 25.0], [55.0, 45.0]])
    scales = np.array([0.3, 0.25, 0.62, 0.24, 0.12, 0.7])     # 0.62 / 0.7 * 200 / 48 >= 2: the pre-shrink branch
    rots = np.array([0.0, 25.0, 0.0, -40.0, 0.0, 13.0])
    flips = np.array([False, True, False, True, True, False])
    gains = rng.uniform(0.6, 1.4, size=(6, 3))
    out = cu_net_amd.augment_batch([torch.from_numpy(i).cuda() for i in imgs], centers, scales, rots, flips, gains, res=res).cpu().numpy()
    assert out.shape == (6, 3, res, res)
    for i in range(6):
        ref = A.augment_sample(imgs[i], centers[i], float(scales[i]), float(rots[i]), bool(flips[i]), gains[i], res=res)
        assert np.array_equal(out[i], ref), (i, int((out[i] != ref).sum()), float(np.abs(out[i] - ref).max()) * 255)
        assert out[i].max() <= 1.0 and out[i].min() >= 0.0
    assert out[0].max() > 0.3 and (out[1] == 0).any()          # sample 1 hangs over the border: zero canvas shows
    assert out[4].min() == 0.0 and out[4].max() > 0.95         # interior window of the dim image (values 0.1 .. 0.7): stretched to the full range


def test_augment_identity_window_is_the_byte_scaled_window():
    """A window of exactly res x res pixels without rotation or jitter: no resampling at all -- the output is the byte-scaled
    zero-padded window (scipy.misc.toimage's contrast stretch over the canvas), mirrored when flipped."""
    rng = np.random.RandomState(4)
    img = (rng.randint(0, 256, size=(3, 40, 50)) / 255.0).astype(np.float32)
    res = 32
    c = np.array([[25.0, 20.0], [25.0, 20.0]])
    s = np.array([res / 200.0, res / 200.0])
    out = cu_net_amd.augment_batch([torch.from_numpy(img).cuda()] * 2, c, s, flips=[False, True], res=res).cpu().numpy()
    for i, im in enumerate((img, img[:, :, ::-1])):
        canvas = A.crop_canvas(np.transpose(im.astype(np.float64), (1, 2, 0)), c[i], float(s[i]), 0, res, 200)
        want = np.transpose(A.bytescale(canvas), (2, 0, 1)).astype(np.float32) / np.float32(255)
        assert np.array_equal(out[i], want)


def test_right_angle_rotations_and_the_degenerate_box():
    """(1) rotations by multiples of 90 degrees: PIL's transpose shortcuts equal its affine path there (tests/test_oracle_aux.py), so
    the kernel's rotate stage covers them; (2) a person box too large to pre-shrink: HumanAug.crop returns the image unchanged
    (pylib/HumanAug.py:124-125) -- passed through (flip, gains, clamp) when the image already has the network's size, a CUNetError for
    that sample otherwise (the reference fails in its collate there)."""
    rng = np.random.RandomState(5)
    res = 48
    imgs = [(rng.randint(0, 256, size=(3, 64, 72)) / 255.0).astype(np.float32) for _ in range(4)]
    centers = np.array([[36.0, 30.0]] * 4)
    scales = np.array([0.3, 0.26, 0.3, 0.62])
    rots = np.array([90.0, 180.0, -90.0, 270.0])
    flips = np.array([False, True, False, True])
    gains = rng.uniform(0.6, 1.4, size=(4, 3))
    out = cu_net_amd.augment_batch([torch.from_numpy(i).cuda() for i in imgs], centers, scales, rots, flips, gains, res=res).cpu().numpy()
    for i in range(4):
        ref = A.augment_sample(imgs[i], centers[i], float(scales[i]), float(rots[i]), bool(flips[i]), gains[i], res=res)
        assert np.array_equal(out[i], ref), (i, int((out[i] != ref).sum()))
    # degenerate: scale * 200 / res >= 2 and floor(max(H, W) / sf) < 2
    sq = (rng.randint(0, 256, size=(3, res, res)) / 255.0).astype(np.float32)
    big = 0.5 * res * res / 200.0 + 1.0                              # sf = scale * 200 / res > res / 2  ->  floor(res / sf) < 2
    out = cu_net_amd.augment_batch([torch.from_numpy(sq).cuda(), torch.from_numpy(imgs[0]).cuda()], np.array([[24.0, 24.0], [36.0, 30.0]]),
                                   np.array([big, 0.3]), flips=[True, False], gains=np.array([[1.3, 0.7, 1.0], [1.0, 1.0, 1.0]]), res=res).cpu().numpy()
    ref0 = A.augment_sample(sq, np.array([24.0, 24.0]), big, 0.0, True, (1.3, 0.7, 1.0), res=res)
    assert ref0.shape == (3, res, res) and np.array_equal(out[0], ref0)
    assert np.array_equal(out[1], A.augment_sample(imgs[0], np.array([36.0, 30.0]), 0.3, 0.0, False, (1.0, 1.0, 1.0), res=res))
    with pytest.raises(cu_net_amd.CUNetError):
        cu_net_amd.augment_batch([torch.from_numpy(imgs[0]).cuda()], np.array([[36.0, 30.0]]), np.array([0.5 * 72 * res / 200.0 + 1.0]), res=res)


def test_prepare_batch_is_the_loaders_getitem():
    """cu_net_amd.prepare_batch == oracle restatement of MPII.__getitem__ (data/mpii_for_mpii_22.py:86-145) sample by sample under the
    same numpy seed: network input (bit for bit, the reference's 8-bit resamplers), target heat maps (bit-exact renderer), meta."""
    from oracle import decode_ref as D
    rng = np.random.RandomState(11)
    samples, raw = [], []
    for i in range(6):
        h, w = int(rng.randint(120, 200)), int(rng.randint(120, 260))
        img = (rng.randint(0, 256, size=(3, h, w)) / 255.0).astype(np.float32)
        objpos = [w * rng.uniform(0.4, 0.6), h * rng.uniform(0.4, 0.6)]
        scale = rng.uniform(0.3, 0.6)
        joints = np.concatenate([np.stack([objpos[0] + rng.uniform(-40, 40, 16), objpos[1] + rng.uniform(-50, 50, 16)], 1), np.ones((16, 1))], 1)
        raw.append((img, joints, objpos, scale))
        samples.append({'img': torch.from_numpy(img).cuda(), 'joint_self': joints, 'objpos': objpos, 'scale_provided': scale})
    inp, heat, meta = cu_net_amd.prepare_batch(samples, True, inp_res=64, out_res=16, rng=np.random.RandomState(77))
    r2 = np.random.RandomState(77)
    flips = 0
    for i, (img, joints, objpos, scale) in enumerate(raw):
        ref_inp, pts_aug, c, s, r, pts = A.getitem_train(img, joints, objpos, scale, r2, inp_res=64, out_res=16)
        assert np.array_equal(inp[i].cpu().numpy(), ref_inp), i
        assert np.array_equal(meta['pts_aug'][i], pts_aug) and np.array_equal(meta['pts'][i], pts)
        assert meta['scale'][i] == s and meta['rot'][i] == r and np.array_equal(meta['center'][i], c)
        ref_heat = D.pts2heatmap(pts_aug.astype(np.float64), (16, 16), 1)[0]
        assert np.array_equal(heat[i].cpu().numpy(), ref_heat.astype(np.float32)), i
        flips += int(not np.array_equal(pts, joints[:, :2]))
    assert 0 < flips < 6                                     # both branches were taken
    # validation samples: no jitter, no draws
    inp_v, heat_v, meta_v = cu_net_amd.prepare_batch(samples[:2], False, inp_res=64, out_res=16)
    ref_v = A.getitem_train(raw[0][0], raw[0][1], raw[0][2], raw[0][3], None, inp_res=64, out_res=16, is_train=False)
    assert np.array_equal(inp_v[0].cpu().numpy(), ref_v[0]) and meta_v['rot'][0] == 0
