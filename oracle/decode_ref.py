"""CPU oracle for the heat-map -> landmark decode -- TEST INFRASTRUCTURE ONLY.

Restates pylib/Evaluation.py:6-23 (`get_preds`), :108-132 (`final_preds`) and :152-187
(`GetTransform` / `TransformPts`) with the same torch / numpy calls in the same order, so integer
outputs are bit-identical to the reference on the same host.  Pinned by tools/gen_golden.py against the
reference's own module (tests/golden/G8_decode.npz)."""
from __future__ import annotations

import math

import numpy as np
import torch


def get_preds(scores: torch.Tensor) -> torch.Tensor:
    """pylib/Evaluation.py:6-23."""
    assert scores.dim() == 4, 'Score maps should be 4-dim'
    maxval, idx = torch.max(scores.view(scores.size(0), scores.size(1), -1), 2)
    maxval = maxval.view(scores.size(0), scores.size(1), 1)
    idx = idx.view(scores.size(0), scores.size(1), 1) + 1
    preds = idx.repeat(1, 1, 2).float()
    preds[:, :, 0] = (preds[:, :, 0] - 1) % scores.size(3) + 1
    preds[:, :, 1] = torch.floor((preds[:, :, 1] - 1) / scores.size(2)) + 1
    pred_mask = maxval.gt(0).repeat(1, 1, 2).float()
    preds *= pred_mask
    return preds


def get_transform(center, scale, rot, res, size):
    """pylib/Evaluation.py:152-178."""
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


def transform_pts(pts, center, scale, rot, res, size, invert=0):
    """pylib/Evaluation.py:180-187."""
    nlmk, dim = pts.shape
    t = get_transform(center, scale, rot, res, size)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.concatenate((pts - 1, np.ones((nlmk, 1))), axis=1).T
    new_pt = np.dot(t, new_pt)
    new_pt = new_pt[0:2, :].T
    return new_pt.astype(int) + 1


def final_preds(output: torch.Tensor, center: torch.Tensor, scale: torch.Tensor, res, rot: torch.Tensor) -> torch.Tensor:
    """pylib/Evaluation.py:108-132 (+ transform_preds :134-150)."""
    coords = get_preds(output)
    for n in range(coords.size(0)):
        for p in range(coords.size(1)):
            hm = output[n][p]
            px = int(math.floor(coords[n][p][0]))
            py = int(math.floor(coords[n][p][1]))
            if 1 < px < res[0] and 1 < py < res[1]:
                diff = torch.Tensor([hm[py - 1][px] - hm[py - 1][px - 2], hm[py][px - 1] - hm[py - 2][px - 1]])
                coords[n][p] += diff.sign() * .25
    coords += 0.5
    preds = coords.clone()
    for i in range(coords.size(0)):
        # (transform_preds :134-150 hands GetTransform float32 0-d arrays -- `rot.numpy()` too: the angle, its sine and cosine are float32)
        pts = transform_pts(coords[i].numpy(), center[i].numpy(), scale[i].numpy(), rot[i].float().numpy(), res[0], size=200, invert=1)
        preds[i] = torch.from_numpy(pts)
    return preds


# ---- flip test-time augmentation and PCK accuracy (validation loop, cu-net.py:240-258) -----------------
def flip_merge(out1: torch.Tensor, out2: torch.Tensor, flip_indxs) -> torch.Tensor:
    """(output1 + shuffle(flip(output2))) / 2 -- cu-net.py:247-249 with pylib/HumanAug.py:196-208
    (flip_channels: reverse the width axis) and :177-194 (shuffle_channels_for_horizontal_flipping:
    swap the channel pairs of `flip_indxs`, in order)."""
    m = out2.flip(3).clone()
    for idx1, idx2 in [tuple(int(v) for v in p) for p in flip_indxs]:
        tmp = m[:, idx1].clone()
        m[:, idx1] = m[:, idx2]
        m[:, idx2] = tmp
    return (out1 + m) / 2


def calc_dists(preds: torch.Tensor, target: torch.Tensor, normalize: torch.Tensor, use_zero: bool = False) -> torch.Tensor:
    """pylib/Evaluation.py:24-39: K x N distances, -1 where the ground truth is missing (coordinate <= boundary)."""
    preds, target, normalize = preds.float(), target.float(), normalize.float()
    dists = torch.zeros(preds.size(1), preds.size(0))
    boundary = 0 if use_zero else 1
    for n in range(preds.size(0)):
        for c in range(preds.size(1)):
            if target[n, c, 0] > boundary and target[n, c, 1] > boundary:
                dists[c, n] = torch.dist(preds[n, c, :], target[n, c, :]) / normalize[n]
            else:
                dists[c, n] = -1
    return dists


def dist_acc(dists: torch.Tensor, thr: float = 0.5):
    """pylib/Evaluation.py:41-53 (NB `le(thr).eq(ne(-1))`: a -1 entry counts as a hit of the numerator's
    equality test only when both sides are False, i.e. never -- but a valid entry ABOVE thr does not, and
    a -1 entry is `le(thr)` yet `ne(-1)` False, so it is not counted either)."""
    if dists.ne(-1).sum() > 0:
        return dists.le(thr).eq(dists.ne(-1)).sum().float() / dists.ne(-1).sum().float()
    return -1


def accuracy(output: torch.Tensor, target: torch.Tensor, idxs, thr: float = 0.5) -> torch.Tensor:
    """pylib/Evaluation.py:55-83."""
    preds, gts = get_preds(output), get_preds(target)
    norm = torch.ones(preds.size(0)) * output.size(3) / 10
    dists = calc_dists(preds, gts, norm)
    acc = torch.zeros(len(idxs) + 1)
    avg_acc, cnt = 0, 0
    for i in range(len(idxs)):
        acc[i + 1] = dist_acc(dists[idxs[i]], thr)
        if acc[i + 1] >= 0:
            avg_acc = avg_acc + acc[i + 1]
            cnt += 1
    if cnt != 0:
        acc[0] = avg_acc / cnt
    return acc


# ---- target synthesis (training labels) -----------------------------------------------------------------
def draw_gaussian(img, pt, sigma):
    """pylib/HumanPts.py:49-76: paste an un-normalised (peak 1) Gaussian patch of half-width ceil(3 sigma) whose
    centre is pixel (int(pt[0]), int(pt[1])) (x, y); the part outside the map is cropped; float64 numpy."""
    import numpy as np
    tmp_size = np.ceil(3 * sigma)
    ul = [int(pt[0] - tmp_size), int(pt[1] - tmp_size)]
    br = [int(pt[0] + tmp_size), int(pt[1] + tmp_size)]
    if ul[0] >= img.shape[1] or ul[1] >= img.shape[0] or br[0] < 0 or br[1] < 0:
        return img
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, float)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    g = np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (tmp_size ** 2))
    g_x = max(0, -ul[0]), min(br[0] + 1, img.shape[1]) - max(0, ul[0]) + max(0, -ul[0])
    g_y = max(0, -ul[1]), min(br[1] + 1, img.shape[0]) - max(0, ul[1]) + max(0, -ul[1])
    img_x = max(0, ul[0]), min(br[0] + 1, img.shape[1])
    img_y = max(0, ul[1]), min(br[1] + 1, img.shape[0])
    img[img_y[0]:img_y[1], img_x[0]:img_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    return img


def pts2heatmap(pts, heatmap_shape, sigma=1):
    """pylib/HumanPts.py:35-47: K x H x W float64 maps and the K x 2 points that were drawn (x <= 0 or y <= 0: skipped)."""
    import numpy as np
    heatmap = np.zeros((pts.shape[0], heatmap_shape[0], heatmap_shape[1]))
    valid_pts = np.zeros((pts.shape))
    for i in range(0, pts.shape[0]):
        if pts[i][0] <= 0 or pts[i][1] <= 0:
            continue
        heatmap[i] = draw_gaussian(heatmap[i], pts[i], sigma)
        valid_pts[i] = pts[i]
    return heatmap, valid_pts
