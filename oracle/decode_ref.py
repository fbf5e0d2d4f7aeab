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
        pts = transform_pts(coords[i].numpy(), center[i].numpy(), scale[i].numpy(), float(rot[i]), res[0], size=200, invert=1)
        preds[i] = torch.from_numpy(pts)
    return preds
