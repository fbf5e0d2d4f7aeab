"""Fused train step and decode on the HIP path.

`FusedTrainer.step(img, heatmap)` is the body of the reference's training loop
(cu-net.py:171-183: forward, sum of per-head MSE, zero_grad/backward, RMSprop step) executed
without leaving the HIP library: loss and d(loss)/d(out) come from the MSE kernel, gradients land
in the flat arena, optional RCCL all-reduce runs per gradient bucket while backward is still in
flight, and one fused RMSprop kernel updates every parameter.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import CUNetError, check, lib
from .module import CUNet, _ptr, _stream_ptr


class FusedTrainer:
    def __init__(self, net: CUNet, lr: float = 2.5e-4, alpha: float = 0.99, eps: float = 1e-8,
                 process_group=None, overlap: bool = True, quan_op=None, bf16: bool = False, bf16_grads: bool = False,
                 quan_input_bits: int = 0, popcount: bool = False):
        """RMSprop hyper-parameters default to cu-net.py:60-61. `process_group`: a torch.distributed
        group (backend nccl == RCCL) for data parallelism, or None."""
        if not isinstance(net, CUNet):
            raise CUNetError('FusedTrainer needs a cu_net_amd.CUNet')
        self.net = net
        self.lr, self.alpha, self.eps = float(lr), float(alpha), float(eps)
        self.square_avg = None
        self.bf16 = bool(bf16) or bool(bf16_grads)   # bf16 activation storage + bf16 MFMA forward; weights and the optimiser stay fp32
        self.bf16_grads = bool(bf16_grads)           # ... and the gradient tensors of backward (dY, dz, dX) stored as bf16 too
        self.steps_done = 0           # optimiser steps taken (the `step` entry of torch's RMSprop state)
        self.quan_op = quan_op        # cu_net_amd.quant.QuanOp / BinOp: quantised training (cu-net-prev-version-wig.py:163-190)
        # QuanInput2d in front of the 3x3 and head convs (the reference's quantised model places it there); with `popcount`
        # the forward of those convs whose weights QuanOp keeps ternary (bits_w 1 / 2) runs on the AND-popcount kernel
        if popcount and not quan_input_bits:
            quan_input_bits = getattr(quan_op, 'bits_i', 8) or 8
        if quan_input_bits:
            if self.bf16:
                raise CUNetError('the quantised-input mode is fp32 only')
            tern = ()
            if popcount:
                if quan_op is None or quan_op.bits_w not in (1, 2) or quan_op.keep_scale:
                    raise CUNetError('popcount=True needs a QuanOp with bits_w 1 or 2 (ternary weights during forward / backward)')
                tern = tuple(quan_op.target_names)
            net.set_quant_input(quan_input_bits, tern)
        self.quan_input_bits = int(quan_input_bits)
        self.popcount = bool(popcount)
        self.pg = process_group
        self.world = 1
        self.overlap = overlap
        from .parallel import BucketAllReducer
        self.reducer = BucketAllReducer(net._buckets, process_group, overlap)
        self.world = self.reducer.world

    # ---- data parallel plumbing (cu-net.py:59 DataParallel -> one process per GPU + RCCL) --------
    def broadcast_parameters(self, src: int = 0):
        """One-time replacement of DataParallel's per-iteration replicate."""
        from .parallel import broadcast_state
        broadcast_state([self.net._param_arena, self.net._buffer_arena, self.net._counter_arena], src, self.pg)

    def step(self, img: torch.Tensor, heatmap: torch.Tensor) -> torch.Tensor:
        """One optimisation step; returns the loss as a 0-dim device tensor (no host sync)."""
        net = self.net
        if not img.is_cuda or not heatmap.is_cuda:
            raise CUNetError('FusedTrainer.step needs GPU tensors (no CPU fallback)')
        if not net.training:
            raise CUNetError('call net.train() before FusedTrainer.step')
        net._check_aliasing()
        img = img.contiguous()
        heatmap = heatmap.contiguous()
        n, _, h, w = img.shape
        if tuple(heatmap.shape) != (n, net._hyper[3], h // 4, w // 4):
            raise CUNetError(f'heatmap must be {n} x {net._hyper[3]} x {h // 4} x {w // 4}')
        plan = net._get_plan(n, h, w, True, bf16=self.bf16)
        if self.quan_op is not None:
            self.quan_op.quantization()
        reducing = False
        try:
            loss = plan.stage_target(heatmap)     # MSE + d(loss)/d(out) in the head epilogues of the forward (cu-net.py:175-178)
            if self.bf16:
                plan.forward_bf16(img, 2 if self.bf16_grads else 1, want_outputs=False)
            else:
                plan.forward(img, True, want_outputs=False)
            if self.pg is None:
                plan.backward(None)
            else:
                self.reducer.begin_step()
                reducing = True
                plan.backward(None, on_bucket=lambda b: self.reducer.reduce_bucket(net._grad_arena, b, plan.side_stream_join))
        except BaseException:
            # a failed forward / backward (e.g. a bucket callback that could not issue its collective) must not leave the
            # QUANTISED weights in the arena -- the latent fp32 weights would be lost -- nor the communication stream unjoined
            if self.quan_op is not None:
                self.quan_op.restore()
            if reducing:
                self.reducer.finish(net._grad_arena)
            raise
        if reducing:
            self.reducer.finish(net._grad_arena)
        gscale = 1.0 / self.world
        if self.quan_op is not None:
            self.quan_op.restore()
            if self.world > 1:                      # the reference rewrites the already averaged gradient
                net._grad_arena.mul_(gscale)
                gscale = 1.0
            self.quan_op.updateQuanGradWeight()
        if self.square_avg is None or self.square_avg.device != net._param_arena.device:
            self.square_avg = torch.zeros_like(net._param_arena)
        check(lib().cunet_rmsprop_step(_ptr(net._param_arena), _ptr(net._grad_arena), _ptr(self.square_avg),
                                       net._n_params, self.lr, self.alpha, self.eps, gscale,
                                       _stream_ptr(img.device)), 'cunet_rmsprop_step')
        self.steps_done += 1
        return loss

    def last_outputs(self, img_shape):
        """Heat maps of the last step (NCHW copies), e.g. for the per-step accuracy of cu-net.py:191."""
        n, _, h, w = img_shape
        plan = self.net._get_plan(n, h, w, True, bf16=self.bf16)
        d = plan.handle.describe()
        heads = sorted((nd['head'], d['tensors'][nd['out']]['name']) for nd in d['nodes'] if nd.get('head', -1) >= 0)
        return [plan.debug_tensor(name) for _, name in heads]


def get_preds(scores: torch.Tensor) -> torch.Tensor:
    """pylib/Evaluation.py:6-23 on the GPU: N x K x H x W heat maps -> N x K x 2 float (x, y), 1-based,
    zeros where the map's maximum is <= 0. Bit-exact with the reference (ties -> lowest index)."""
    assert scores.dim() == 4, 'Score maps should be 4-dim'
    if not scores.is_cuda:
        raise CUNetError('get_preds: GPU tensor required (the CPU oracle is oracle/decode_ref.py)')
    s = scores.contiguous().float()
    n, k, h, w = s.shape
    preds = torch.empty((n, k, 2), dtype=torch.float32, device=s.device)
    check(lib().cunet_get_preds(_ptr(s), _ptr(preds), n, k, h, w, _stream_ptr(s.device)), 'cunet_get_preds')
    return preds


def _inverse_crop_transforms(center: torch.Tensor, scale: torch.Tensor, rot: torch.Tensor, res: int, size: int = 200):
    """Rows 0 and 1 of inv(GetTransform(center, scale, rot, res, size)) per image, N x 6 float64 (pylib/Evaluation.py:152-178,
    inverted as TransformPts :183-184 does).  Host-side numpy with the REFERENCE'S dtypes: it receives float32 0-d arrays
    (`center[i].numpy()` ...), so size * scale, res / h, the translation terms and the angle in radians -- hence sin / cos -- are
    float32 results stored into a float64 matrix; the 3x3 products and the inverse are float64 (np.dot / np.linalg.inv)."""
    import numpy as np
    c = center.detach().cpu().float().numpy()
    s = scale.detach().cpu().float().numpy()
    r = rot.detach().cpu().float().numpy()
    out = np.zeros((c.shape[0], 6))
    half = res / 2
    to_centre = np.array([[1.0, 0.0, -half], [0.0, 1.0, -half], [0.0, 0.0, 1.0]])
    back = np.array([[1.0, 0.0, half], [0.0, 1.0, half], [0.0, 0.0, 1.0]])
    for i in range(c.shape[0]):
        # (every float32 step spelled out: under NumPy >= 2 (NEP 50) a python scalar next to a float32 0-d array stays float32 -- what the
        # G8 / G8r fixtures and the device path of the rot == 0 case pin; NumPy 1.x would promote the same expressions to float64)
        f32 = np.float32
        h = f32(size) * s[i]
        zoom = f32(res) / h
        t = np.array([[zoom, 0.0, f32(res) * (-c[i, 0] / h + f32(.5))],
                      [0.0, zoom, f32(res) * (-c[i, 1] / h + f32(.5))],
                      [0.0, 0.0, 1.0]], dtype=np.float64)
        if r[i] != 0:
            rad = -r[i] * f32(np.pi) / f32(180)            # (to match the direction of the crop's rotation, :164)
            sn, cs = np.sin(rad), np.cos(rad)              # float32 in, float32 out
            turn = np.array([[cs, -sn, 0.0], [sn, cs, 0.0], [0.0, 0.0, 1.0]], dtype=np.float64)
            t = np.dot(back, np.dot(turn, np.dot(to_centre, t)))          # rotate about the centre of the crop (:171-177)
        out[i] = np.linalg.inv(t)[:2].reshape(6)
    return out


def final_preds(output: torch.Tensor, center: torch.Tensor, scale: torch.Tensor, res, rot=None) -> torch.Tensor:
    """pylib/Evaluation.py:108-132 on the GPU: N x K x H x W heat maps + per-image crop centre / scale (/ rotation in degrees) ->
    N x K x 2 original-image coordinates.  rot == 0 everywhere (the validation path, cu-net.py:272): the whole transform on the
    device (cunet_final_preds); any rot != 0: the 3x3 inverse per image on the host, applied on the device (cunet_final_preds_affine)."""
    if not output.is_cuda:
        raise CUNetError('final_preds: GPU tensor required (the CPU oracle is oracle/decode_ref.py)')
    s = output.contiguous().float()
    n, k, h, w = s.shape
    dev = s.device
    if rot is not None and bool(torch.as_tensor(rot).ne(0).any()):
        inv = torch.from_numpy(_inverse_crop_transforms(center, scale, torch.as_tensor(rot), int(res[0]))).to(dev)
        preds = torch.empty((n, k, 2), dtype=torch.float32, device=dev)
        check(lib().cunet_final_preds_affine(_ptr(s), _ptr(inv), _ptr(preds), n, k, h, w, int(res[0]), int(res[1]),
                                             _stream_ptr(dev)), 'cunet_final_preds_affine')
        return preds
    c = center.contiguous().float().to(dev)
    sc = scale.contiguous().float().to(dev)
    preds = torch.empty((n, k, 2), dtype=torch.float32, device=dev)
    check(lib().cunet_final_preds(_ptr(s), _ptr(c), _ptr(sc), _ptr(preds), n, k, h, w, int(res[0]), int(res[1]),
                                  _stream_ptr(dev)), 'cunet_final_preds')
    return preds


def flip_merge(out1: torch.Tensor, out2: torch.Tensor, flip_indxs) -> torch.Tensor:
    """Flip test-time augmentation of the validation loop (cu-net.py:240-249) on the GPU:
    `(out1 + shuffle_channels_for_horizontal_flipping(flip_channels(out2), flip_indxs)) / 2`
    (pylib/HumanAug.py:177-208), where out2 = net(img flipped along the width).  N x K x H x W."""
    if not (out1.is_cuda and out2.is_cuda):
        raise CUNetError('flip_merge: GPU tensors required (the CPU oracle is oracle/decode_ref.py)')
    a = out1.contiguous().float()
    b = out2.contiguous().float()
    if a.shape != b.shape or a.dim() != 4:
        raise CUNetError('flip_merge: two N x K x H x W tensors of equal shape expected')
    n, k, h, w = a.shape
    perm = list(range(k))
    for p in flip_indxs:                              # the reference swaps the pairs one after the other
        i1, i2 = int(p[0]), int(p[1])
        perm[i1], perm[i2] = perm[i2], perm[i1]
    pd = torch.tensor(perm, dtype=torch.int32, device=a.device)
    out = torch.empty_like(a)
    check(lib().cunet_flip_merge(_ptr(a), _ptr(b), _ptr(pd), _ptr(out), n, k, h, w, _stream_ptr(a.device)), 'cunet_flip_merge')
    return out


def _calc_dists(preds: torch.Tensor, target: torch.Tensor, normalize: torch.Tensor, use_zero: bool) -> torch.Tensor:
    """pylib/Evaluation.py:24-39 without the N x K python loop: K x N, -1 where the ground truth is missing.
    Coordinates are integer valued, so the fp32 distance is the same number on either device."""
    boundary = 0 if use_zero else 1
    d = (preds.float() - target.float()).pow(2).sum(-1).sqrt() / normalize.float().view(-1, 1)
    ok = (target[..., 0] > boundary) & (target[..., 1] > boundary)
    return torch.where(ok, d, torch.full_like(d, -1.0)).t().contiguous()


def _acc_from_dists(dists: torch.Tensor, idxs, thr: float) -> torch.Tensor:
    """pylib/Evaluation.py:41-53,69-83 on a K x N distance matrix; returns the reference's CPU vector
    [mean over the joints that have ground truth, per-joint accuracies (or -1)]."""
    sel = dists[torch.as_tensor(list(idxs), dtype=torch.long, device=dists.device)]
    valid = sel.ne(-1)
    nvalid = valid.sum(1)
    hit = sel.le(thr).eq(valid).sum(1).float()
    per = torch.where(nvalid > 0, hit / nvalid.clamp(min=1).float(), torch.full_like(hit, -1.0))
    acc = torch.zeros(len(idxs) + 1)
    per_cpu = per.cpu()
    acc[1:] = per_cpu
    has = per_cpu >= 0
    if bool(has.any()):
        avg = torch.zeros(())
        for v in per_cpu[has]:                        # same left-to-right fp32 sum as the reference's loop
            avg = avg + v
        acc[0] = avg / int(has.sum())
    return acc


def accuracy(output: torch.Tensor, target: torch.Tensor, idxs, thr: float = 0.5) -> torch.Tensor:
    """pylib/Evaluation.py:55-83 (PCK on heat-map resolution) with the arg-max decode and distances on the GPU."""
    preds, gts = get_preds(output), get_preds(target)
    norm = torch.ones(preds.size(0), device=preds.device) * output.size(3) / 10
    return _acc_from_dists(_calc_dists(preds, gts, norm, False), idxs, thr)


def accuracy_origin_res(output: torch.Tensor, center, scale, res, grnd_pts, normalizers, rot=None) -> torch.Tensor:
    """pylib/Evaluation.py:86-106 (PCKh in original-image coordinates; the reference's fixed MPII joint list)."""
    idxs = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15]
    pred_pts = final_preds(output, center, scale, res, rot)
    dev = pred_pts.device
    dists = _calc_dists(pred_pts, torch.as_tensor(grnd_pts).to(dev), torch.as_tensor(normalizers).to(dev), True)
    return _acc_from_dists(dists, idxs, 0.5)


def pts2heatmap(pts: torch.Tensor, heatmap_shape, sigma: float = 1) -> torch.Tensor:
    """Training targets on the GPU (pylib/HumanPts.py:35-76): `pts` is N x K x 2 (or K x 2) of (x, y) heat-map
    coordinates, the result N x K x H x W fp32 -- what `torch.from_numpy(pts2heatmap(...)[0]).float()` gives per sample.
    The (2*ceil(3 sigma)+1)^2 Gaussian patch is tabulated on the host in float64 exactly as the reference does."""
    import numpy as np
    if not pts.is_cuda:
        raise CUNetError('pts2heatmap: GPU tensor required (the CPU oracle is oracle/decode_ref.py)')
    squeeze = pts.dim() == 2
    p = (pts.unsqueeze(0) if squeeze else pts).contiguous().double()
    n, k, _ = p.shape
    h, w = int(heatmap_shape[0]), int(heatmap_shape[1])
    tmp = np.ceil(3 * sigma)
    size = 2 * tmp + 1
    x = np.arange(0, size, 1, float)
    g = np.exp(-((x - size // 2) ** 2 + (x[:, np.newaxis] - size // 2) ** 2) / (tmp ** 2))
    patch = torch.from_numpy(g.astype(np.float32)).to(p.device).contiguous()
    out = torch.empty((n, k, h, w), dtype=torch.float32, device=p.device)
    check(lib().cunet_render_targets(_ptr(p), _ptr(patch), int(tmp), _ptr(out), n * k, h, w, _stream_ptr(p.device)),
          'cunet_render_targets')
    return out[0] if squeeze else out
