"""Python-3 training / validation driver around the HIP hot path, with the reference's option names,
learning-rate schedule and checkpoint layout, so that runs started with the reference's `cu-net.py`
can be resumed here and vice versa.

Mirrors (behaviour, not code):
    options/base_options.py:13-37, options/train_options.py:7-37    command-line flags
    utils/util.py:8-46                                              TrainHistory
    utils/util.py:106-119                                           adjust_lr (x0.2 at epoch 101, x0.5 at 141 and 161)
    utils/checkpoint.py:13-31,40-67                                 checkpoint files {train_history, state_dict, optimizer}
    cu-net.py:59-61                                                 DataParallel + RMSprop(lr, alpha 0.99, eps 1e-8)
    cu-net.py:152-216, 219-258                                      train / validate loops

What is different on purpose: one process per GPU under `torch.distributed.run` instead of
DataParallel (checkpoints still carry the `module.` key prefix DataParallel produced, so the files are
interchangeable); forward, loss, backward and the optimiser step are one `FusedTrainer.step`; heat maps
never leave the GPU for decoding / accuracy.  The MPII loader of the reference (`data/mpii_for_mpii_22.py`)
depends on `scipy.misc` and on files that are not part of the repository: pass any iterable of
`(img, heatmap, ...)` batches as `train_loader` / `val_loader`; `SyntheticLoader` stands in for it.
"""
from __future__ import annotations

import argparse
import os
from collections import OrderedDict
from typing import Iterable, List, Optional

import torch

from .module import CUNet, create_cu_net
from .trainer import FusedTrainer, accuracy, flip_merge

# cu-net.py:34-35 (MPII left/right joint pairs) and :137 (joints scored during training)
JOINT_FLIP_INDEX = [[1, 4], [0, 5], [12, 13], [11, 14], [10, 15], [2, 3]]
TRAIN_ACC_IDX = [0, 1, 2, 3, 4, 5, 10, 11, 14, 15]


def _str2bool(v):       # the reference declares these flags with type=bool (any non-empty string is True there)
    return str(v).lower() not in ('', '0', 'false', 'no')


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='CU-Net training on the MI355X hot path (reference flag names)')
    # options/base_options.py
    p.add_argument('--data_dir', type=str, default='./dataset')
    p.add_argument('--exp_dir', type=str, default='./exp')
    p.add_argument('--exp_id', type=str, default='')
    p.add_argument('--gpu_id', type=str, default='0', help='ignored: one process per GPU, use torch.distributed.run')
    p.add_argument('--nThreads', type=int, default=4)
    p.add_argument('--is_train', type=_str2bool, default=True)
    p.add_argument('--dataset', type=str, default='mpii')
    # options/train_options.py
    p.add_argument('--layer_num', type=int, default=2)
    p.add_argument('--order', type=int, default=1)
    p.add_argument('--class_num', type=int, default=16)
    p.add_argument('--loss_num', type=int, default=16)
    p.add_argument('--lr', type=float, default=2.5e-4)
    p.add_argument('--bs', type=int, default=24, help='GLOBAL mini-batch (split over the ranks like DataParallel splits it)')
    p.add_argument('--load_checkpoint', type=_str2bool, default=False)
    p.add_argument('--adjust_lr', type=_str2bool, default=False, help='declared but never read by the reference (cu-net.py:125 '
                   'always applies the schedule); kept for command-line compatibility, ignored here too')
    p.add_argument('--no_lr_schedule', type=_str2bool, default=False, help='not in the reference: opt OUT of the x0.2 / x0.5 / x0.5 '
                   'decays at epochs 101 / 141 / 161')
    p.add_argument('--resume_prefix', type=str, default='')
    p.add_argument('--nEpochs', type=int, default=200)
    p.add_argument('--best_pckh', type=float, default=0.)
    p.add_argument('--print_freq', type=int, default=10)
    p.add_argument('--display_freq', type=int, default=10)
    p.add_argument('--bits_w', type=int, default=0, help='>0: quantised training (utils/quantize.py QuanOp); 0 = full precision')
    p.add_argument('--bits_i', type=int, default=8)
    p.add_argument('--bits_g', type=int, default=8)
    # not in the reference: there is no dataset in this repository
    p.add_argument('--bf16', type=_str2bool, default=False, help='bf16 activation storage + bf16 MFMA forward (gradients, weights, RMSprop fp32)')
    p.add_argument('--bf16_grads', type=_str2bool, default=False, help='with --bf16: gradient tensors of backward stored as bf16 too')
    p.add_argument('--synthetic', type=int, default=0, help='>0: that many synthetic batches per epoch instead of MPII')
    p.add_argument('--augment', action='store_true', help='with --synthetic: raw variable-size samples prepared on the GPU '
                   '(scale / rotation jitter, flip, colour, crop: data/mpii_for_mpii_22.py:120-145) instead of ready-made batches')
    return p


def parse_options(argv=None):
    """options/base_options.py:39-76: parse, require --exp_id, create the experiment directory, normalise
    --resume_prefix ('lr-...-12.pth.tar' -> 'lr-...-12-'), write opt.txt."""
    opt = build_parser().parse_args(argv)
    if opt.exp_id == '':
        raise SystemExit('Please set the experimental ID with option --exp_id')
    exp_dir = os.path.join(opt.exp_dir, opt.exp_id)
    os.makedirs(exp_dir, exist_ok=True)
    if opt.resume_prefix != '':
        trunc = opt.resume_prefix.index('pth')
        opt.resume_prefix = opt.resume_prefix[0:trunc - 1] + '-'
    with open(os.path.join(exp_dir, 'opt.txt'), 'wt') as f:
        f.write('------------ Options -------------\n')
        for k, v in sorted(vars(opt).items()):
            f.write('%s: %s\n' % (str(k), str(v)))
        f.write('-------------- End ----------------\n')
    return opt


def adjust_lr(opt, trainer: FusedTrainer, epoch: int) -> float:
    """utils/util.py:106-119.  Mutates opt.lr and the trainer's learning rate; returns the rate in force."""
    if epoch < 101:
        return trainer.lr
    if epoch == 101:
        opt.lr = opt.lr * 0.2
    elif epoch == 141:
        opt.lr = opt.lr * 0.5
    elif epoch == 161:
        opt.lr = opt.lr * 0.5
    trainer.lr = float(opt.lr)
    return trainer.lr


class TrainHistory:
    """utils/util.py:8-46: per-epoch records; the same state_dict layout."""

    def __init__(self):
        self.epoch, self.lr, self.loss, self.pckh = [], [], [], []
        self.best_pckh = 0.
        self.is_best = True

    def update(self, epoch, lr, loss, pckh):
        self.epoch.append(epoch); self.lr.append(lr); self.loss.append(loss); self.pckh.append(pckh)
        self.is_best = pckh['val_pckh'] > self.best_pckh
        self.best_pckh = max(pckh['val_pckh'], self.best_pckh)

    def state_dict(self):
        return OrderedDict([('epoch', self.epoch), ('lr', self.lr), ('loss', self.loss), ('pckh', self.pckh),
                            ('best_pckh', self.best_pckh), ('is_best', self.is_best)])

    def load_state_dict(self, sd):
        self.epoch, self.lr, self.loss, self.pckh = sd['epoch'], sd['lr'], sd['loss'], sd['pckh']
        self.best_pckh, self.is_best = sd['best_pckh'], sd['is_best']


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val; self.sum += val * n; self.count += n; self.avg = self.sum / self.count


# ---- checkpoints ------------------------------------------------------------------------------
def _torch_rmsprop_for(net: CUNet, trainer: FusedTrainer) -> torch.optim.RMSprop:
    """A torch RMSprop over the module's parameters whose state tensors are views of the fused trainer's
    flat square-average arena: torch's own (de)serialiser then reads / writes the reference's optimizer layout."""
    opt = torch.optim.RMSprop(net.parameters(), lr=trainer.lr, alpha=trainer.alpha, eps=trainer.eps, momentum=0, weight_decay=0)
    if trainer.square_avg is None:
        trainer.square_avg = torch.zeros_like(net._param_arena)
    off = {name: (o, nmel, shape) for name, kind, shape, o, nmel in net._entries if kind == 0}
    for name, p in net.named_parameters():
        o, nmel, shape = off[name]
        opt.state[p] = {'step': torch.tensor(float(trainer.steps_done)), 'square_avg': trainer.square_avg[o:o + nmel].view(shape)}
    return opt


def checkpoint_dict(net: CUNet, trainer: FusedTrainer, history: TrainHistory) -> dict:
    """utils/checkpoint.py:17-19 with the `module.` prefix of the reference's DataParallel wrapper."""
    sd = OrderedDict(('module.' + k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    osd = _torch_rmsprop_for(net, trainer).state_dict()
    for st in osd['state'].values():
        st['square_avg'] = st['square_avg'].detach().cpu().clone()
    return {'train_history': history.state_dict(), 'state_dict': sd, 'optimizer': osd}


def save_checkpoint(save_prefix: str, net: CUNet, trainer: FusedTrainer, history: TrainHistory) -> str:
    """File name rule of utils/checkpoint.py:14-15: <prefix>lr-<lr without trailing zeros>-<epoch>.pth.tar."""
    lr_prefix = ('lr-%.15f' % history.lr[-1]['lr']).rstrip('0').rstrip('.')
    path = save_prefix + lr_prefix + ('-%d.pth.tar' % history.epoch[-1]['epoch'])
    torch.save(checkpoint_dict(net, trainer, history), path)
    if history.is_best:
        import shutil
        shutil.copyfile(path, save_prefix + lr_prefix + ('-%d-model-best.pth.tar' % history.epoch[-1]['epoch']))
    return path


def load_checkpoint(path: str, net: CUNet, trainer: FusedTrainer, history: TrainHistory) -> bool:
    """utils/checkpoint.py:40-67: copy every stored tensor whose key is known, report the others; keys may or may
    not carry the `module.` prefix.  Restores the optimiser state (square averages, learning rate) as well."""
    if not os.path.isfile(path):
        print("=> no checkpoint found at '{}'".format(path))
        return False
    ck = torch.load(path, map_location='cpu', weights_only=False)
    history.load_state_dict(ck['train_history'])
    net_dict = net.state_dict()
    with torch.no_grad():
        for name, param in ck['state_dict'].items():
            key = name[7:] if name.startswith('module.') and name[7:] in net_dict else name
            if key not in net_dict:
                print("=> not load weights '{}'".format(name))
                continue
            net_dict[key].copy_(param.data if isinstance(param, torch.nn.Parameter) else param)
    opt = _torch_rmsprop_for(net, trainer)
    views = {p: st['square_avg'] for p, st in opt.state.items()}
    opt.load_state_dict(ck['optimizer'])            # torch maps the stored per-parameter state onto net.parameters()
    with torch.no_grad():
        for p, st in opt.state.items():
            if st['square_avg'].data_ptr() != views[p].data_ptr():
                views[p].copy_(st['square_avg'])
            trainer.steps_done = int(float(st.get('step', 0)))
    trainer.lr = float(opt.param_groups[0]['lr'])
    return True


# ---- data -------------------------------------------------------------------------------------
class SyntheticLoader:
    """`nbatches` batches of MPII-shaped synthetic data (uniform images, one Gaussian blob per joint; the bench's
    generator): (img N x 3 x 256 x 256, heatmap N x K x 64 x 64)."""

    def __init__(self, nbatches: int, batch: int, class_num: int, device, seed: int = 0):
        self.nbatches, self.batch, self.k, self.device, self.seed = nbatches, batch, class_num, device, seed

    def __len__(self):
        return self.nbatches

    def __iter__(self):
        from .trainer import pts2heatmap
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.nbatches):
            img = torch.rand(self.batch, 3, 256, 256, generator=g)
            pts = torch.randint(3, 61, (self.batch, self.k, 2), generator=g).double()
            yield img.to(self.device), pts2heatmap(pts.to(self.device), (64, 64), 1)     # targets rendered on the GPU


class SyntheticRawSource:
    """Stand-in for the decoded MPII samples (`--synthetic N --augment`): `count` random images of varying size with an
    annotation record each ('joint_self', 'objpos', 'scale_provided'), already on the GPU as a JPEG decoder would leave them."""

    def __init__(self, count: int, class_num: int, device, seed: int = 0):
        self.count, self.k, self.device, self.seed = count, class_num, device, seed

    def __iter__(self):
        import numpy as np
        rng = np.random.RandomState(self.seed)
        for _ in range(self.count):
            h, w = int(rng.randint(300, 480)), int(rng.randint(300, 640))
            img = torch.from_numpy(rng.uniform(0, 1, size=(3, h, w)).astype('float32')).to(self.device)
            objpos = [w * rng.uniform(0.35, 0.65), h * rng.uniform(0.35, 0.65)]
            scale = rng.uniform(0.9, 1.6)
            joints = np.stack([objpos[0] + rng.uniform(-60, 60, self.k) * scale, objpos[1] + rng.uniform(-80, 80, self.k) * scale], 1)
            yield {'img': img, 'joint_self': joints, 'objpos': objpos, 'scale_provided': scale, 'dataset': 'MPII'}


class AugmentedLoader:
    """Batches raw samples and prepares them on the GPU the way `MPII.__getitem__` does on the CPU
    (data/mpii_for_mpii_22.py:86-145; cu_net_amd.augment.prepare_batch): yields (inp, heatmap, meta)."""

    def __init__(self, source: Iterable, batch: int, is_train: bool = True, scale_factor: float = 0.25, rot_factor: float = 30.0, seed=None):
        self.source, self.batch, self.is_train = source, batch, is_train
        self.scale_factor, self.rot_factor = scale_factor, rot_factor
        import numpy as np
        self.rng = np.random.RandomState(seed) if seed is not None else np.random

    def __iter__(self):
        from .augment import prepare_batch
        buf = []
        for a in self.source:
            buf.append(a)
            if len(buf) == self.batch:
                yield prepare_batch(buf, self.is_train, scale_factor=self.scale_factor, rot_factor=self.rot_factor, rng=self.rng)
                buf = []
        if buf:
            yield prepare_batch(buf, self.is_train, scale_factor=self.scale_factor, rot_factor=self.rot_factor, rng=self.rng)


# ---- loops ------------------------------------------------------------------------------------
def train_epoch(loader: Iterable, trainer: FusedTrainer, epoch: int, opt, idx: List[int] = TRAIN_ACC_IDX, log=print):
    """cu-net.py:152-216: returns (mean loss, mean PCKh on heat-map resolution)."""
    net = trainer.net
    net.train()
    losses, pckhs = AverageMeter(), AverageMeter()
    for i, batch in enumerate(loader):
        img, heatmap = batch[0], batch[1]
        loss = trainer.step(img, heatmap)
        out = trainer.last_outputs(img.shape)[-1]
        idx_ok = [j for j in idx if j < heatmap.shape[1]]
        acc = accuracy(out, heatmap, idx_ok)
        losses.update(float(loss))          # cu-net.py:189,192: update(x) with n = 1 (a mean over batches)
        pckhs.update(float(acc[0]))
        if i % max(int(opt.print_freq), 1) == 0:
            log('epoch %d iter %d  lr %.6g  loss %.6f (%.6f)  pckh %.4f (%.4f)'
                % (epoch, i, trainer.lr, losses.val, losses.avg, pckhs.val, pckhs.avg))
    return losses.avg, pckhs.avg


def validate(loader: Iterable, net: CUNet, idx: List[int] = TRAIN_ACC_IDX, flip_index=JOINT_FLIP_INDEX, process_group=None,
             quan_op=None):
    """cu-net.py:219-258 with flip test-time augmentation; returns (mean loss, mean PCKh, predictions N x K x 2).
    With a process group the two means cover every rank's shard of the validation set (predictions stay per rank).
    `quan_op`: the quantised drivers validate on QUANTISED weights -- quantization() before the loop, restore() after it
    (cu-net-prev-version-wig.py:230,285)."""
    net.eval()
    losses, pckhs = AverageMeter(), AverageMeter()
    preds = []
    if quan_op is not None:
        quan_op.quantization()
    try:
        _validate_batches(loader, net, idx, flip_index, losses, pckhs, preds)
    finally:
        if quan_op is not None:
            quan_op.restore()
    return _validate_finish(net, losses, pckhs, preds, process_group)


def _validate_batches(loader, net, idx, flip_index, losses, pckhs, preds):
    from .trainer import get_preds
    with torch.no_grad():
        for batch in loader:
            img, heatmap = batch[0], batch[1]
            out1 = net(img)
            loss = sum(((o - heatmap) ** 2).sum() / o.numel() for o in out1)
            out2 = net(img.flip(3))
            k = heatmap.shape[1]
            pairs = [p for p in flip_index if max(p) < k]
            out = flip_merge(out1[-1], out2[-1], pairs)
            idx_ok = [j for j in idx if j < k]
            acc = accuracy(out, heatmap, idx_ok)
            losses.update(float(loss))      # cu-net.py:256,265: n = 1
            pckhs.update(float(acc[0]))
            preds.append(get_preds(out).cpu())


def _validate_finish(net, losses, pckhs, preds, process_group):
    loss_avg, pckh_avg = losses.avg, pckhs.avg
    if process_group is not None:
        # every rank validated its own shard of the loader: the reference validates the WHOLE set, so the means are
        # taken over all ranks' batches (sum and count all-reduced) before they reach the history / best-model logic
        import torch.distributed as dist
        dev = next(net.parameters()).device
        acc = torch.tensor([losses.sum, pckhs.sum, float(losses.count)], dtype=torch.float64,
                           device=dev if dist.get_backend(process_group) == 'nccl' else 'cpu')
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=process_group)
        cnt = max(float(acc[2]), 1.0)
        loss_avg, pckh_avg = float(acc[0]) / cnt, float(acc[1]) / cnt
    return loss_avg, pckh_avg, (torch.cat(preds) if preds else torch.zeros(0))


def main(argv=None, train_loader: Optional[Iterable] = None, val_loader: Optional[Iterable] = None):
    """cu-net.py:22-150 for one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run)."""
    opt = parse_options(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD
    exp_dir = os.path.join(opt.exp_dir, opt.exp_id)
    net = create_cu_net(neck_size=4, growth_rate=32, init_chan_num=128, class_num=opt.class_num,
                        layer_num=opt.layer_num, order=opt.order, loss_num=opt.loss_num).to(dev)
    quan = None
    if opt.bits_w > 0:
        from .quant import QuanOp
        quan = QuanOp(net, bits_w=opt.bits_w, bits_i=opt.bits_i, bits_g=opt.bits_g)
    trainer = FusedTrainer(net, lr=opt.lr, alpha=0.99, eps=1e-8, process_group=pg, quan_op=quan, bf16=opt.bf16, bf16_grads=opt.bf16_grads)
    history = TrainHistory()
    save_prefix = exp_dir + '/'
    start_epoch = 0
    if opt.resume_prefix != '':
        if load_checkpoint(os.path.join(exp_dir, opt.resume_prefix)[0:-1] + '.pth.tar', net, trainer, history):
            opt.lr = trainer.lr
            start_epoch = history.epoch[-1]['epoch'] + 1
    trainer.broadcast_parameters(0)
    per_rank = max(opt.bs // world, 1)
    if train_loader is None:
        if opt.synthetic <= 0:
            raise SystemExit('no dataset in this repository: pass --synthetic N or call main(train_loader=..., val_loader=...)')
        if getattr(opt, 'augment', False):
            train_loader = AugmentedLoader(SyntheticRawSource(opt.synthetic * per_rank, opt.class_num, dev, seed=100 + rank), per_rank,
                                           is_train=True, seed=200 + rank)
        else:
            train_loader = SyntheticLoader(opt.synthetic, per_rank, opt.class_num, dev, seed=100 + rank)
        val_loader = SyntheticLoader(max(opt.synthetic // 4, 1), per_rank, opt.class_num, dev, seed=9000 + rank)
    log = print if rank == 0 else (lambda *a, **k: None)
    if not opt.is_train:
        val_loss, val_pckh, _ = validate(val_loader, net, process_group=pg, quan_op=quan)
        log('val loss %.6f  pckh %.4f' % (val_loss, val_pckh))
        return history
    return fit(opt, trainer, history, train_loader, val_loader, start_epoch, rank=rank, process_group=pg,
               save_prefix=save_prefix, log=log)


def fit(opt, trainer: FusedTrainer, history: TrainHistory, train_loader, val_loader, start_epoch: int, rank: int = 0,
        process_group=None, save_prefix: Optional[str] = None, log=print, train_fn=None, validate_fn=None):
    """The epoch loop of cu-net.py:121-150: learning-rate schedule (applied on EVERY epoch, as the reference does),
    one training epoch, validation over the whole set, history record, checkpoint on rank 0."""
    train_fn = train_fn or train_epoch
    validate_fn = validate_fn or validate
    net = trainer.net
    for epoch in range(start_epoch, opt.nEpochs):
        if not opt.no_lr_schedule:          # cu-net.py:125 calls adjust_lr on every epoch, whatever --adjust_lr says
            adjust_lr(opt, trainer, epoch)
        train_loss, train_pckh = train_fn(train_loader, trainer, epoch, opt, log=log)
        vkw = {'quan_op': trainer.quan_op} if getattr(trainer, 'quan_op', None) is not None else {}
        val_loss, val_pckh, _ = validate_fn(val_loader, net, process_group=process_group, **vkw)
        history.update(OrderedDict([('epoch', epoch)]), OrderedDict([('lr', trainer.lr)]),
                       OrderedDict([('train_loss', train_loss), ('val_loss', val_loss)]), OrderedDict([('val_pckh', val_pckh)]))
        if rank == 0 and save_prefix is not None:
            path = save_checkpoint(save_prefix, net, trainer, history)
            log("=> saving '%s'  train %.6f / %.4f  val %.6f / %.4f" % (path, train_loss, train_pckh, val_loss, val_pckh))
    return history
