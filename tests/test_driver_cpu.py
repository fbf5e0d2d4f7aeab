"""CPU: the driver's host logic -- reference flag names, learning-rate schedule, checkpoint layout
(options/*.py, utils/util.py:8-46,106-119, utils/checkpoint.py:13-67).  No compute kernels are called."""
import os
from collections import OrderedDict

import torch

import cu_net_amd
from cu_net_amd import driver as D
from cu_net_amd.trainer import FusedTrainer

TINY = dict(neck_size=2, growth_rate=4, init_chan_num=8, class_num=3, layer_num=2, order=1, loss_num=2)


def test_options_follow_the_reference(tmp_path):
    opt = D.parse_options(['--exp_id', 'e1', '--exp_dir', str(tmp_path), '--layer_num', '8', '--loss_num', '8',
                           '--resume_prefix', 'lr-0.00025-12.pth.tar', '--adjust_lr', 'True'])
    assert (opt.layer_num, opt.order, opt.class_num, opt.loss_num, opt.bs) == (8, 1, 16, 8, 24)      # train_options.py defaults
    assert opt.lr == 2.5e-4 and opt.nEpochs == 200 and opt.adjust_lr is True
    assert opt.resume_prefix == 'lr-0.00025-12-'                                                     # base_options.py:63-66
    txt = open(os.path.join(str(tmp_path), 'e1', 'opt.txt')).read()
    assert 'layer_num: 8' in txt and txt.startswith('------------ Options')
    try:
        D.parse_options(['--exp_dir', str(tmp_path)])
        assert False, 'missing --exp_id must stop the run (base_options.py:57-59)'
    except SystemExit:
        pass


def test_adjust_lr_schedule():
    net = cu_net_amd.create_cu_net(**TINY)
    tr = FusedTrainer(net, lr=2.5e-4)
    opt = D.build_parser().parse_args(['--exp_id', 'x'])
    seen = {}
    for epoch in range(0, 170):
        seen[epoch] = D.adjust_lr(opt, tr, epoch)
    assert seen[100] == 2.5e-4
    assert abs(seen[101] - 2.5e-4 * 0.2) < 1e-18 and seen[140] == seen[101]
    assert abs(seen[141] - 2.5e-4 * 0.1) < 1e-18
    assert abs(seen[161] - 2.5e-4 * 0.05) < 1e-18 and seen[169] == seen[161]
    assert tr.lr == seen[169] and opt.lr == seen[169]


def test_checkpoint_layout_and_round_trip(tmp_path):
    torch.manual_seed(0)
    net = cu_net_amd.create_cu_net(**TINY)
    tr = FusedTrainer(net, lr=1e-3)
    tr.square_avg = torch.rand_like(net._param_arena)
    tr.steps_done = 7
    hist = D.TrainHistory()
    hist.update(OrderedDict(epoch=3), OrderedDict(lr=1e-3), OrderedDict(train_loss=0.5, val_loss=0.6), OrderedDict(val_pckh=0.25))
    path = D.save_checkpoint(str(tmp_path) + '/', net, tr, hist)
    assert path.endswith('lr-0.001-3.pth.tar')                                     # utils/checkpoint.py:14-15
    assert os.path.isfile(str(tmp_path) + '/lr-0.001-3-model-best.pth.tar')        # first epoch is the best so far
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {'train_history', 'state_dict', 'optimizer'}
    assert all(k.startswith('module.') for k in ck['state_dict'])                   # what DataParallel's state_dict looks like
    assert len(ck['state_dict']) == len(net.state_dict())
    pg = ck['optimizer']['param_groups'][0]
    assert pg['lr'] == 1e-3 and pg['alpha'] == 0.99 and pg['eps'] == 1e-8 and len(pg['params']) == len(list(net.parameters()))
    # load into a fresh model
    net2 = cu_net_amd.create_cu_net(**TINY)
    tr2 = FusedTrainer(net2, lr=5.0)
    hist2 = D.TrainHistory()
    assert D.load_checkpoint(path, net2, tr2, hist2)
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    for name, kind, shape, o, nmel in net._entries:          # (alignment padding between buckets belongs to no parameter)
        if kind == 0:
            assert torch.equal(tr2.square_avg[o:o + nmel], tr.square_avg[o:o + nmel]), name
    assert tr2.lr == 1e-3 and tr2.steps_done == 7
    assert hist2.epoch[-1]['epoch'] == 3 and hist2.best_pckh == 0.25
    net2._check_aliasing()                                                          # parameters still alias the flat arena


def test_reference_style_checkpoint_loads(tmp_path):
    """A file as the reference writes it (torch 0.4 layout: integer `step`, parameter ids 0..n-1, `module.` keys,
    an entry the model does not have)."""
    torch.manual_seed(1)
    src = cu_net_amd.create_cu_net(**TINY)
    params = list(src.parameters())
    state = {i: {'step': 11, 'square_avg': torch.full_like(p, 0.01 * (i + 1))} for i, p in enumerate(params)}
    osd = {'state': state, 'param_groups': [{'lr': 5e-5, 'momentum': 0, 'alpha': 0.99, 'eps': 1e-8, 'centered': False,
                                             'weight_decay': 0, 'params': list(range(len(params)))}]}
    sd = OrderedDict(('module.' + k, v.clone()) for k, v in src.state_dict().items())
    sd['module.not_in_this_model.weight'] = torch.zeros(3)
    hist = D.TrainHistory()
    hist.update(OrderedDict(epoch=120), OrderedDict(lr=5e-5), OrderedDict(train_loss=0.1, val_loss=0.2), OrderedDict(val_pckh=0.8))
    path = os.path.join(str(tmp_path), 'lr-0.00005-120.pth.tar')
    torch.save({'train_history': hist.state_dict(), 'state_dict': sd, 'optimizer': osd}, path)
    net = cu_net_amd.create_cu_net(**TINY)
    tr = FusedTrainer(net)
    h = D.TrainHistory()
    assert D.load_checkpoint(path, net, tr, h)
    for (k, a), (_, b) in zip(src.state_dict().items(), net.state_dict().items()):
        assert torch.equal(a, b), k
    off = {name: (o, nmel) for name, kind, shape, o, nmel in net._entries if kind == 0}
    for i, (name, p) in enumerate(net.named_parameters()):
        o, nmel = off[name]
        assert torch.all(tr.square_avg[o:o + nmel] == 0.01 * (i + 1)), name
    assert tr.lr == 5e-5 and tr.steps_done == 11 and h.epoch[-1]['epoch'] == 120


def test_epoch_loop_applies_the_schedule_without_a_flag(tmp_path):
    """cu-net.py:125 calls adjust_lr on every epoch and never reads --adjust_lr: a run with the reference's command
    line must decay at epochs 101 / 141 / 161, and the checkpoint names / history.lr must carry the decayed rate."""
    net = cu_net_amd.create_cu_net(**TINY)
    opt = D.parse_options(['--exp_id', 'sched', '--exp_dir', str(tmp_path), '--nEpochs', '163', '--lr', '2.5e-4'])
    assert opt.adjust_lr is False and opt.no_lr_schedule is False
    tr = FusedTrainer(net, lr=opt.lr)
    hist = D.TrainHistory()
    calls = []

    def fake_train(loader, trainer, epoch, o, log=None):
        calls.append((epoch, trainer.lr))
        return 1.0 / (epoch + 1), 0.5

    def fake_validate(loader, n, process_group=None):
        return 0.25, 0.1 + 1e-3 * len(calls), torch.zeros(0)

    D.fit(opt, tr, hist, None, None, 99, save_prefix=None, log=lambda *a: None, train_fn=fake_train, validate_fn=fake_validate)
    lr = {e['epoch']: r['lr'] for e, r in zip(hist.epoch, hist.lr)}
    assert lr[100] == 2.5e-4
    assert abs(lr[101] - 5e-5) < 1e-18 and lr[140] == lr[101]
    assert abs(lr[141] - 2.5e-5) < 1e-18 and abs(lr[161] - 1.25e-5) < 1e-18 and tr.lr == lr[162]
    assert dict(calls)[101] == lr[101]                       # the decayed rate is in force DURING epoch 101
    # opt-out (not a reference feature)
    opt2 = D.parse_options(['--exp_id', 'sched2', '--exp_dir', str(tmp_path), '--nEpochs', '103', '--no_lr_schedule', 'True'])
    tr2 = FusedTrainer(net, lr=opt2.lr)
    D.fit(opt2, tr2, D.TrainHistory(), None, None, 100, save_prefix=None, log=lambda *a: None, train_fn=fake_train, validate_fn=fake_validate)
    assert tr2.lr == 2.5e-4
