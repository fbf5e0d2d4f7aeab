"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by tools/gen_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CFG_KEYS = ('neck_size', 'growth_rate', 'init_chan_num', 'class_num', 'layer_num', 'order', 'loss_num')
TINY = ['G1_L2_o1', 'G2_L3_o2', 'G3_L4_o1_ln2', 'G4_L2_o0', 'G9_L2_o1_c32', 'G6_L2_o1_hw64']


class Golden:
    def __init__(self, tag):
        self.tag = tag
        self.z = np.load(os.path.join(GOLDEN_DIR, tag + '.npz'))
        self.cfg = {k: int(v) for k, v in zip(CFG_KEYS, self.z['cfg'])}

    def t(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def group(self, prefix):
        """dict of tensors whose npz key starts with `prefix/` (insertion order preserved)."""
        p = prefix + '/'
        return {k[len(p):]: torch.from_numpy(np.asarray(self.z[k])) for k in self.z.files if k.startswith(p)}

    def list(self, prefix):
        g = self.group(prefix)
        return [g[str(i)] for i in range(len(g))]
