"""TEST-ONLY: execute the C++ plan description (cunet_plan_describe JSON) with plain torch ops on CPU.

This checks the *wiring* the C++ plan builder produced (which tensors feed which fused node, in what
channel order, through which index map) against the oracle without needing a GPU, and provides
per-tensor activations/gradients so a GPU run can be compared tensor-by-tensor in execution order.
It is not part of the product and the product never imports it."""
import torch
import torch.nn.functional as F


def run_plan(desc, state, x, training=True, want_grads=False, target=None):
    """Returns (outputs list, acts dict name->NCHW tensor, grads dict name->NCHW tensor or {})."""
    T = desc['tensors']
    acts = {}
    st = state

    def bn(inp, name):
        return F.batch_norm(inp, None if training else st[name + '.running_mean'],
                            None if training else st[name + '.running_var'],
                            st[name + '.weight'], st[name + '.bias'], training, 0.1, 1e-5)

    heads = {}
    for n in desc['nodes']:
        op = n['op']
        if op == 'stem_conv':
            y = F.conv2d(x, st[n['conv'] + '.weight'], None, 2, 3)
        elif op == 'stem_bnpool':
            y = F.max_pool2d(F.relu(bn(acts[n['segs'][0]['t']], n['bn'])), 2, 2)
        elif op == 'pool':
            y = F.max_pool2d(acts[n['segs'][0]['t']], 2, 2)
        elif op == 'conv':
            parts = []
            for s in n['segs']:
                a = acts[s['t']]
                if s['ups']:
                    a = F.interpolate(a, scale_factor=2, mode='nearest')
                parts.append(a)
            cat = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
            y = F.conv2d(F.relu(bn(cat, n['bn'])), st[n['conv'] + '.weight'], None, 1, 1 if n['taps'] == 9 else 0)
        else:
            raise ValueError(op)
        if want_grads:
            y.retain_grad()
        acts[n['out']] = y
        if n.get('head', -1) >= 0:
            heads[n['head']] = y
    outs = [heads[i] for i in range(len(heads))]
    grads = {}
    loss = None
    if want_grads:
        loss = 0
        for o in outs:
            d = (o - target) ** 2
            loss = loss + d.sum() / d.numel()
        loss.backward()
        grads = {T[k]['name']: v.grad for k, v in acts.items() if v.grad is not None}
    named = {T[k]['name']: v for k, v in acts.items()}
    return outs, named, grads, loss
