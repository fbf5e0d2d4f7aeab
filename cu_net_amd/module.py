"""`create_cu_net(...)` on the MI355X-native HIP path.

Mirrors the reference's public surface (file:line in the reference tree):
  * `create_cu_net(neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num)`
    -> `nn.Module`                                              models/cu_net.py:362-368
  * `net(img)` -> python list of `loss_num` tensors N x K x H/4 x W/4      models/cu_net.py:336-360
  * `state_dict()` keys / shapes / order identical to the reference (so its checkpoints load,
    utils/checkpoint.py:40-67), real `nn.Conv2d` / `nn.BatchNorm2d` leaves in the reference's
    `modules()` order (so utils/quantize.py:81-102 finds the same weights), `.train()/.eval()`,
    `.parameters()` for torch optimisers, `loss.backward()` through one autograd Function.

Unlike the reference, the leaves are only *parameter containers*: every parameter is a view into
one flat fp32 arena (bucket-major, see include/cunet.h) and `forward` hands raw device pointers to
libcunet_hip.so, which runs the whole network as hand-written HIP kernels.  There is no PyTorch /
CPU fallback: a CPU tensor, a missing library or an unsupported mode raises `CUNetError`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from ._lib import BUCKET_CB, CUNetError, PlanHandle, check, lib

__all__ = ['create_cu_net', 'CUNet', 'CUNetError']


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _BoundPlan:
    """A plan specialised for (batch, H, W) with its workspace, bound to the module's arenas."""

    def __init__(self, net: 'CUNet', n: int, h: int, w: int, training_ws: bool, bf16: bool = False):
        self.handle = PlanHandle(*net._hyper, batch=n, height=h, width=w)
        self.shape = (n, h, w)
        self.training_ws = training_ws
        dev = net._param_arena.device
        nbytes = self.handle.workspace_bytes((3 if training_ws else 2) if bf16 else training_ws)      # 2 / 3: + bf16 arena
        self.bf16 = bf16
        self.grad_bf16 = False
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.hw_out = (h // 4, w // 4)
        self.num_heads = self.handle.num_heads
        self.generation = 0
        check(lib().cunet_bind(self.handle.h, _ptr(net._param_arena), _ptr(net._grad_arena) if training_ws else None,
                               _ptr(net._buffer_arena), _ptr(net._counter_arena), _ptr(self.workspace), nbytes,
                               1 if training_ws else 0, _stream_ptr(dev)), 'cunet_bind')
        self._cb_keepalive = None
        import weakref
        self._net = weakref.ref(net)
        self.popcount_nodes = 0
        if net._quant_input[0] and not bf16:
            self.popcount_nodes = self.handle.set_quant_input(*net._quant_input)

    def forward(self, x: torch.Tensor, training: bool, want_outputs: bool = True) -> List[torch.Tensor]:
        n, h, w = self.shape
        outs = []
        arr = (C.c_void_p * self.num_heads)()
        if want_outputs:
            k = self.handle.cfg.class_num
            for i in range(self.num_heads):
                o = torch.empty((n, k, self.hw_out[0], self.hw_out[1]), dtype=torch.float32, device=x.device)
                outs.append(o)
                arr[i] = o.data_ptr()
        if self.popcount_nodes:
            # the AND-popcount forward is only correct on ternary weights: it runs while a QuanOp holds the module's
            # weights quantised (between quantization() and restore()), the MFMA path with the same quantiser otherwise
            net = self._net()
            live = 1 if (net is not None and net._weights_ternary) else 0
            check(lib().cunet_set_popcount_live(self.handle.h, live), 'cunet_set_popcount_live')
        check(lib().cunet_forward(self.handle.h, _ptr(x), arr if want_outputs else None, 1 if training else 0,
                                  _stream_ptr(x.device)), 'cunet_forward')
        self.generation += 1
        self._last_x = x          # the stem's weight gradient re-reads the image during backward
        return outs

    def forward_bf16(self, x: torch.Tensor, training: bool = False, want_outputs: bool = True) -> List[torch.Tensor]:
        n, h, w = self.shape
        k = self.handle.cfg.class_num
        outs = []
        arr = None
        if want_outputs:
            outs = [torch.empty((n, k, self.hw_out[0], self.hw_out[1]), dtype=torch.float32, device=x.device) for _ in range(self.num_heads)]
            arr = (C.c_void_p * self.num_heads)(*[o.data_ptr() for o in outs])
        self.grad_bf16 = int(training) == 2               # True/1: bf16 activations; 2: bf16 gradient tensors as well
        check(lib().cunet_forward_bf16(self.handle.h, _ptr(x), arr, int(training), _stream_ptr(x.device)), 'cunet_forward_bf16')
        self.generation += 1
        self._last_x = x
        return outs

    def loss_mse(self, target: torch.Tensor) -> torch.Tensor:
        check(lib().cunet_loss_mse(self.handle.h, _ptr(target), _ptr(self.loss), _stream_ptr(target.device)),
              'cunet_loss_mse')
        return self.loss

    def stage_target(self, target: torch.Tensor) -> torch.Tensor:
        """cunet_loss_mse_fused: the NEXT training forward computes the MSE and its gradient in the heads' epilogues; returns the
        0-dim device tensor that forward will write the loss to."""
        check(lib().cunet_loss_mse_fused(self.handle.h, _ptr(target), _ptr(self.loss), _stream_ptr(target.device)),
              'cunet_loss_mse_fused')
        self._staged_target = target          # read asynchronously by the transpose
        return self.loss

    def backward(self, grad_heat=None, on_bucket=None):
        dev = self.workspace.device
        arr = None
        if grad_heat is not None:
            arr = (C.c_void_p * self.num_heads)()
            for i, g in enumerate(grad_heat):
                arr[i] = g.data_ptr() if g is not None else None
        if on_bucket is None:
            check(lib().cunet_backward(self.handle.h, arr, _stream_ptr(dev)), 'cunet_backward')
        else:
            # ctypes prints and SWALLOWS an exception raised inside a C callback: catch it here, tell C to stop
            # (non-zero return -> CUNET_ERR_CALLBACK) and re-raise it once cunet_backward_ex has returned
            failure = []

            def _cb(b, _u):
                try:
                    on_bucket(int(b))
                    return 0
                except BaseException as e:          # noqa: BLE001 - must not propagate into C
                    failure.append(e)
                    return 1
            cb = BUCKET_CB(_cb)
            self._cb_keepalive = cb
            rc = lib().cunet_backward_ex(self.handle.h, arr, _stream_ptr(dev), cb, None)
            if failure:
                raise failure[0]
            check(rc, 'cunet_backward_ex')

    def side_stream_join(self, stream_ptr):
        """Make the stream (raw hipStream_t as c_void_p) wait for the weight gradients enqueued so far."""
        check(lib().cunet_side_stream_join(self.handle.h, stream_ptr), 'cunet_side_stream_join')

    def debug_poke(self, name: str, value: torch.Tensor, grad: bool = True):
        """Overwrite an internal NHWC tensor from an NCHW one (kernel unit tests only)."""
        d = self.handle.describe()
        t = [t for t in d['tensors'] if t['name'] == name][0]
        off = self.handle.tensor_offset(name, 1 if grad else 0)
        rows = t['N'] * t['H'] * t['W']
        if grad and self.grad_bf16 and self._grad_is_bf16(d, t):       # a bf16 gradient tensor sits in the first half of its slot
            gld = t.get('gld16', t['ld'])                               # (heads: K padded to a 32-multiple, zeros behind it)
            flat = self.workspace[off: off + rows * gld * 2].view(torch.bfloat16).view(t['N'], t['H'], t['W'], gld)
        else:
            flat = self.workspace[off: off + rows * t['ld'] * 4].view(torch.float32).view(t['N'], t['H'], t['W'], t['ld'])
        flat.zero_()
        flat[..., :t['C']] = value.to(flat.device).permute(0, 2, 3, 1).to(flat.dtype)

    @staticmethod
    def _grad_is_bf16(d, t) -> bool:
        """With bf16 gradient storage every gradient tensor is bf16 except d(loss)/d(stem conv output)."""
        return t['id'] != d['nodes'][0]['out']

    def debug_set_option(self, name: str, value: int):
        """include/cunet.h cunet_debug_set_plan_option: flips a launch-time option (wgrad_bf16_dma) in this live plan's snapshot."""
        check(lib().cunet_debug_set_plan_option(self.handle.h, name.encode(), int(value)), 'cunet_debug_set_plan_option')

    def debug_run_node_backward(self, node_index: int):
        check(lib().cunet_debug_run_node_backward(self.handle.h, node_index, _stream_ptr(self.workspace.device)),
              'cunet_debug_run_node_backward')

    def debug_tensor(self, name: str, grad: bool = False) -> torch.Tensor:
        """NCHW copy of an internal NHWC tensor (tests only)."""
        if grad:      # a gradient tensor the last backward did not materialise (planner option stem_fuse_dz) is written now
            check(lib().cunet_debug_materialise(self.handle.h, _stream_ptr(self.workspace.device)), 'cunet_debug_materialise')
        d = self.handle.describe()
        t = [t for t in d['tensors'] if t['name'] == name][0]
        rows = t['N'] * t['H'] * t['W']
        if grad and self.grad_bf16 and self._grad_is_bf16(d, t):
            off = self.handle.tensor_offset(name, 1)
            gld = t.get('gld16', t['ld'])
            flat = self.workspace[off: off + rows * gld * 2].view(torch.bfloat16).float()
            return flat.view(t['N'], t['H'], t['W'], gld)[..., :t['C']].permute(0, 3, 1, 2).contiguous()
        elif self.bf16 and not grad and self._in_bf16_arena(d, t):
            base = (self.handle.workspace_bytes(self.training_ws) + 255) // 256 * 256      # where the bf16 arena starts
            flat = self.workspace[base + 2 * t['act']: base + 2 * (t['act'] + rows * t['ld'])].view(torch.bfloat16).float()
        else:
            off = self.handle.tensor_offset(name, 1 if grad else 0)
            flat = self.workspace[off: off + rows * t['ld'] * 4].view(torch.float32)
        return flat.view(t['N'], t['H'], t['W'], t['ld'])[..., :t['C']].permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def _in_bf16_arena(d, t) -> bool:
        """With bf16 activations everything but the stem conv output and the heads lives in the bf16 arena."""
        fp32 = {d['nodes'][0]['out']} | {n['out'] for n in d['nodes'] if n.get('head', -1) >= 0}
        return t['id'] not in fp32


class _CUNetFunction(torch.autograd.Function):
    """One autograd node for the whole network: the HIP side owns activations and recompute."""

    @staticmethod
    def forward(ctx, net, plan, x, *params):
        outs = plan.forward(x, True)
        ctx.net = net
        ctx.plan = plan
        ctx.generation = plan.generation
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        net, plan = ctx.net, ctx.plan
        if plan.generation != ctx.generation:
            raise CUNetError('backward() after another forward() on the same (batch, H, W): the HIP plan keeps '
                             'one set of activations per shape')
        gs = [g.contiguous() if g is not None else None for g in grad_outs]
        plan.backward(gs)
        grads = net._grad_views_for_autograd()
        return (None, None, None) + tuple(grads)


class CUNet(nn.Module):
    def __init__(self, neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num):
        super().__init__()
        # same checks as models/cu_net.py:274-287 (exit() replaced by ValueError)
        assert loss_num <= layer_num and loss_num >= 1
        if order >= layer_num:
            raise ValueError('order is larger than the layer number.')
        self._hyper = (int(neck_size), int(growth_rate), int(init_chan_num), int(class_num), int(layer_num),
                       int(order), int(loss_num))
        self.layer_num = layer_num
        layout = PlanHandle(*self._hyper, batch=1, height=64, width=64)   # raises CUNetError if unsupported
        self.loss_anchors = layout.anchors()
        assert layer_num in self.loss_anchors and loss_num == len(self.loss_anchors)
        self._entries = layout.state_entries()
        self._n_params = layout.param_numel
        self._n_buffers = layout.buffer_numel
        self._n_counters = layout.counter_numel
        self._buckets = layout.buckets()
        self._build_tree()
        self._reference_init()
        self._plans: Dict[Tuple[int, int, int], _BoundPlan] = {}
        self._quant_input = (0, ())        # (bits_i, ternary conv names): see set_quant_input
        self._weights_ternary = False      # set by QuanOp.quantization() (bits_w 1 / 2, scale dropped), cleared by restore()
        self._param_arena = None
        self._flatten(torch.device('cpu'))

    # ---- module tree: real Conv2d / BatchNorm2d leaves under the reference's attribute names ----
    def _build_tree(self):
        leaves: Dict[str, dict] = {}
        order: List[str] = []
        for name, kind, shape, off, numel in self._entries:
            path, attr = name.rsplit('.', 1)
            if path not in leaves:
                leaves[path] = {}
                order.append(path)
            leaves[path][attr] = shape

        def construction_key(path):   # the reference CREATES down/up blocks interleaved (cu_net.py:234-242)
            p = path.split('.')
            if p[0] == 'features':
                return (0, 0, 0)
            if p[0] == 'hg':
                if p[1] == 'down_blocks':
                    return (1, int(p[2]), 0)
                if p[1] == 'up_blocks':
                    return (1, int(p[2]), 1)
                return (2, 0, 0)
            return (3 if p[0] == 'linears' else 4, 0, 0)

        made = {}
        for path in sorted(order, key=lambda q: (construction_key(q), order.index(q))):
            attrs = leaves[path]
            if 'running_mean' in attrs:
                made[path] = nn.BatchNorm2d(attrs['weight'][0])
            else:
                o, i, kh, kw = attrs['weight']
                stride, pad = (2, 3) if kh == 7 else (1, kh // 2)
                made[path] = nn.Conv2d(i, o, kernel_size=kh, stride=stride, padding=pad, bias=False)
        # attach in registration (= state_dict) order
        for path in order:
            parts = path.split('.')
            cur = self
            for depth, part in enumerate(parts[:-1]):
                nxt = cur._modules.get(part)
                if nxt is None:
                    numeric_children = parts[depth + 1].isdigit()
                    nxt = nn.ModuleList() if numeric_children else nn.Module()
                    cur.add_module(part, nxt)
                cur = nxt
            cur.add_module(parts[-1], made[path])
        self._leaf_paths = order

    def _reference_init(self):
        """models/cu_net.py:322-334, same module order, same RNG stream."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
                stdv = 1 / math.sqrt(n)
                m.weight.data.uniform_(-stdv, stdv)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.uniform_()
                m.bias.data.zero_()

    # ---- flat arenas ---------------------------------------------------------------------------
    def _leaf_tensor(self, name: str):
        path, attr = name.rsplit('.', 1)
        mod = self.get_submodule(path)
        return mod, attr

    def _flatten(self, device):
        """(Re)create the arenas on `device` from the current leaf values and alias the leaves.
        When the arenas already live on `device` they are kept (and so are the bound plans): only
        leaves whose storage was replaced are copied back in."""
        device = torch.device(device)
        fresh = getattr(self, '_param_arena', None) is None or self._param_arena.device != device
        if fresh:
            self._param_arena = torch.zeros(self._n_params, dtype=torch.float32, device=device)
            self._grad_arena = torch.zeros(self._n_params, dtype=torch.float32, device=device)
            self._buffer_arena = torch.zeros(self._n_buffers, dtype=torch.float32, device=device)
            self._counter_arena = torch.zeros(max(self._n_counters, 1), dtype=torch.int64, device=device)
            self._plans = {}
        pa, ba, ca = self._param_arena, self._buffer_arena, self._counter_arena
        self._param_list: List[nn.Parameter] = []
        self._param_meta: List[Tuple[int, int, tuple, bool]] = []
        anchors = set(self.loss_anchors)
        with torch.no_grad():
            for name, kind, shape, off, numel in self._entries:
                mod, attr = self._leaf_tensor(name)
                if kind == 0:
                    p = mod._parameters[attr]
                    view = pa[off:off + numel].view(shape)
                    if p.data_ptr() != view.data_ptr() or p.device != device:
                        view.copy_(p.data.to(device=device, dtype=torch.float32))
                        p.data = view
                        p.grad = None
                    self._param_list.append(p)
                    head_unused = name.startswith('linears.') and (int(name.split('.')[1]) + 1) not in anchors
                    self._param_meta.append((off, numel, shape, head_unused))
                else:
                    b = mod._buffers[attr]
                    view = ba[off:off + numel].view(shape) if kind == 1 else ca[off:off + 1].view(())
                    if b.data_ptr() != view.data_ptr() or b.device != device:
                        view.copy_(b.to(device=device, dtype=view.dtype))
                        mod._buffers[attr] = view

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        # .cuda()/.to()/.cpu() moved every leaf separately: rebuild the arenas where they landed
        p0 = self._param_list[0]
        if p0.dtype != torch.float32:
            raise CUNetError('the HIP path is fp32; do not cast the module (bf16 storage is a plan option, not .half())')
        self._flatten(p0.device)
        return self

    def _check_aliasing(self):
        """A quantiser may replace `.data` (utils/quantize.py:115 `w.data = w.data.add(..)`): copy such
        strays back into the arena and re-alias (arenas and plans are kept)."""
        base = self._param_arena.data_ptr()
        for p, (off, numel, shape, _) in zip(self._param_list, self._param_meta):
            if p.data_ptr() != base + 4 * off:
                self._flatten(self._param_arena.device)
                return

    def _grad_views_for_autograd(self):
        # views of ONE clone of the gradient arena: autograd may keep (steal) what we return, and the
        # arena itself is overwritten by the next backward
        g = self._grad_arena.clone()
        out = []
        for off, numel, shape, unused in self._param_meta:
            out.append(None if unused else g[off:off + numel].view(shape))
        return out

    # ---- plans -----------------------------------------------------------------------------------
    def _get_plan(self, n, h, w, need_grad, bf16: bool = False) -> _BoundPlan:
        key = ('bf16', n, h, w) if bf16 else (n, h, w)
        plan = self._plans.get(key)
        if plan is None or (need_grad and not plan.training_ws):
            if plan is None and len(self._plans) >= 4:      # bound the workspaces kept alive
                self._plans.pop(next(iter(self._plans)))
            self._plans.pop(key, None)
            plan = _BoundPlan(self, n, h, w, need_grad, bf16=bf16)
            self._plans[key] = plan
        return plan

    def set_quant_input(self, bits_i: int = 8, ternary_convs=()):
        """Quantised-input mode: the reference's QuanInput2d (utils/quantize.py:47-73) between the ReLU and every 3x3 conv and
        every head conv, where models/cu_net_prev_version_wig.py:96-98,277-279 places it.  `ternary_convs`: module paths of
        convs whose weights are kept in {-1, 0, +1} during forward / backward (QuanOp targets at bits_w 1 or 2): their
        forward runs on the AND-popcount kernel.  bits_i = 0 switches the mode off.  Applies to every (batch, H, W) plan."""
        self._quant_input = (int(bits_i), tuple(ternary_convs))
        for key, plan in self._plans.items():
            if not plan.bf16:
                plan.popcount_nodes = plan.handle.set_quant_input(*self._quant_input)

    def forward_bf16(self, x):
        """Inference with bf16 storage: activations and weights are bf16 between the stem and the heads, contracted
        with bf16 MFMA into fp32 accumulators; BatchNorm (running statistics) + ReLU in fp32.  Same inputs and outputs
        as `forward` in eval mode (fp32 NCHW); heat maps agree with it to about 1e-2 of their range.  Not a reference
        feature (the reference is fp32): BASELINE config 3's storage format, inference half."""
        if self.training:
            raise CUNetError('forward_bf16 is an inference path: call net.eval() first')
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda or x.dtype != torch.float32:
            raise CUNetError('input must be an fp32 N x 3 x H x W GPU tensor')
        self._check_aliasing()
        x = x.contiguous()
        n, _, h, w = x.shape
        # the bf16 kernels work on whole 32-row tiles at every level: N * (H/64) * (W/64) must be a multiple of 32 at the neck.  In
        # eval mode images do not interact (running statistics), so a batch that does not satisfy it -- e.g. ONE 256 x 256 image --
        # is padded with zero images and the extra heat maps are dropped
        per = (h // 64) * (w // 64)
        pad = 0
        while ((n + pad) * per) % 32:
            pad += 1
        if pad:
            x = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], 0)
        outs = self._get_plan(n + pad, h, w, False, bf16=True).forward_bf16(x)
        return [o[:n] for o in outs] if pad else outs

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise CUNetError('input must be an N x 3 x H x W tensor')
        if not x.is_cuda:
            raise CUNetError('the CU-Net HIP path needs a GPU tensor (there is no CPU fallback; the CPU oracle lives in oracle/)')
        if x.dtype != torch.float32:
            raise CUNetError('input must be fp32')
        if getattr(self, '_is_replica', False):
            raise CUNetError('torch.nn.DataParallel replication is not supported: run one process per GPU and use '
                             'cu_net_amd.parallel (RCCL all-reduce on gradient buckets)')
        if self._param_arena.device != x.device:
            raise CUNetError(f'module is on {self._param_arena.device}, input on {x.device}: call net.cuda() first')
        self._check_aliasing()
        x = x.contiguous()
        n, _, h, w = x.shape
        need_grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list)
        plan = self._get_plan(n, h, w, need_grad)
        if need_grad:
            return list(_CUNetFunction.apply(self, plan, x, *self._param_list))
        return plan.forward(x, self.training)


def create_cu_net(neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num):
    """Drop-in for models/cu_net.py:362-368."""
    return CUNet(neck_size=neck_size, growth_rate=growth_rate, init_chan_num=init_chan_num,
                 class_num=class_num, layer_num=layer_num, order=order, loss_num=loss_num)
