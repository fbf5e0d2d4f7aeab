"""ctypes binding of libcunet_hip.so (include/cunet.h).

The library is built in-tree by `cu_net_amd/csrc/build.sh` (or `__graft_entry__.build()`).
There is NO fallback: if the shared object is missing or a call fails, a CUNetError is raised.
"""
from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUNET_LIB_PATH: tools/ point this at libcunet_hip_tuning.so (the -DCUNET_TUNING build with its environment knobs);
# tests, smoke() and bench.py run the shipped library.
LIB_PATH = os.environ.get('CUNET_LIB_PATH') or os.path.join(_HERE, 'libcunet_hip.so')


class CUNetError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('neck_size', 'growth_rate', 'init_chan_num', 'class_num', 'layer_num',
                                         'order', 'loss_num', 'batch', 'height', 'width')]


class StateDesc(C.Structure):
    _fields_ = [('name', C.c_char * 160), ('kind', C.c_int32), ('ndim', C.c_int32), ('shape', C.c_int64 * 4),
                ('offset', C.c_int64), ('numel', C.c_int64)]


BUCKET_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p)     # int (*)(int bucket, void* user): non-zero aborts backward

_lib = None


def lib():
    """Load the HIP library (once). Raises CUNetError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CUNetError(f'{LIB_PATH} not found: build it with cu_net_amd/csrc/build.sh '
                         '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        'cunet_last_error': (C.c_char_p, []),
        'cunet_version': (C.c_char_p, []),
        'cunet_plan_create': (i32, [C.POINTER(Cfg), C.POINTER(vp)]),
        'cunet_set_planner_option': (i32, [C.c_char_p, i32]),
        'cunet_get_planner_option': (i32, [C.c_char_p, C.POINTER(i32)]),
        'cunet_debug_set_plan_option': (i32, [vp, C.c_char_p, i32]),
        'cunet_debug_materialise': (i32, [vp, vp]),
        'cunet_plan_destroy': (None, [vp]),
        'cunet_state_count': (i32, [vp]),
        'cunet_state_entry': (i32, [vp, i32, C.POINTER(StateDesc)]),
        'cunet_param_numel': (i64, [vp]),
        'cunet_buffer_numel': (i64, [vp]),
        'cunet_counter_numel': (i64, [vp]),
        'cunet_workspace_bytes': (i64, [vp, i32]),
        'cunet_num_heads': (i32, [vp]),
        'cunet_loss_anchors': (i32, [vp, C.POINTER(C.c_int32), i32]),
        'cunet_plan_describe': (C.c_char_p, [vp]),
        'cunet_bind': (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp]),
        'cunet_set_quant_input': (i32, [vp, i32, C.POINTER(C.c_char_p), i32]),
        'cunet_set_popcount_live': (i32, [vp, i32]),
        'cunet_forward': (i32, [vp, vp, C.POINTER(vp), i32, vp]),
        'cunet_loss_mse': (i32, [vp, vp, vp, vp]),
        'cunet_loss_mse_fused': (i32, [vp, vp, vp, vp]),
        'cunet_backward': (i32, [vp, C.POINTER(vp), vp]),
        'cunet_num_buckets': (i32, [vp]),
        'cunet_bucket_range': (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
        'cunet_bucket_order': (i32, [vp, C.POINTER(C.c_int32), i32]),
        'cunet_backward_ex': (i32, [vp, C.POINTER(vp), vp, BUCKET_CB, vp]),
        'cunet_side_stream_join': (i32, [vp, vp]),
        'cunet_forward_bf16': (i32, [vp, vp, C.POINTER(vp), i32, vp]),
        'cunet_rmsprop_step': (i32, [vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, vp]),
        'cunet_get_preds': (i32, [vp, vp, i32, i32, i32, i32, vp]),
        'cunet_final_preds': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        'cunet_final_preds_affine': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        'cunet_flip_merge': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        'cunet_augment_batch': (i32, [vp, vp, i32, vp, i32, vp]),
        'cunet_render_targets': (i32, [vp, vp, i32, vp, i32, i32, i32, vp]),
        'cunet_debug_tensor_offset': (i64, [vp, C.c_char_p, i32]),
        'cunet_quant_prepare': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        'cunet_quant_restore': (i32, [vp, vp, vp, i32, vp]),
        'cunet_quant_grad': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        'cunet_ternary_pack': (i32, [vp, vp, vp, i32, i32, i32, vp]),
        'cunet_ternary_conv': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
        'cunet_ternary_conv_ex': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
        'cunet_debug_run_node_backward': (i32, [vp, i32, vp]),
        'cunet_profile_begin': (i32, [vp, i32, i32]),
        'cunet_profile_reset': (i32, [vp]),
        'cunet_profile_collect': (i32, [vp]),
        'cunet_profile_num_classes': (i32, []),
        'cunet_profile_class_name': (C.c_char_p, [i32]),
        'cunet_profile_get': (i32, [vp, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
        'cunet_profile_get_stream': (i32, [vp, i32, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)      # AttributeError here == the ABI in include/cunet.h is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED = ['cunet_last_error', 'cunet_version', 'cunet_plan_create', 'cunet_set_planner_option', 'cunet_get_planner_option', 'cunet_debug_set_plan_option', 'cunet_debug_materialise', 'cunet_plan_destroy', 'cunet_state_count',
            'cunet_state_entry', 'cunet_param_numel', 'cunet_buffer_numel', 'cunet_counter_numel',
            'cunet_workspace_bytes', 'cunet_num_heads', 'cunet_loss_anchors', 'cunet_plan_describe', 'cunet_bind', 'cunet_set_quant_input', 'cunet_set_popcount_live',
            'cunet_forward', 'cunet_loss_mse', 'cunet_loss_mse_fused', 'cunet_backward', 'cunet_backward_ex', 'cunet_side_stream_join', 'cunet_forward_bf16', 'cunet_bucket_order', 'cunet_num_buckets',
            'cunet_bucket_range', 'cunet_rmsprop_step', 'cunet_get_preds', 'cunet_final_preds', 'cunet_final_preds_affine', 'cunet_flip_merge', 'cunet_augment_batch', 'cunet_render_targets',
            'cunet_debug_tensor_offset', 'cunet_quant_prepare', 'cunet_quant_restore', 'cunet_quant_grad',
            'cunet_ternary_pack', 'cunet_ternary_conv', 'cunet_ternary_conv_ex', 'cunet_debug_run_node_backward', 'cunet_profile_begin', 'cunet_profile_reset', 'cunet_profile_collect',
            'cunet_profile_num_classes', 'cunet_profile_class_name', 'cunet_profile_get', 'cunet_profile_get_stream']


def check(rc: int, what: str = ''):
    if rc < 0:
        msg = lib().cunet_last_error().decode()
        raise CUNetError(f'{what}: {msg} (status {rc})')
    return rc


PLANNER_OPTIONS = ('wgrad3_min_rows', 'wgrad3_min_chunks', 'wgrad3_max_splits', 'wgrad3_min_chunks_bf16', 'wgrad3_max_splits_bf16',
                   'wgrad3_stem', 'conv3x3_ring_min_rows', 'wgrad_fork_group', 'wgrad_fork_group_bf16', 'fwd_fork_min_w', 'pair_adapters',
                   'heads_on_side', 'dgrad_nt', 'wgrad_bf16_dma', 'fuse_wgrad', 'dgrad_prefetch', 'dgrad_rows', 'f32_split', 'dgrad3_nt',
                   'dgrad3_ring', 'stem_split', 'dgrad_rows_v', 'popcount_pixels', 'stem_fuse_dz', 'stem_wgrad_split',
                   'fuse_pool_gather', 'fuse_z_gather', 'stem_wgrad_caller', 'wgrad_split_planes', 'stem_wgrad_planes')


def set_planner_option(name: str, value: int):
    """include/cunet.h cunet_set_planner_option: kernel selection knobs read when a plan is created."""
    check(lib().cunet_set_planner_option(name.encode(), int(value)), 'cunet_set_planner_option')


def get_planner_option(name: str) -> int:
    """include/cunet.h cunet_get_planner_option: the process-wide value a plan created now would snapshot."""
    v = C.c_int32()
    check(lib().cunet_get_planner_option(name.encode(), C.byref(v)), 'cunet_get_planner_option')
    return int(v.value)


def planner_options_snapshot() -> dict:
    return {k: get_planner_option(k) for k in PLANNER_OPTIONS}


class PlanHandle:
    """Owns one cunet_plan_t (host object; device memory stays caller-owned)."""

    def __init__(self, neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num,
                 batch, height, width):
        L = lib()
        self.cfg = Cfg(neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num,
                       batch, height, width)
        h = C.c_void_p()
        check(L.cunet_plan_create(C.byref(self.cfg), C.byref(h)), 'cunet_plan_create')
        self.h = h
        self._desc = None

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                lib().cunet_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- layout
    def state_entries(self):
        L = lib()
        out = []
        d = StateDesc()
        for i in range(L.cunet_state_count(self.h)):
            check(L.cunet_state_entry(self.h, i, C.byref(d)), 'cunet_state_entry')
            out.append((d.name.decode(), int(d.kind), tuple(int(d.shape[k]) for k in range(d.ndim)),
                        int(d.offset), int(d.numel)))
        return out

    @property
    def param_numel(self):
        return int(lib().cunet_param_numel(self.h))

    @property
    def buffer_numel(self):
        return int(lib().cunet_buffer_numel(self.h))

    @property
    def counter_numel(self):
        return int(lib().cunet_counter_numel(self.h))

    def workspace_bytes(self, training):
        """training: False / True, or 2 for inference plus the bf16 arena."""
        return int(lib().cunet_workspace_bytes(self.h, int(training)))

    @property
    def num_heads(self):
        return int(lib().cunet_num_heads(self.h))

    def anchors(self):
        buf = (C.c_int32 * 64)()
        n = check(lib().cunet_loss_anchors(self.h, buf, 64), 'cunet_loss_anchors')
        return [int(buf[i]) for i in range(n)]

    def buckets(self):
        """[(begin, count)] float ranges of the parameter/gradient arena; last one is the stem."""
        L = lib()
        out = []
        b, c = C.c_int64(), C.c_int64()
        for i in range(L.cunet_num_buckets(self.h)):
            check(L.cunet_bucket_range(self.h, i, C.byref(b), C.byref(c)), 'cunet_bucket_range')
            out.append((int(b.value), int(c.value)))
        return out

    # ---- per-kernel-class timing
    def profile_begin(self, mode: int, cls: int = -1):
        check(lib().cunet_profile_begin(self.h, mode, cls), 'cunet_profile_begin')

    def profile_reset(self):
        check(lib().cunet_profile_reset(self.h), 'cunet_profile_reset')

    def profile_collect(self, which: int = -1):
        """dict class name -> (launches, ms, algorithmic flops, algorithmic bytes); waits for the events.  which: -1 every launch,
        0 only launches on the caller's stream, 1 only launches on the library's internal side stream (cunet_profile_get_stream)."""
        L = lib()
        check(L.cunet_profile_collect(self.h), 'cunet_profile_collect')
        out = {}
        cnt, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        for i in range(L.cunet_profile_num_classes()):
            check(L.cunet_profile_get_stream(self.h, i, int(which), C.byref(cnt), C.byref(ms), C.byref(fl), C.byref(by)), 'cunet_profile_get_stream')
            out[L.cunet_profile_class_name(i).decode()] = (int(cnt.value), float(ms.value), float(fl.value), float(by.value))
        return out

    @staticmethod
    def profile_class_index(name: str) -> int:
        L = lib()
        for i in range(L.cunet_profile_num_classes()):
            if L.cunet_profile_class_name(i).decode() == name:
                return i
        raise KeyError(name)

    def set_quant_input(self, bits_i: int, ternary_convs=()):
        """include/cunet.h cunet_set_quant_input; returns the number of nodes whose forward runs on AND-popcount."""
        names = [n.encode() for n in ternary_convs]
        arr = (C.c_char_p * max(len(names), 1))(*names) if names else None
        return check(lib().cunet_set_quant_input(self.h, int(bits_i), arr, len(names)), 'cunet_set_quant_input')

    def bucket_order(self):
        buf = (C.c_int32 * 256)()
        n = check(lib().cunet_bucket_order(self.h, buf, 256), 'cunet_bucket_order')
        return [int(buf[i]) for i in range(n)]

    def describe(self):
        if self._desc is None:
            self._desc = json.loads(lib().cunet_plan_describe(self.h).decode())
        return self._desc

    def tensor_offset(self, name: str, which: int = 0) -> int:
        return int(lib().cunet_debug_tensor_offset(self.h, name.encode(), which))
