"""Kernel name -> bench.py class name (`roofline.kernel`), shared by tools/pmc_traffic.py and tools/pmc_mfma_busy.py."""
import re

CLASSES = [      # first match wins: the bf16-MFMA kernels carry bench.py's *_bf16 class names
    (r'wgrad3_stem_kernel|wgrad3_stem_planes_kernel', 'stem_bwd_weight'),
    (r'wgrad3_3x3_bf16_kernel', 'conv3x3_bwd_weight_bf16'),
    (r'wgrad3_bf16_kernel|wgrad4_bf16_kernel', 'conv1x1_bwd_weight_bf16'),
    (r'wgrad3_3x3_kernel', 'conv3x3_bwd_weight'),
    (r'wgrad3_kernel', 'conv1x1_bwd_weight'),
    (r'wgrad_reduce_kernel', 'wgrad_partial_reduce'),
    (r'wgrad2_stem_kernel', 'stem_bwd_weight'),
    (r'wgrad2_kernel', 'conv1x1_bwd_weight'),
    (r'wgrad_kernel<\(?cunet::\)?1|wgrad_kernel<1,', 'conv3x3_bwd_weight'),
    (r'wgrad_kernel<2,', 'stem_bwd_weight'),
    (r'dgrad_bf16(_pair)?_kernel<1,', 'conv1x1_bwd_data_bf16'),
    (r'dgrad_bf16(_pair)?_kernel<9,', 'conv3x3_bwd_data_bf16'),
    (r'conv3x3_tapsplit_bf16_kernel|conv3x3_ring_bf16_kernel|conv_bf16_kernel<9,', 'conv3x3_fwd_bf16'),
    (r'conv_bf16(_pair)?_kernel<1,', 'conv1x1_fwd_bf16'),
    (r'conv3x3_tapsplit_kernel|conv3x3_ring_kernel|conv3x3_ring_split_kernel', 'conv3x3_fwd'),
    (r'conv(_pair)?_kernel<0, 0,|conv1x1_splitk(_pair)?_kernel', 'conv1x1_fwd'),
    (r'conv_kernel<1, 0,', 'conv3x3_fwd'),
    (r'conv_kernel<4, 0,|stem_fwd_split_kernel', 'stem_conv_fwd'),
    (r'conv(_pair)?_kernel<2, 1,|dgrad1x1_rows_kernel|dgrad1x1_rows_split2?_kernel', 'conv1x1_bwd_data'),
    (r'conv_kernel<3, 1,|dgrad3x3_ring_split_kernel', 'conv3x3_bwd_data'),
    (r'grad_gather(_rows)?_kernel', 'bn_bwd_apply'),
    (r'gather_pool_pair_kernel', 'bn_bwd_apply'),      # round 6: gather + pool backward + the skip adapter's gather in one launch
    (r'pool_fwd_kernel<0>|pool_bf16_kernel', 'pool_fwd'),
    (r'pool_bwd_kernel', 'pool_bwd'),
    (r'ternary_conv_planes_kernel|ternary_conv_pixels_kernel|ternary_planes_kernel|ternary_conv_kernel', 'conv_fwd_popcount'),
]


def classify(name):
    for pat, cls in CLASSES:
        if re.search(pat, name):
            return cls
    return None
