"""cu_net_amd -- MI355X-native (gfx950) Coupled U-Net hot path.

(The directory is named `cu_net_amd` rather than `cu-net_amd` because a hyphen is not importable.)

Public surface mirrors the reference (zhiqiangdon/CU-Net):
    create_cu_net(neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num)
plus the fused train step (`FusedTrainer`), data parallelism over RCCL (`cu_net_amd.parallel`),
the validation-loop pieces (`get_preds`, `final_preds`, `flip_merge`, `accuracy`), the weight quantisers (`cu_net_amd.quant`)
and the training-sample preparation (`cu_net_amd.augment`: flip, colour gain, crop / rotate / resize on the GPU).
"""
from ._lib import CUNetError, LIB_PATH  # noqa: F401
from .module import CUNet, create_cu_net  # noqa: F401
from .trainer import FusedTrainer, accuracy, accuracy_origin_res, final_preds, flip_merge, get_preds, pts2heatmap  # noqa: F401
from .augment import augment_batch, draw_train_params, prepare_batch, shufflelr, transform_pts  # noqa: F401
