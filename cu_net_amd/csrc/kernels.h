// Host-side launchers of the HIP kernels (definitions in *_kernels.hip).
#pragma once
#include "common.h"

namespace cunet {

enum WgLoadSel { WGL_SEG = 0, WGL_3X3 = 1, WGL_STEM = 2 };

hipError_t launch_conv(const ConvArgs& a, int load, int epi, int num_cus, hipStream_t s);
// two convolutions of one shape in ONE launch (the ahead / skip adapters of a down block, forward and data gradient);
// hipErrorNotSupported with nothing launched when the shapes differ or the shape has no pair kernel
hipError_t launch_conv_pair(const ConvArgs& a, const ConvArgs& b, int load, int epi, int num_cus, hipStream_t s);
int conv_fused_wgrad_splits(const ConvArgs& a, bool pair, int num_cus);      // > 0: partial tiles a fused data + weight gradient launch writes
hipError_t launch_wgrad(WgradArgs a, int load, int num_cus, hipStream_t s);
bool wgrad3_supported(const WgradArgs& a);
hipError_t launch_wgrad3(const WgradArgs& a, float* part, int S, int rows_per_split, hipStream_t s);
size_t wgrad3_stem_lds_bytes(int IW, int rows);
bool wgrad3_stem_supported(const WgradArgs& a, int rows);
hipError_t launch_wgrad3_stem(const WgradArgs& a, float* part, int wpi, int rows, hipStream_t s);
bool wgrad3_3x3_supported(const WgradArgs& a);
bool wgrad3_3x3_on_bf16_mfma(const WgradArgs& a);      // bf16 x and dY, W in {16, 32, 64}
hipError_t launch_wgrad3_3x3(const WgradArgs& a, float* part, int S, int rows_per_split, hipStream_t s);
hipError_t launch_wgrad_reduce(const WgReduceEntry* tab, int n, int max_numel, const float* ws, float* grads, hipStream_t s);
hipError_t launch_grad_gather(const GradGatherArgs& a, int num_cus, hipStream_t s);
hipError_t launch_gather_pool_pair(const GradGatherArgs& a, const GradGatherArgs& b, const PoolArgs& pool, int num_cus, hipStream_t s);      // hipErrorNotSupported: nothing launched
hipError_t launch_bn_param_grad(const BnParamGradArgs& a, hipStream_t s);
hipError_t launch_pool_fwd(const PoolArgs& a, int mode, int num_cus, hipStream_t s);
hipError_t launch_pool_bwd(const PoolArgs& a, int num_cus, hipStream_t s);
hipError_t launch_stem_bwd(const PoolArgs& a, int pass, float* dgamma, float* dbeta, int num_cus, hipStream_t s);
hipError_t launch_transpose(const float* src, float* dst, int N, int C, int HW, int ld, int to_nhwc, hipStream_t s);
hipError_t launch_mse(const float* out, const float* tgt, float* dout, double* loss_acc, long rows, int C, int ld, int ldd,
                      int grad_bf16, int num_cus, hipStream_t s);
hipError_t launch_loss_finalize(const double* acc, float* loss, hipStream_t s);
hipError_t launch_running_update(const RunStatEntry* tab, int n, const double* stats_base, float* buffers,
                                 int64_t* counters, int mode, hipStream_t s);
hipError_t launch_repack(const RepackEntry* tab, int n, const float* params, float* ws, hipStream_t s);
hipError_t launch_rmsprop(float* p, const float* g, float* v, long n, float lr, float alpha, float one_minus_alpha, float eps,
                          float gscale, hipStream_t s);
hipError_t launch_cvt_bf16(const float* src, void* dst, double* ystats, long rows, int C, int num_cus, hipStream_t s);
hipError_t launch_repack_bf16(const RepackEntry* tab, int n, const float* params, void* arena, int with_backward, hipStream_t s);
hipError_t launch_dgrad_bf16(const ConvArgs& a, int num_cus, hipStream_t s);
hipError_t launch_dgrad_bf16_pair(const ConvArgs& a, const ConvArgs& b, int num_cus, hipStream_t s);      // as launch_conv_pair
hipError_t launch_pool_bf16(const void* x, void* y, double* ystats, int N, int H, int W, int C, int num_cus, hipStream_t s);
hipError_t launch_conv_bf16(const ConvArgs& a, int out_f32, int num_cus, hipStream_t s);
hipError_t launch_conv_bf16_pair(const ConvArgs& a, const ConvArgs& b, int num_cus, hipStream_t s);       // as launch_conv_pair (bf16 out)
hipError_t launch_render_targets(const double* pts, const float* patch, int half, float* out, int NK, int H, int W, hipStream_t s);
hipError_t launch_flip_merge(const float* a, const float* b, const int* perm, float* out, int N, int K, int H, int W, hipStream_t s);
hipError_t launch_final_preds(const float* heat, const float* center, const float* scale, const double* inv, float* preds, int N, int K,
                              int H, int W, int res0, int res1, hipStream_t s);
hipError_t launch_augment(const AugSample* tab_dev, const AugSample* tab_host, int n, float* out, int res, hipStream_t s);
hipError_t launch_get_preds(const float* heat, float* preds, int maps, int H, int W, hipStream_t s);

hipError_t launch_quant_prepare(const QuantEntry* tab, int nconv, int maxO, int maxN, float* params, float* saved,
                                int bits_w, int bits_g, int keep_scale, hipStream_t s);
hipError_t launch_quant_restore(const QuantEntry* tab, int nconv, float* params, const float* saved, hipStream_t s);
hipError_t launch_quant_grad(const QuantEntry* tab, int nconv, int maxO, const float* params, float* grads,
                             int bits_w, int bits_g, int keep_scale, hipStream_t s);
hipError_t launch_ternary_pack(const float* w, uint64_t* wpos, uint64_t* wneg, int O, int C, int taps, int Opad, hipStream_t s);
hipError_t launch_ternary_conv(const TernArgs& a, int num_cus, hipStream_t s);
hipError_t launch_ternary_pack_all(const TernPackEntry* tab, int n, const float* params, uint64_t* masks, hipStream_t s);

}  // namespace cunet
