/*
 * cunet.h -- C ABI of the MI355X-native CU-Net hot path (libcunet_hip.so).
 *
 * The reference (zhiqiangdon/CU-Net) is pure Python on top of torch.nn and has no FFI of its
 * own; this header is the boundary a maintainer would bind from Python (ctypes, see
 * INTEGRATION.md) to replace the body of
 *     models/cu_net.py:336-360   _CU_Net_Wrapper.forward
 *     cu-net.py:175-183          loss + backward + optimizer.step
 * while keeping `models/cu_net.py:362-368 create_cu_net(...)` as the user-facing surface.
 *
 * Conventions
 *   - plain C types only; every device buffer is CALLER-OWNED (torch-allocated) and passed as a
 *     raw device pointer; the plan owns no device memory.
 *   - all work is enqueued on the hipStream_t the caller passes (void* here); no hidden sync.
 *   - every function returns 0 on success or a negative cunet_status; cunet_last_error()
 *     returns a thread-local message for the last failure.
 *   - one plan per (config, batch, H, W); a plan is not thread-safe, distinct plans are independent.
 *   - activations are fp32; public image/heat-map tensors are NCHW like the reference's.
 */
#ifndef CUNET_H
#define CUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cunet_plan cunet_plan_t;

typedef enum {
    CUNET_OK = 0,
    CUNET_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    CUNET_ERR_STATE = -2,     /* call order violated (e.g. backward before forward) */
    CUNET_ERR_HIP = -3,       /* a HIP runtime call failed */
    CUNET_ERR_NOMEM = -4,
    CUNET_ERR_CALLBACK = -5   /* a caller-supplied callback reported failure (cunet_backward_ex) */
} cunet_status;

/* Mirrors create_cu_net(neck_size, growth_rate, init_chan_num, class_num, layer_num, order,
 * loss_num) -- models/cu_net.py:362-368 -- plus the input geometry the plan is specialised for. */
typedef struct {
    int32_t neck_size, growth_rate, init_chan_num, class_num, layer_num, order, loss_num;
    int32_t batch, height, width;   /* input N x 3 x height x width; height, width % 64 == 0 */
} cunet_cfg;

/* One state_dict entry (reference key order, models/cu_net.py:299-320). */
typedef struct {
    char name[160];
    int32_t kind;        /* 0 = parameter (float arena), 1 = running stat (buffer arena),
                            2 = num_batches_tracked (int64 counter arena) */
    int32_t ndim;
    int64_t shape[4];
    int64_t offset;      /* element offset inside its arena */
    int64_t numel;
} cunet_state_desc;

const char* cunet_last_error(void);
const char* cunet_version(void);

/* ---- plan lifetime (host only; touches no device) ------------------------------------------ */
int cunet_plan_create(const cunet_cfg* cfg, cunet_plan_t** out);
void cunet_plan_destroy(cunet_plan_t* plan);

/* ---- state layout: lets the host alias nn.Parameters onto flat arenas ----------------------- */
int cunet_state_count(const cunet_plan_t* plan);
int cunet_state_entry(const cunet_plan_t* plan, int index, cunet_state_desc* out);
int64_t cunet_param_numel(const cunet_plan_t* plan);     /* floats: parameter arena == gradient arena */
int64_t cunet_buffer_numel(const cunet_plan_t* plan);    /* floats: running_mean / running_var arena */
int64_t cunet_counter_numel(const cunet_plan_t* plan);   /* int64: num_batches_tracked arena */
int64_t cunet_workspace_bytes(const cunet_plan_t* plan, int training);  /* 0 inference, 1 training, 2 / 3: the same plus the bf16 activation arena */
int cunet_num_heads(const cunet_plan_t* plan);           /* == loss_num */
/* anchors[i] = 1-based U-Net index whose head produces output i (models/cu_net.py:275-283) */
int cunet_loss_anchors(const cunet_plan_t* plan, int32_t* anchors, int capacity);

/* JSON description of tensors and nodes (tests and tooling; valid until the plan is destroyed) */
const char* cunet_plan_describe(const cunet_plan_t* plan);

/* ---- binding caller-owned device memory ------------------------------------------------------
 * params/grads: cunet_param_numel floats each; buffers: cunet_buffer_numel floats;
 * counters: cunet_counter_numel int64; workspace: cunet_workspace_bytes bytes, 256-B aligned.
 * Uploads the plan's small device-side tables into the workspace on `stream`. */
int cunet_bind(cunet_plan_t* plan, float* params, float* grads, float* buffers, int64_t* counters,
               void* workspace, int64_t workspace_bytes, int training, void* stream);

/* ---- the hot path ---------------------------------------------------------------------------
 * forward: replaces _CU_Net_Wrapper.forward (models/cu_net.py:336-360).
 *   x      : N x 3 x H x W fp32 NCHW (device)
 *   heat[i]: N x class_num x H/4 x W/4 fp32 NCHW (device), i < loss_num; may be NULL to skip the copy-out
 *   training != 0: BatchNorm uses batch statistics and running stats / counters are updated
 *                  (once per BN; the reference's extra checkpoint-recompute update is applied
 *                  by cunet_backward, as in the reference where it happens inside backward()).
 * Stream semantics of the hot path: every call only ENQUEUES work and never waits for the device.  A training forward and the
 * backward also use a library-internal low-priority side stream (heat-map heads, skip adapters without a pair kernel, weight
 * gradients); before cunet_forward / cunet_backward return, `stream` has been made to wait for everything they put there, so the
 * caller only ever orders against / synchronises `stream` (cunet_backward_ex: see the bucket callback's contract below). */
int cunet_forward(cunet_plan_t* plan, const float* x, float* const* heat, int training, void* stream);

/* loss: replaces cu-net.py:175-178. Writes sum_k mean((out_k - target)^2) to *loss (device float)
 * and stages d(loss)/d(out_k) for cunet_backward. target is N x class_num x H/4 x W/4 NCHW. */
int cunet_loss_mse(cunet_plan_t* plan, const float* target, float* loss, void* stream);

/* The same loss FUSED into the next training forward: call before cunet_forward / cunet_forward_bf16 (training != 0) on the same
 * stream.  The target is staged now; the heads' convolution epilogues then write d(loss)/d(out_k) and sum the squared error while
 * the heat maps are still in registers (no separate pass over them), *loss is written at the end of that forward, and
 * cunet_backward(plan, NULL, ...) may follow directly.  One-shot: it arms exactly one forward.  Same numbers as cunet_loss_mse
 * up to the order of the fp64 partial sums. */
int cunet_loss_mse_fused(cunet_plan_t* plan, const float* target, float* loss, void* stream);

/* backward: replaces loss.backward() (cu-net.py:182) for this network.
 *   grad_heat: NULL -> use the gradients staged by cunet_loss_mse;
 *              else loss_num NCHW tensors d(loss)/d(heat[i]) (autograd path).
 *   Fills the bound gradient arena (overwrites; parameters that received no gradient stay 0). */
int cunet_backward(cunet_plan_t* plan, const float* const* grad_heat, void* stream);

/* bf16 activation storage (BASELINE config 3 direction; nothing like it in the reference, which is fp32):
 * same inputs / outputs as cunet_forward -- fp32 NCHW image in, loss_num fp32 NCHW heat maps out -- but activations
 * and the forward weight operands are held as bf16 between the stem and the heads and contracted with bf16 MFMA (fp32
 * accumulation, BatchNorm + ReLU in fp32).  training = 0: running statistics; needs a plan bound with training = 0 on
 * cunet_workspace_bytes(plan, 2) bytes.  training = 1: batch statistics of the bf16-rounded tensors; needs a training
 * bind on cunet_workspace_bytes(plan, 3) bytes, and cunet_loss_mse / cunet_backward after it read x as bf16 while
 * gradients, weights and the optimiser stay fp32.  training = 2: as 1, and the gradient TENSORS of backward (dY, dz, dX
 * of every node) are stored as bf16 too (parameter gradients, weights and the optimiser stay fp32; the loss gradient
 * must come from cunet_loss_mse).  Channel counts must be multiples of 32, rows of 32 at every level. */
int cunet_forward_bf16(cunet_plan_t* plan, const float* x, float* const* heat, int training, void* stream);

/* Gradient buckets for data parallelism.  The parameter/gradient arena is laid out bucket-major:
 * bucket i < layer_num holds every parameter used by U-Net index i, bucket layer_num the stem.
 * cunet_backward_ex calls on_bucket(b, user) on the calling thread right after the LAST kernel
 * writing bucket b has been enqueued (order: layer_num-1, ..., 0, stem), so the host can start that
 * bucket's RCCL all-reduce on another stream while the rest of backward runs.  Weight gradients are
 * enqueued on a library-internal side stream, the rest on `stream`: inside the callback the consumer
 * stream must (1) wait for an event recorded on `stream` and (2) call cunet_side_stream_join(plan,
 * consumer) -- `stream` itself is NOT made to wait for the weight gradients at a bucket boundary (that
 * stalled the data-gradient chain for 2.6 ms per step).  After cunet_backward_ex returns, `stream` has
 * joined the side stream.  If the callback returns non-zero (e.g. the collective could not be issued) no further
 * kernels are enqueued, `stream` joins the side stream and CUNET_ERR_CALLBACK is returned: the gradient arena is
 * then INCOMPLETE and must not reach the optimiser.  (Replaces the grad reduce inside torch.nn.DataParallel,
 * cu-net.py:59.) */
typedef int (*cunet_bucket_cb)(int bucket, void* user);   /* return 0 to continue; non-zero aborts backward */
int cunet_num_buckets(const cunet_plan_t* plan);
int cunet_bucket_range(const cunet_plan_t* plan, int bucket, int64_t* begin, int64_t* count);
int cunet_backward_ex(cunet_plan_t* plan, const float* const* grad_heat, void* stream,
                      cunet_bucket_cb on_bucket, void* user);
/* make `stream` wait for everything enqueued so far on the plan's internal side stream (no-op without one) */
int cunet_side_stream_join(cunet_plan_t* plan, void* stream);
/* the order in which cunet_backward_ex reports buckets (host only; returns the count) */
int cunet_bucket_order(const cunet_plan_t* plan, int32_t* order, int capacity);

/* fused RMSprop over a flat arena: torch.optim.RMSprop(lr, alpha, eps, momentum=0, weight_decay=0)
 * (cu-net.py:60-61,183). g is multiplied by grad_scale first (1/world_size under data parallelism).
 * Hyper-parameters are doubles, as torch receives them (python floats): 1 - alpha is formed in double before it is
 * rounded to fp32 -- (float)(1 - 0.99) and 1.f - 0.99f differ by 9e-7 relative, which would be a systematic bias of v. */
int cunet_rmsprop_step(float* params, const float* grads, float* square_avg, int64_t n,
                       double lr, double alpha, double eps, double grad_scale, void* stream);

/* argmax landmark decode: replaces pylib/Evaluation.py:6-23 get_preds.
 *   heat : N x K x H x W fp32 NCHW; preds: N x K x 2 fp32 (1-based x, y; 0,0 where max <= 0) */
int cunet_get_preds(const float* heat, float* preds, int n, int k, int h, int w, void* stream);

/* ---- weight / gradient quantisers on the flat arenas (utils/quantize.py:104-175 QuanOp; with
 * keep_scale = 1 and bits_g = 32: BinOp, models/cu_net_prev_version.py:45-92).  `table` is a DEVICE array
 * of nconv records {int64 offset; int32 O, I, KK, pad} describing the target conv weights [O][I][KK];
 * max_o / max_n are the largest O and I*KK in the table.  One launch per phase for all convs. */
int cunet_quant_prepare(float* params, float* saved, const void* table, int nconv, int max_o, int max_n,
                        int bits_w, int bits_g, int keep_scale, void* stream);
int cunet_quant_restore(float* params, const float* saved, const void* table, int nconv, void* stream);
int cunet_quant_grad(const float* params, float* grads, const void* table, int nconv, int max_o,
                     int bits_w, int bits_g, int keep_scale, void* stream);

/* ---- quantised-input mode of a plan: the QuanInput2d placement of the reference's quantised model
 * (models/cu_net_prev_version_wig.py:96-98 before every 3x3 conv, :277-279 before every head conv; utils/quantize.py:47-73)
 * on this network: an activation quantiser of bits_i bits (3..15) between the ReLU and those convs -- forward
 * Q(C(x, bits_i), bits_i), backward straight-through with no gradient where the activation is >= 1; the weight gradient
 * of such a conv contracts d(loss)/d(out) with the quantised activation.  bits_i = 0 switches the mode off.
 * ternary_convs: module paths (e.g. "hg.down_blocks.0.layers.0.conv2") of those convs whose weights the caller keeps in
 * {-1, 0, +1} across forward/backward (QuanOp with bits_w 1 or 2, utils/quantize.py:125-149): their FORWARD then runs on
 * the multiplier-free AND-popcount kernel (below) instead of MFMA -- bit-identical results, every partial sum being a
 * multiple of 2^-(bits_i-1).  Other quantised-input convs (e.g. the last head, which QuanOp leaves alone) stay on MFMA
 * with the quantiser folded into their operand loads.  fp32 plans only.  Returns the number of popcount nodes (>= 0). */
int cunet_set_quant_input(cunet_plan_t* plan, int bits_i, const char* const* ternary_convs, int n_ternary);
/* The AND-popcount forward packs only the SIGN planes of a weight: on weights that are not ternary it would silently compute a
 * sign(w) convolution.  It therefore runs only while the caller declares the listed convs' weights ternary: live = 1 between
 * QuanOp.quantization() and restore() (cu-net-prev-version-wig.py:165,189 in training, :230,285 around validation), live = 0
 * (the default after cunet_set_quant_input) otherwise -- those convs then take the MFMA path with the quantiser in their loads,
 * which is correct for any weights.  Host only; takes effect at the next cunet_forward. */
int cunet_set_popcount_live(cunet_plan_t* plan, int live);

/* ---- multiplier-free ternary convolution (AND + popcount over activation bit-planes): the non-MFMA
 * alternative for conv weights in {-1,0,+1} on bits_i-bit activations (QuanInput2d placement,
 * models/cu_net_prev_version_wig.py:96-98,277-279).  cunet_ternary_pack turns torch-layout weights
 * [O][C][taps] into two uint64 mask tensors of taps*ceil(C/64)*roundup(O,64) words each;
 * cunet_ternary_conv computes y[N*H*W][O] = conv(QuanInput(relu(x*scale+shift))) for NHWC x[N*H*W][C],
 * taps = 1 (1x1) or 9 (3x3, pad 1).  Exact: bit-identical to the fp32 convolution of the quantised input. */
int cunet_ternary_pack(const float* w, uint64_t* wpos, uint64_t* wneg, int o, int c, int taps, void* stream);
int cunet_ternary_conv(const float* x, const float* scale, const float* shift, const uint64_t* wpos,
                       const uint64_t* wneg, float* y, int n, int h, int w, int c, int o, int taps,
                       int bits_i, void* stream);
/* the same operator through the two-kernel path the network's forward uses: the activation is quantised and cut into bit-plane records
 * once (`planes`: (n*h*w + 1) * 16 uint64 of scratch, device), then counted by `variant` 1 = lane-per-pixel kernel with scalar masks
 * (round 5), 0 = wave-per-pixel kernel (round 3).  c <= 128, bits_i <= 8.  ystats (nullable): [2][o] fp64, zeroed by the caller --
 * receives sum(y), sum(y^2) per output channel, exact (integers / 2^(bits_i-1)). */
int cunet_ternary_conv_ex(const float* x, const float* scale, const float* shift, const uint64_t* wpos, const uint64_t* wneg,
                          uint64_t* planes, float* y, double* ystats, int n, int h, int w, int c, int o, int taps, int bits_i,
                          int variant, void* stream);

/* ---- per-kernel-class timing (bench.py roofline) ------------------------------------------------
 * HIP events are recorded on the launch stream around every launch of the selected class(es):
 * mode 0 = off, 1 = every class, 2 = only class `cls`.  cunet_profile_collect waits for the pending
 * events and accumulates; cunet_profile_get returns launches, total milliseconds, total ALGORITHMIC
 * flops (2*M*K*N*taps per GEMM-shaped launch) and algorithmic bytes of the class. */
int cunet_profile_begin(cunet_plan_t* plan, int mode, int cls);
int cunet_profile_reset(cunet_plan_t* plan);
int cunet_profile_collect(cunet_plan_t* plan);
int cunet_profile_num_classes(void);
const char* cunet_profile_class_name(int cls);
int cunet_profile_get(const cunet_plan_t* plan, int cls, int64_t* count, double* ms, double* flops, double* bytes);
/* the same split by the stream the launches ran on: which = 0 the caller's stream (the step's critical path), 1 the library's
 * internal low-priority side stream (weight gradients, heat-map heads of a training pass: a launch's duration there includes
 * the time it is switched out for the caller's kernels), -1 both (= cunet_profile_get). */
int cunet_profile_get_stream(const cunet_plan_t* plan, int cls, int which, int64_t* count, double* ms, double* flops, double* bytes);

/* full-resolution landmark decode: replaces pylib/Evaluation.py:108-132 final_preds for rot == 0
 * (quarter-pixel refinement, +0.5, inverse crop transform :152-187 with size 200, truncation).
 *   center: N x 2, scale: N (fp32, device); res0/res1: heat-map resolution bounds of the refinement test */
int cunet_final_preds(const float* heat, const float* center, const float* scale, float* preds, int n, int k,
                      int h, int w, int res0, int res1, void* stream);
/* the same decode for ANY crop transform, i.e. rot != 0 (pylib/Evaluation.py:163-178, the rotation branch of GetTransform):
 *   inv: N x 6 float64 (device) -- rows 0 and 1 of np.linalg.inv(GetTransform(center, scale, rot, res, 200)) per image, built by the
 *   host (cu_net_amd/trainer.py:_inverse_crop_transforms does it with the reference's numpy operations and dtypes);
 *   new = inv . [x - 1, y - 1, 1] in float64 (fused multiply-adds in k order), truncated toward zero, + 1 (:180-187). */
int cunet_final_preds_affine(const float* heat, const double* inv, float* preds, int n, int k, int h, int w, int res0, int res1,
                             void* stream);

/* flip test-time augmentation merge: replaces cu-net.py:247-249 + pylib/HumanAug.py:177-208
 * (flip_channels, shuffle_channels_for_horizontal_flipping), which the reference runs on the CPU:
 *   out[n][c][y][x] = (a[n][c][y][x] + b[n][perm[c]][y][w-1-x]) / 2
 * a = heat maps of the image, b = heat maps of the horizontally flipped image, perm = channel
 * permutation built from the flip index pairs (int32[k], device).  All N x K x H x W fp32 NCHW, device. */
int cunet_flip_merge(const float* a, const float* b, const int32_t* perm, float* out, int n, int k, int hh, int w,
                     void* stream);

/* training targets on the device: replaces pylib/HumanPts.py:35-76 (pts2heatmap / draw_gaussian), which the
 * reference's data loader runs per sample on the CPU.
 *   pts:   N*K x 2 float64 (x, y) heat-map coordinates; a point with x <= 0 or y <= 0 leaves its map zero
 *   patch: (2*half+1)^2 fp32 Gaussian patch, row-major (the caller tabulates exp(-(dx^2+dy^2)/half^2) once on the
 *          host, in float64 like the reference, so the maps are bit-identical to the reference's `.float()`)
 *   out:   N*K x H x W fp32, fully written */
int cunet_render_targets(const double* pts, const float* patch, int half, float* out, int nk, int hh, int w, void* stream);

/* Planner options (host only, process-wide; a plan takes a SNAPSHOT of them when it is created: later changes do not
 * alter the kernel selection of live plans).  They select between equivalent kernels and
 * never change results beyond summation order:
 *   "wgrad3_min_rows"    1x1 weight gradients of nodes with at least this many output rows (N*H*W) use the LDS-staged
 *                        atomics-free kernel, smaller ones the per-wave atomic kernel (default 0: every eligible node --
 *                        measured best on MI355X; tests also run with a large value to keep the other kernel covered)
 *   "wgrad3_min_chunks"  at least this many 32-pixel chunks per workgroup of that kernel (default 2)
 *   "wgrad3_max_splits"  at most this many workgroups (= partial tiles) per launch; default 0 = 192 with f32_split (measured on the CU-Net-2 step:
 *                        3906-3942 img/s at 256, 3983-4017 at 224, 4016-4024 at 192, 3994-4027 at 160, 3947-3966 at 128), 256 on the fp32 matrix pipe
 *   "wgrad3_min_chunks_bf16", "wgrad3_max_splits_bf16"   the same two with bf16 gradient tensors, where these kernels are
 *                        HBM-bound and fewer, longer splits win (defaults 4 and 128; 96 before the LDS-DMA kernel of round 4)
 *   "wgrad3_stem"        1 (default): the stem's 7x7 weight gradient on the LDS-staged atomics-free kernel where the shape
 *                        allows (128 output channels, output width a multiple of 64); 0: per-wave atomic kernel
 *   "conv3x3_ring_min_rows"  the 3x3 forward of 64-pixel-wide levels runs on the LDS row ring when the batch has at least this many
 *                        image rows N*H (default 512 = two per CU; tests lower it to cover the kernel at small batches)
 *   "wgrad_fork_group"   backward hands the weight gradients to the internal side stream in groups of this many nodes (default 0 = by depth:
 *                        2 for layer_num <= 4, 4 beyond -- round 5: CU-Net-2 4243-4298 img/s at 2 vs 4188-4206 at 4; CU-Net-16 578 vs 581):
 *                        every hand-over is an event record on the caller's stream, i.e. a marker packet the next kernel waits
 *                        behind (6-7 us of bubble on the critical path each); 1 = one hand-over per node
 *   "wgrad_fork_group_bf16"  the same when the gradient tensors are stored as bf16 (default 8)
 *   "fwd_fork_min_w"     forward: the down blocks' skip adapters run on the side stream at levels at least this wide (default 0: all)
 *   "pair_adapters"      1 (default): the ahead and the skip adapter of a down block (two 1x1 convolutions over the same concat,
 *                        models/cu_net.py:139-142) share ONE launch, forward and data gradient, where the shape has a pair kernel;
 *                        0: one launch each (forward: the skip adapter on the side stream)
 *   "heads_on_side"      1: in a training pass the heat-map heads (forward with the fused loss, data and weight gradient)
 *                        run on the internal side stream -- nothing on the caller's stream reads a head's output before the loss
 *                        is finalised, and its backward depends on the loss gradient only; the caller's stream waits for a head's DATA
 *                        gradient where the head's turn would be, the heads' weight gradients follow behind all of them.  2 (default): as
 *                        1, but the last U-Net's head -- whose data gradient the caller's stream needs before anything else of backward --
 *                        runs that data gradient on the caller's stream (+1 % on the CU-Net-2 step).  0: in node order on the caller's stream
 *   "dgrad_nt"           fp32 1x1 data gradient: a wave owns up to this many 32-channel tiles of dz for its 32 rows (default 4): dY is
 *                        read once per dgrad_nt * 32 output channels and each of its fragments feeds that many independent MFMA
 *                        accumulator chains; 1 = one tile per wave (rounds 2-3: dY re-read by every 32-channel slice)
 *   "stem_split"         1 (default; with f32_split): the stem's 7x7 / 2 convolution walks output rows over an LDS ring of input rows, every
 *                        image element cut into its bf16 pieces once, the contraction's k re-ordered to (channel, kernel row) x 8 consecutive
 *                        pixels (stem_fwd_split_kernel): 206 -> 106 us per launch at batch 24 (CU-Net-2 step +1.6 %, eval forward +6 %);
 *                        0: the im2col-gather kernel on the fp32 matrix pipe.  Also used by the bf16 storage modes (their stem is fp32)
 *   "dgrad3_ring"        fp32 3x3 data gradient on the split contraction: launches over at least this many image rows (N * H; default 768 = the
 *                        64 x 64 and 32 x 32 levels at batch 24; 0 = never) walk image rows of a 32-pixel strip with dY in an LDS ring, every
 *                        element cut into its bf16 pieces once (dgrad3x3_ring_split_kernel) instead of gathering nine shifted taps per
 *                        column slice: class 0.55 -> 0.50 ms per CU-Net-2 step alone, step +1.0 ... 1.7 %
 *   "dgrad3_nt"          fp32 3x3 data gradient: 32-channel tiles of dz a wave owns per row tile (1 = default, 2; 4 does not fit the LDS next to
 *                        the operand's three planes): the nine shifted taps of dY gathered (and cut) once per 64 output channels instead of
 *                        once per 32.  Measured: class 0.540 -> 0.526 ms per CU-Net-2 step alone, step +-0
 *   "dgrad_rows"         fp32 1x1 data gradient of a 128-output-channel node: launches with at least this many 32-row tiles run the row-tile
 *                        kernel -- every input channel of a row tile in one workgroup, weights in registers, dY staged once per workgroup by
 *                        LDS-DMA instead of re-read by every 32-channel slice; 0 = never.  Default (-1, the one negative value accepted): 3072 with f32_split (the tile is cut into
 *                        its bf16 operand planes once per workgroup instead of once per column slice: +0.7 ... 1.7 % on the CU-Net-2 step when
 *                        only the 64 x 64 launches of batch 24 take it), 0 on the fp32 matrix pipe (measured equal alone, 1.60 vs 1.57 ms per
 *                        CU-Net-2 step, and 2.3 % slower in the step, where its nodes cannot share a launch as adapter pairs do)
 *   "dgrad_rows_v"       which row-tile kernel "dgrad_rows" selects on the split contraction: 2 (default, round 5) = the kernel owns its
 *                        vector-memory queue -- x by asm requests, ONE counted wait per tile that leaves the previous tile's dz stores in
 *                        flight, operand planes read one step ahead, BatchNorm sums in registers; 1 = the round-4 kernel (two vmcnt(0)
 *                        per tile).  Same arithmetic per element; the fp64 BatchNorm sums are added in another order
 *   "popcount_pixels"    AND-popcount forward of the quantised-input mode (cunet_set_quant_input + cunet_set_popcount_live): 1 (default, round 5)
 *                        = ternary_conv_pixels_kernel, lane = pixel with the weight masks as scalar operands and one mask per weight word
 *                        (2 popc(P & x) - popc(x) + popc(Z & x)); 0 = ternary_conv_planes_kernel of round 3 (wave = pixel, v_readlane per plane
 *                        word).  Integer arithmetic either way: bit-identical outputs and statistics
 *   "stem_fuse_dz"       1 (default, round 5): the stem's weight gradient computes d(loss)/d(conv0 output) itself while it stages its chunks --
 *                        from conv0's output, the gradient of the pooled features and the reductions of the first stem pass -- so the second pass
 *                        of the stem's BatchNorm-ReLU-pool backward and its 200 MB gradient tensor (24 x 128 x 128 x 128 fp32) never exist in a
 *                        training step; 0: two stem passes + the weight gradient reading that tensor (rounds 1-4).  Bit-identical dW.
 *                        (cunet_debug_run_node_backward always runs the unfused kernels: its tensors are real)
 *   "stem_wgrad_split"   1 (default, round 5; acts with f32_split): the stem's LDS-staged weight gradient contracts on the bf16 matrix pipe with
 *                        three-piece operands like every other fp32 convolution (it was the last kernel on the fp32 pipe and the last long
 *                        kernel of a step); 0: v_mfma_f32_32x32x2_f32
 *   "fuse_pool_gather"   1 (default, round 6): in backward, gather(pool output) -> pool backward and the gather of the skip adapter's output --
 *                        three element-wise launches in front of a down block's adapter pair, the last independent of the other two -- run as
 *                        ONE launch (gather_pool_pair_kernel): +1.7 ... 2.4 % on the CU-Net-2 step.  Same operations on the same values
 *   "fuse_z_gather"      0 (default; round 6, bf16 gradient tensors only): 1 = the gather of a tensor with ONE plain consumer (a dense layer's
 *                        bottleneck output) is folded into the operand load of the 1x1 data gradient that reads it, which also writes the
 *                        tensor.  Measured -2.5 % on config 3 (every column slice repeats the assembly): kept for re-measurement only
 *   "wgrad_split_planes" 0 (default; round 6, acts with f32_split): 1 = the fp32 1x1 weight gradient cuts its operands ONCE, on the way into LDS (three
 *                        bf16 planes, fragments by ds_read_b64_tr_b16: wgrad5_split_kernel) for slices of at most 8 channel tiles.  Bit-identical
 *                        partial tiles; 11-16 % less time for those launches alone, +-0.3 % on the CU-Net-2 step, +0.75 % on CU-Net-16
 *   "stem_wgrad_planes"  1 (default; round 6, acts with f32_split + stem_wgrad_split): the stem's weight gradient cuts BOTH operands once on their way
 *                        into LDS (image rows as three bf16 planes de-interleaved by column parity in a ring of 16 row slots, the dY chunk as three
 *                        planes read by ds_read_b64_tr_b16), splits its workgroup into four consumer waves that only request fragments and multiply and
 *                        four producer waves that only load / compute dz / cut, and walks its chunks by pooling-window row pair (conv0's output is read
 *                        once instead of twice): wgrad3_stem_planes_kernel, 161 -> 112 us alone at the end of every step, +0.65 % on the CU-Net-2 step.
 *                        Same pieces and products as wgrad3_stem_kernel<*, true>; the pixels enter the fp32 accumulators in another order (agreement
 *                        ~3e-7 of the gradient's magnitude, tests/test_gpu_exact.py).  0 = that kernel (also the fallback for IW > 340)
 *   "stem_wgrad_caller"  0 (default; round 6): 1 = the stem's weight gradient and its reduce on the caller's stream behind the stem's BatchNorm
 *                        pass instead of behind the last bucket's work on the side stream.  Measured +0.1 ... 0.2 %
 *   "dgrad_prefetch"     fp32 1x1 data gradient over 128 output channels (every bottleneck / adapter), one channel tile per wave: 2 = two
 *                        32-channel chunks of dY on the way per wave, requested across the tile boundary; 1 (default) = one (rounds 1-3):
 *                        measured equal (3504 vs 3491 img/s)
 *   "f32_split"          1: the fp32 convolutions contract on the bf16 matrix pipe.  Every fp32 operand value is cut into three bf16 pieces
 *                        x = h + m + l (8 + 8 + 8 significand bits: exact) and a product is hh' + hm' + mh' + mm' + hl' + lh', six
 *                        v_mfma_f32_32x32x16_bf16 (192 cycles per 16 k) instead of eight v_mfma_f32_32x32x2_f32 (512 cycles), accumulated in
 *                        fp32 as before; inputs, outputs, statistics and every tensor in HBM stay fp32.  The dropped terms are below 2^-24
 *                        of |x y|: measured against fp64 the result is as close as the fp32 MFMA's (profiles/r04_split_bf16_probe.txt;
 *                        tests/test_gpu_exact.py::test_backward_error_vs_fp64_tracks_torch_fp32[*-1]).  Kernels: 1x1 forward, heads, 1x1 and
 *                        3x3 data gradients (conv_body XBG = 6 / 7), 3x3 forward on the row ring, weight gradients, stem.  1 (DEFAULT since round 4).
 *                        0: the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; rounds 1-3).  Limitation of 1: an operand that is +-Inf, NaN or
 *                        finite above the bf16 maximum (3.39e38) contributes NaN (h = Inf, x - h = NaN) where the fp32 MFMA would give
 *                        +-Inf; a diverged run shows as NaN loss instead of Inf loss.  Finite values below that are exact
 *   "fuse_wgrad"         1: the fp32 data gradient of a 1x1 node (128 output channels) also computes the node's weight gradient from the dY
 *                        and x tiles it holds and writes one partial tile per row block (summed by the bucket's reduce): one pass over dY
 *                        and x per node, no wgrad launch on the side stream.  0 (default): separate launches -- measured 4.8 % faster in
 *                        the overlapped CU-Net-2 step on MI355X (the fused kernel is 1.9 % faster when nothing overlaps)
 *   "wgrad_bf16_dma"     1 (default): the 1x1 weight gradient of the bf16 storage mode streams dY and x into an LDS ring by LDS-DMA
 *                        (global_load_lds) and takes its MFMA operands with the LDS transpose read where every pixel range is whole
 *                        32-pixel slots; 0: always the register-staged kernel of rounds 2-3 (bit-identical results).  Snapshotted
 *                        into the plan like every other option (round 5); cunet_debug_set_plan_option flips it on a live plan
 * Returns 0, or CUNET_ERR_INVALID for an unknown name / negative value. */
int cunet_set_planner_option(const char* name, int value);
/* reads the process-wide value back (tests save / restore the options they change).  0, or CUNET_ERR_INVALID for an unknown name. */
int cunet_get_planner_option(const char* name, int* value);
/* debugging aid of the test-suite: changes "wgrad_bf16_dma", "wgrad_split_planes" or "stem_wgrad_planes" -- the options that only pick between
 * interchangeable kernels (same operands, same partial-tile layout) at launch time -- in the snapshot of a LIVE plan.  Any other name: CUNET_ERR_INVALID (those options shaped the plan's layout). */
int cunet_debug_set_plan_option(cunet_plan_t* plan, const char* name, int value);
/* debugging aid: tensors that a training step no longer materialises are written now, from the state the last cunet_backward left in the
 * workspace (today: d(loss)/d(conv0 output) under planner option "stem_fuse_dz").  The Python binding calls it before it reads a gradient
 * tensor for the test-suite; a no-op when nothing is missing. */
int cunet_debug_materialise(cunet_plan_t* plan, void* stream);

/* training-sample preparation on the device: replaces the per-sample CPU work of data/mpii_for_mpii_22.py:127-141 between the
 * decoded image and the network input -- horizontal flip (pylib/HumanAug.py:267-271), per-channel colour gain with clamp to
 * [0, 1], and HumanAug.crop (:115-172) INCLUDING its resamplers: scipy.misc.imresize / imrotate were 8-bit PIL operations behind
 * scipy's byte-scale (a contrast stretch by the minimum / maximum of the array), so crop()'s stages are kept as stages with uint8
 * intermediates and PIL's arithmetic restated exactly (triangle-filter resize in 22-bit fixed point, bilinear rotate truncated to
 * uint8): outputs are bit-identical to the reference function executed over PIL (tests/golden/G16_crop.npz).
 *   table_dev / table_host: the same array of n 192-byte records, on the device (read by the kernels) and on the host (read
 *          for launch geometry): {const float* src (3 x sh x sw fp32 CHW in [0,1]); uint32* mm (5 words scratch); uint8* i8, t1,
 *          i1 (pre-shrink intermediates or NULL), c8 (canvas ch x cw x 3), r8 (rotated: win_h x win_w x 3, else NULL),
 *          t2 (win_h x res x 3), o8 (res x res x 3); double rm[6] (PIL's rotate matrix); int32 sh, sw, sh1, sw1, ulx, uly, cw,
 *          ch, win_w, win_h, pad, flip, rotated, pre; float gain[3]; int32 pad}.  All scratch is caller-owned; the crop geometry
 *          is computed by the caller exactly as the reference computes it (cu_net_amd/augment.py::_geometry)
 *   out:   n x 3 x res x res fp32 (uint8 / 255, utils/imutils.py:31-36) */
int cunet_augment_batch(const void* table_dev, const void* table_host, int n, float* out, int res, void* stream);

/* ---- introspection for tests ------------------------------------------------------------------
 * byte offset inside the workspace of a named tensor's activation (which=0) or gradient (which=1);
 * negative if unknown. Names are those listed by cunet_plan_describe. */
int64_t cunet_debug_tensor_offset(const cunet_plan_t* plan, const char* name, int which);
/* Runs the backward of ONE node (index into the describe() node list) in isolation: clears the
 * node's reduction scratch and the whole gradient arena, then writes (never accumulates) the input
 * gradients.  The caller pokes d(loss)/d(output) into the workspace first.  Kernel unit tests.
 * An adapter of a pair ("pair": 1 in the node list, and the node after it) runs its data gradient as cunet_backward does, in
 * the pair's launch when the shape has a pair kernel; the partner's results are not read. */
int cunet_debug_run_node_backward(cunet_plan_t* plan, int node, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CUNET_H */
