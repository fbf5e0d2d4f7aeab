// Shared device/host structures of the CU-Net HIP path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace cunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int MAXSEG = 8;        // segments of one virtual concat (2 inputs + order carried + 1 new)
constexpr int WAVE = 64;
constexpr float BN_EPS = 1e-5f;  // nn.BatchNorm2d default (models/cu_net.py:22,41,45,195,301)

// One source tensor of a virtual channel concat (models/cu_net.py:13 torch.cat is never
// materialised: consumers read the segments in place).
struct Seg {
    const float* x;        // activation [rows][ld]
    float* gx;             // gradient   [rows][ld]
    const double* stats;   // [2][C]: sum, sum of squares over the tensor's rows
    double count;          // rows of the source tensor (BN sample count per channel)
    int C;                 // channels taken from this tensor (all of it)
    int ld;                // row pitch in floats
    int ups;               // 1: tensor is at half resolution, read through nearest-upsample (cu_net.py:250,265)
    int pad0_;
    int choff;             // channel offset inside the concat
    int pad_;
};

// Loader / epilogue selectors of conv_kernel
enum ConvLoad { LD_SEG = 0, LD_3X3 = 1, LD_PLAIN = 2, LD_PLAIN3 = 3, LD_STEM = 4 };
enum ConvEpi { EP_FWD = 0, EP_BWD = 1 };

struct ConvArgs {
    // ---- concat description: A operand for LD_SEG/LD_3X3 (forward), X of the BN being
    //      differentiated for EP_BWD
    int nseg;
    int Ccat;              // total channels of the concat
    Seg seg[MAXSEG];
    const float* gamma;    // BN over the concat
    const float* beta;
    const float* rmean;
    const float* rvar;
    int training;
    // ---- plain A operand (LD_PLAIN / LD_PLAIN3: backward-data)
    const float* a;
    int lda;
    // ---- contraction
    int K;                 // channels contracted per tap
    int taps;              // 1 or 9
    const float* wB;       // repacked weights [taps][Kpad/4][Npad][4]
    int Kpad, Npad;
    // ---- output
    float* y;
    int ldy;
    int Nout;              // valid output channels
    double* ystats;        // EP_FWD: [2][Nout] or null; EP_BWD: [2][Nout] sum dz, sum dz*xhat
    // ---- geometry of the output rows
    int M, H, W;
    // ---- stem (LD_STEM): NCHW image
    const float* img;
    int IH, IW;
    int xbf16;             // EP_BWD: 1 = the segments' x are bf16; 2 = x, the A operand (dY) and the dz output are bf16 (wB stays fp32)
    int qin_bits;          // > 0: QuanInput2d of that many bits sits between the ReLU and the conv (utils/quantize.py:47-73, placement
                           // models/cu_net_prev_version_wig.py:96-98,277-279): forward loaders quantise the activation,
                           // EP_BWD applies its straight-through mask (no gradient where the activation is >= 1)
    int wshift, hwshift;   // log2(W), log2(H*W) when both are powers of two (every level of a 2^k x 2^k input), else -1: the kernels
                           // split a row index with shifts instead of two emulated divisions per tile and per epilogue row group
    int any_ups;           // some segment is read through the nearest-upsample map
    int xcd_gx, xcd_gy;    // > 0: the grid is 1-D (8 * ceil(gx / 8) * gy blocks) and decoded so that the gy column-slice blocks of a row
                           // block are consecutive workgroups of ONE XCD (round-robin dispatch: id % 8): they share A through its L2
    // ---- fused pixelwise MSE of a heat-map head (EP_FWD, cu-net.py:175-178): with mse_tgt != null the epilogue also writes
    //      d(loss)/d(out) = 2 (out - target) / numel and adds sum((out - target)^2) / numel to *mse_acc
    const float* mse_tgt;  // target [M][ldy] (NHWC like y)
    float* mse_dout;       // d(loss)/d(out) [M][ldy], fp32 or (mse_gbf16) bf16; pad columns Nout..ldy-1 are zeroed
    double* mse_acc;
    double mse_inv;        // 1 / (M * Nout)
    int mse_gbf16;
    int mse_ldd;           // leading dimension of mse_dout (elements): ldy, or the padded K of a bf16 head gradient (see head_grad_ld)
    int col_slices;        // set by the launcher: column slices of the launch (gridDim.y, or xcd_gy on the 1-D grid).  The fp32 data gradient
                           // with the LDS-tile epilogue deals the 32-column tiles of the output to the slices as evenly as they go:
    int sl_rem, sl_gx_big, sl_gx_small;      // its first sl_rem slices carry one tile more than the others and get sl_gx_big row blocks each,
                           // the others sl_gx_small -- row blocks in proportion to the tiles, so that every block does the same work
                           // (sl_gx_small > 0 selects this decoding of the 1-D grid, see conv_body)
    float* wg_part;        // EP_BWD, fp32, 1x1: non-null = also compute this node's weight gradient (conv_body's fused tile loop) and store the
                           // block's partial tile into wg_part[row block][K][Nout]; the bucket's reduce kernel sums the row blocks
    int dgrad_rows;        // EP_BWD, fp32 1x1 over K = 128: > 0 = launches with at least this many 32-row tiles run dgrad1x1_rows_kernel (all columns
                           // of a row tile in one workgroup, dY staged once by LDS-DMA); 0 = always the column-sliced kernel (planner option dgrad_rows)
    int dgrad_rows_v;      // which row-tile kernel on the split contraction: 2 = dgrad1x1_rows_split2_kernel (round 5: counted waits, pipelined plane
                           // reads), 1 = dgrad1x1_rows_split_kernel (round 4) -- planner option dgrad_rows_v
    int dgrad_prefetch;    // EP_BWD, fp32 1x1 over K = 128, one channel tile per wave: 2 = two chunks of dY in flight per wave (conv_body's PF2 loop),
                           // else one (the plan's snapshot of planner option dgrad_prefetch)
    int dgrad3_ring;       // EP_BWD, fp32 3x3 on the split contraction: > 0 = launches over at least this many image rows run dgrad3x3_ring_split_kernel
                           // (planner option dgrad3_ring)
    int dgrad3_nt;         // EP_BWD, fp32 3x3: 32-column tiles of dz a wave owns per row tile (planner option dgrad3_nt; 0 / 1 = one)
    int dgrad_nt;          // EP_BWD, fp32: most 32-column tiles of dz a wave owns per row tile (the plan's snapshot of planner option
                           // dgrad_nt; 0 = the default 4, 1 = one tile per wave as in rounds 2-3)
    int split;             // fp32 operands, nothing ragged: 1 = contract on the bf16 matrix pipe with every operand value cut into three bf16
                           // pieces, six products (conv_body's XBG = 6 / 7; the plan's snapshot of planner option f32_split)
    int ring_min_rows;     // 3x3 forward: LDS row ring when the batch has at least this many image rows (the plan's snapshot of
                           // planner option conv3x3_ring_min_rows; 0 = the default 512)
    // ---- round 6, EP_BWD 1x1 over K = 128 (planner option fuse_z_gather): the A operand is NOT read from `a`.  `a` is the gradient of a tensor
    //      with ONE plain consumer (the bottleneck output z of a dense layer, read by its 3x3 conv only), and this launch applies that
    //      consumer's BatchNorm backward on the load -- a[m][k] = fz_A[k] * fz_dz[m][fz_choff + k] + fz_E[k] - fz_D[k] * fz_x[m][k], what
    //      grad_gather_rows_kernel<1, 0> would have written -- from coefficient tables it derives in its prologue; the column slice 0
    //      blocks store the operand into `a` as they go (the node's weight gradient and the tests read it afterwards).  fz_dz == null: off.
    const float* fz_dz;    // the consumer's masked data gradient [M][fz_lddz] (bf16 behind the pointer when xbf16 == 2)
    const double* fz_red;  // the consumer's reductions [2][fz_lddz]
    const float* fz_gamma; // the consumer's BatchNorm weight [fz_lddz]
    const float* fz_x;     // the tensor itself [M][fz_ldx] (bf16 when xbf16 != 0)
    const double* fz_stats;// its batch statistics [2][K]
    double fz_count;
    int fz_lddz, fz_choff, fz_ldx, fz_pad_;
    int dbg;               // timing experiments only (CUNET_CONV_DBG; conv_bf16_kernel: CUNET_B16_DBG = 32 / 64 as below, 2048 no output
                           // stores -- switches inside its chunk loop made the tuning build's kernel 3x slower and were removed):
                           // 1 no stats atomics, 4 no MFMA, 32 no B preload,
                           // 64 no BN table setup, 128 no tile loop (coarse flags only: a flag test inside an
                           // element loop is a branch around a load and serialises it)
};

// two problems of one shape for one launch (the *_pair_kernel variants: blockIdx.z picks the element)
struct ConvPair { ConvArgs a[2]; };

// fields the launcher derives (a pair shares them: the two problems have one shape)
inline void copy_launch_geometry(ConvArgs& b, const ConvArgs& a) {
    b.wshift = a.wshift; b.hwshift = a.hwshift; b.any_ups = a.any_ups; b.xcd_gx = a.xcd_gx; b.xcd_gy = a.xcd_gy; b.dbg = a.dbg; b.col_slices = a.col_slices;
    b.sl_rem = a.sl_rem; b.sl_gx_big = a.sl_gx_big; b.sl_gx_small = a.sl_gx_small;
}

// Two problems can share a launch when everything that shapes the grid, the LDS layout and the code path agrees.
inline bool conv_pairable(const ConvArgs& a, const ConvArgs& b) {
    if (a.nseg != b.nseg || a.Ccat != b.Ccat || a.K != b.K || a.taps != b.taps || a.Kpad != b.Kpad || a.Npad != b.Npad || a.Nout != b.Nout ||
        a.M != b.M || a.H != b.H || a.W != b.W || a.xbf16 != b.xbf16 || a.qin_bits != b.qin_bits || a.training != b.training || a.lda != b.lda ||
        a.mse_tgt || b.mse_tgt)
        return false;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C != b.seg[i].C || a.seg[i].ld != b.seg[i].ld || a.seg[i].ups != b.seg[i].ups) return false;
    return true;
}

struct WgradArgs {
    const float* dy;       // [M][lddy]
    int lddy;
    int Cout;
    int nseg;
    int Ccat;
    Seg seg[MAXSEG];
    const float* gamma;
    const float* beta;
    int taps;
    int M, H, W;
    float* dw;             // [Cout][Ccat][taps] (torch layout), atomically accumulated
    int rows_per_block;
    int ctw;               // c-tiles per job (1x1); 3x3 jobs take one c-tile x 9 taps
    const float* img;      // stem
    int IH, IW;
    int xbf16;             // 1 = the segments' x are bf16; 2 = x and dy are bf16 (dw stays fp32)
    int qin_bits;          // > 0: the conv's input is QuanInput(relu(bn(x))): the weight gradient contracts dY with the QUANTISED activation
    int split;             // fp32 storage: 1 = contract on the bf16 matrix pipe, operands cut into three bf16 pieces (planner option f32_split)
    int split_planes;      // split contraction, 1x1 (round 6, planner option wgrad_split_planes): 1 = operands cut once on the way into LDS, three bf16
                           // planes read back by ds_read_b64_tr_b16 (wgrad5_split_kernel) for slices of at most 9 channel tiles
    int bf16_dma;          // xbf16 == 2, 1x1: 1 = the LDS-DMA ring kernel where its preconditions hold (the plan's snapshot of planner option
                           // wgrad_bf16_dma), 0 = always the register-staged kernel
    // stem only (planner option stem_fuse_dz): sx != null = dy is NOT read; the kernel derives d(loss)/d(conv0 output) itself from the conv's
    // own output sx [M][128], the gradient of the POOLED features sgy [M / 4][128], conv0's batch statistics and the reductions of
    // stem_bwd_kernel<0> -- the second pass of the stem's BatchNorm-ReLU-pool backward and its 200 MB tensor never exist
    const float* sx;
    const float* sgy;
    const double* sstats;  // [2][128] sum, sum of squares of sx
    const double* sred;    // [2][128] sum(dz), sum(dz * xhat) (stem_bwd_kernel<0>)
    double scount;         // rows behind sstats (= M)
};

// third-generation 1x1 weight gradient (wgrad3_kernels.hip): one workgroup owns the whole [128][CW] output for a range
// of pixels and writes a partial tile; wgrad_reduce_kernel sums the partials of a gradient bucket
struct Wg3Args {
    WgradArgs w;
    float* part;           // [splits][128][Ccat] partial sums
    int rows_per_split;    // multiple of 32
    int c0, CW;            // channel slice of this launch (CW <= 320, multiple of 32)
    int any_ups;
    int wshift, hwshift;   // log2(W), log2(H * W) when both are powers of two, else -1 (wgrad4_bf16_kernel's up-sample map)
};
// LDS geometry of the stem weight gradient (wgrad3_stem_kernel): an input image row is stored as 3 channel rows of CP floats
// (3 zero columns left and right), a ring row every RP floats; CP = 17 and RP = 7 (mod 32) put im2col column k = c*49 + ky*7 + kx
// on bank k (mod 32): the 32 lanes of a ds_read_b32 group never collide.
__host__ __device__ inline int stem_cp(int IW) { int cp = IW + 6; return cp + ((17 - cp % 32) + 32) % 32; }
__host__ __device__ inline int stem_rp(int IW) { int rp = 3 * stem_cp(IW); return rp + ((7 - rp % 32) + 32) % 32; }
constexpr int STEM_CHUNK = 64;          // output pixels per dY chunk
constexpr int STEM_K = 147;             // 3 * 7 * 7
struct WgReduceEntry {     // grads[dst + i] = sum_s ws[part + s*numel + i]
    int64_t part;          // float offset in the workspace float region
    int64_t dst;           // float offset in the gradient arena
    int S, numel;
    int taps, pad_;        // > 1: the partials are [taps][numel / taps] and the destination is torch's [numel / taps][taps]
};

constexpr int MAXGSRC = 8;  // conv consumers gathered per launch (more: further launches with accumulate = 1)

struct GradSrc {           // one conv node that reads the tensor: its BN backward contributes A*dz + E - D*x
    const float* dz;       // that node's masked data gradient [Mc][lddz] (EP_BWD epilogue of conv_kernel)
    const double* red;     // that node's reductions [2][lddz]: sum(dz), sum(dz*xhat)
    const float* gamma;    // that node's BN weight [lddz]
    int lddz;              // channels of that node's concat
    int choff;             // where this tensor sits in that concat
    int ups;               // 1: read through the nearest-upsample map (that node runs at 2H x 2W)
    int pad_;
};

struct GradGatherArgs {    // dX = sum_consumers scale*(dz - mean(dz) - xhat*mean(dz*xhat)), written once per tensor
    int xbf16;             // 1: x is stored as bf16; 2: x, every dz and gx are bf16
    int nsrc;
    int accumulate;        // 0: store, 1: add to what is there
    GradSrc src[MAXGSRC];
    const float* x;        // the tensor [rows][ld]
    float* gx;             // its gradient [rows][ld]
    const double* stats;   // its batch statistics [2][C]
    double count;          // rows behind stats
    int C, ld;
    int rows, H, W;        // geometry of the tensor itself
};

constexpr int MAXBNG = 32;
struct BnParamGradArgs {   // dgamma = sum(dz*xhat), dbeta = sum(dz) for a batch of BatchNorms
    int n;
    struct { const double* red; float* dgamma; float* dbeta; int C; int pad_; } e[MAXBNG];
};

// device-side tables (uploaded once at bind)
struct RepackEntry {       // weight -> GEMM operand layouts
    int64_t src;           // float offset in the parameter arena, [Cout][Cin][taps]
    int64_t dstF;          // float offset in workspace: forward B  [taps][KpadF/4][NpadF][4], K=Cin, N=Cout
    int64_t dstB;          // float offset in workspace: backward B [taps][KpadB/4][NpadB][4], K=Cout, N=Cin (taps flipped); -1 if unused
    int Cout, Cin, taps;
    int KpadF, NpadF, KpadB, NpadB;
    int pad_;
};

struct RunStatEntry {      // running_mean / running_var update of one BN segment
    int64_t stats;         // double offset into stats arena: sum[C], sumsq[C]
    int64_t rmean;         // float offset in buffer arena
    int64_t rvar;
    int64_t counter;       // int64 offset in counter arena, or -1 (only the first segment of a BN carries it)
    double count;          // samples behind `stats` (source tensor rows)
    double unbias;         // n/(n-1) with n = samples the BatchNorm saw (4x count for an upsampled segment)
    int C;
    int times;             // 1, or 2 for BNs the reference re-runs under torch.utils.checkpoint (2nd update from backward)
};

struct PoolArgs {
    const float* x;        // [N*H*W][C]   (H, W = input spatial dims)
    float* y;              // [N*H/2*W/2][C]
    double* ystats;        // [2][C]
    const double* xstats;  // MODE 1: stats of x
    const float* gamma; const float* beta; const float* rmean; const float* rvar;
    double count;
    int training;
    int N, H, W, C;
    // backward
    const float* gy;       // [lo][C]
    float* gx;             // [hi][C]
    double* red;           // stem backward reductions [2][C]
    int xbf16;             // pool backward: 1 = x is bf16; 2 = x, gy and gx are bf16
};

struct QuantEntry {      // one target conv: weight [O][I][KK] at float offset `off` of the arena
    int64_t off;
    int O, I, KK, pad_;
};

struct TernArgs {
    const float* x;          // [M][ldx] raw (pre-BN) activations
    const float* scale;      // [C] folded BatchNorm scale / shift, or null: fold in the kernel from the fields below
    const float* shift;
    const uint64_t* wpos;    // [taps][G][Opad] bit c of word (tap,g,o) = (w[o][64g+c][tap] == +1)
    const uint64_t* wneg;
    float* y;                // [M][ldy]
    int M, H, W, C, O, Opad, taps, bits_i;
    int ldx, ldy;
    // in-kernel BatchNorm fold (scale == null): batch statistics of the input tensor (training) or running statistics
    const double* xstats;    // [2][C] sum, sum of squares
    double count;
    const float* gamma; const float* beta; const float* rmean; const float* rvar;
    int training;
    double* ystats;          // [2][O] sum / sum of squares of the output (or null)
    int variant;             // plan path: 1 = ternary_conv_pixels_kernel (lane = pixel, round 5), 0 = ternary_conv_planes_kernel (wave = pixel)
    int pad_;
    uint64_t* planes;        // plan path: [M + 1][16] bit-plane records of the quantised input (ternary_planes_kernel), or null
};
// One record of 16 words per input pixel: words [7 g + b] = bit b of the quantised activation of channels 64 g .. 64 g + 63 (g < 2,
// b < 7: C <= 128, bits_i <= 8), word 14 = the sum of the pixel's quantised activations, word 15 = 0; record M is all zero (what a
// tap outside the image reads).
constexpr int TERN_REC_WORDS = 16;
// Bit order inside a 64-channel plane / mask word (round 5): within each 32-bit half, bit 8 i + k holds channel 4 k + i (i < 4, k < 8) --
// the order in which ternary_planes_rows_kernel finds the channels when it shifts a dword of four quantised bytes (channels 4 k .. 4 k + 3)
// by k and ORs it into the plane: (d >> b) & 0x01010101 leaves bit b of byte i at position 8 i.  Every producer of plane words and every
// packer of weight masks uses the same map, so the AND pairs the right channels; nothing else depends on it.
__host__ __device__ inline int tern_bit_of_chan(int c) { return (c & 32) + 8 * (c & 3) + ((c & 31) >> 2); }      // c in 0 .. 63
__host__ __device__ inline int tern_chan_of_bit(int b) { return (b & 32) + 4 * (b & 7) + ((b & 31) >> 3); }      // b in 0 .. 63

struct AugSample {           // one training sample of cunet_augment_batch (host-computed crop geometry, pylib/HumanAug.py:118-142); 192 bytes
    const float* src;        // 3 x sh x sw fp32 CHW image in [0, 1]
    unsigned* mm;            // 5 words of scratch: min / max bit patterns of the image (pre-shrink) and of the canvas, maximum output byte
    unsigned char* i8;       // pre-shrink only: byte-scaled image sh x sw x 3, its horizontal pass sh x sw1 x 3, the shrunk image sh1 x sw1 x 3
    unsigned char* t1;
    unsigned char* i1;
    unsigned char* c8;       // byte-scaled canvas ch x cw x 3
    unsigned char* r8;       // rotated only: the rotated canvas without its padding, win_h x win_w x 3
    unsigned char* t2;       // horizontal pass of the final resize, win_h x res x 3
    unsigned char* o8;       // res x res x 3
    double rm[6];            // rotated only: PIL's destination -> source affine map of Image.rotate on the cw x ch canvas
    int sh, sw;
    int sh1, sw1;            // pre-shrink only: int(sh / sf), int(sw / sf)
    int ulx, uly;            // upper-left corner of the (padded) canvas in the (shrunk) image
    int cw, ch;              // canvas size incl. the rotation padding
    int win_w, win_h;        // canvas size without it: what is resized to res x res
    int pad;
    int flip, rotated, pre;
    float gain[3];           // per-channel colour gain (clamped to [0, 1] afterwards)
    int pad_;
};
static_assert(sizeof(AugSample) == 192, "AugSample is bound from Python (cu_net_amd/augment.py _REC)");

struct TernPackEntry {       // one conv whose (ternary) weights are packed into AND-popcount bit masks
    int64_t src;             // float offset of the weight [O][C][taps] in the parameter arena
    int64_t dst;             // uint64 offset of wpos in the mask region; wneg follows at dst + taps*G*Opad
    int O, C, taps, Opad;
};

// Explicit global-address-space loads.  A pointer that reaches a lane through LDS or through a
// dynamically indexed kernarg struct has lost its address space, and hipcc then emits flat_load,
// which also counts on LGKM: the next `s_waitcnt lgkmcnt(0)` in front of an LDS-fed MFMA group would
// wait for the whole HBM prefetch (measured: the forward conv ran at 37 % MFMA utilisation because of it).
#if defined(__HIPCC__)
typedef const float __attribute__((address_space(1)))* gptr_f32;
typedef const f32x4 __attribute__((address_space(1)))* gptr_f32x4;
__device__ __forceinline__ float ldg1(const float* p) { return *(gptr_f32)(uintptr_t)p; }
__device__ __forceinline__ float4 ldg4(const float* p) {
    const f32x4 v = *(gptr_f32x4)(uintptr_t)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// X-operand loaders for the two activation storages: XB = 0 fp32, XB = 1 bf16 (training with bf16 activations keeps
// gradients in fp32).  The argument structs type every activation pointer `const float*`; for XB = 1 it addresses bf16
// data and all offsets / leading dimensions count ELEMENTS of the storage type.
typedef const unsigned short __attribute__((address_space(1)))* gptr_u16;
typedef const unsigned __attribute__((address_space(1)))* gptr_u32;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef const u32x2_t __attribute__((address_space(1)))* gptr_u32x2;
__device__ __forceinline__ float bf16_bits_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_bits_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
// pointer arithmetic in elements of the storage type
template <int XB> __device__ __forceinline__ const float* xadv(const float* base, size_t off) {
    return XB ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + off) : base + off;
}
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {      // round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <int XB> __device__ __forceinline__ void stx1(float* base, size_t off, float v) {
    if (XB) reinterpret_cast<unsigned short*>(base)[off] = f32_to_bf16_rne(v);
    else base[off] = v;
}
template <int XB> __device__ __forceinline__ void stx4(float* base, size_t off, float4 v) {      // offset % 4 == 0
    if (XB) {
        uint2 q;
        q.x = (unsigned)f32_to_bf16_rne(v.x) | ((unsigned)f32_to_bf16_rne(v.y) << 16);
        q.y = (unsigned)f32_to_bf16_rne(v.z) | ((unsigned)f32_to_bf16_rne(v.w) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = q;
    } else {
        *reinterpret_cast<float4*>(base + off) = v;
    }
}
template <int XB> __device__ __forceinline__ float ldx1(const float* base, size_t off) {
    if (XB) return bf16_bits_lo((unsigned)*(gptr_u16)(uintptr_t)(reinterpret_cast<const unsigned short*>(base) + off));
    return ldg1(base + off);
}
template <int XB> __device__ __forceinline__ float2 ldx2(const float* base, size_t off) {      // 2 consecutive elements (even offset)
    if (XB) {
        const unsigned v = *(gptr_u32)(uintptr_t)(reinterpret_cast<const unsigned short*>(base) + off);
        return make_float2(bf16_bits_lo(v), bf16_bits_hi(v));
    }
    return make_float2(ldg1(base + off), ldg1(base + off + 1));
}
template <int XB> __device__ __forceinline__ float4 ldx4(const float* base, size_t off) {      // 4 consecutive elements (offset % 4 == 0)
    if (XB) {
        const u32x2_t v = *(gptr_u32x2)(uintptr_t)(reinterpret_cast<const unsigned short*>(base) + off);
        return make_float4(bf16_bits_lo(v.x), bf16_bits_hi(v.x), bf16_bits_lo(v.y), bf16_bits_hi(v.y));
    }
    return ldg4(base + off);
}
#endif

// Tuning knobs.  The SHIPPED library (build.sh) ignores the environment: every knob is its measured default and the
// work-skipping timing switches do not exist.  `CUNET_TUNING=1 build.sh` compiles with -DCUNET_TUNING into a separate
// libcunet_hip_tuning.so in which the knobs are read from CUNET_* variables (tools/ only; never benchmarked as product).
#ifdef CUNET_TUNING
inline int tune_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
inline float tune_float(const char* name, float dflt) { const char* v = getenv(name); return v ? (float)atof(v) : dflt; }
#define CUNET_DBG(p, bit) ((p).dbg & (bit))
#else
inline int tune_int(const char*, int dflt) { return dflt; }
inline float tune_float(const char*, float dflt) { return dflt; }
#define CUNET_DBG(p, bit) 0
#endif

// QuanInput forward on a post-ReLU activation (utils/quantize.py:15-42,52-55 with bits in 3..15): clamp to 1 - 2^-(b-1),
// round half to even onto the grid 2^-(b-1).  (v >= 0 here, so the lower clamp of C() never acts.)
__device__ __forceinline__ float quan_input_act(float v, int bits) {
    const float qs = (float)(1 << (bits - 1));
    return rintf(fminf(v, 1.f - 1.f / qs) * qs) / qs;
}

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace cunet
