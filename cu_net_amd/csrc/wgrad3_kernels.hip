// 1x1 weight gradient, third generation: LDS-staged, atomics-free.
//     dW[n][c] = sum_m dY[m][n] * relu(bn(X))[m][c]          (autograd wgrad of models/cu_net.py:24,43)
//
// Why a third design.  wgrad2 (one wave = 4 x 2 output tiles fed by its own per-lane global loads) moves 6 operand
// dwords per lane for every 8 MFMAs, re-reads dY once per 64-channel group, keeps only 2 waves per SIMD at 256 VGPRs
// and commits with one fp32 atomic per output element per block (23 % of its time): 0.185 of the fp32 MFMA peak, 1.85x
// the algorithmic HBM traffic (profiles/r01_*).  Here a 512-thread workgroup owns the WHOLE output [Cout = 128][CW <= 320]
// for a range of pixels:
//   * a chunk of P = 32 pixels of dY [P][128] and of the ACTIVATED X [P][CW] is staged once in LDS (BatchNorm + ReLU
//     applied on the way in, once per element instead of once per use), double buffered: the global loads of chunk
//     j+1 are issued before the MFMA loop of chunk j and written to the other buffer after it -- one barrier per chunk;
//   * the 8 waves share the chunk: wave w owns output-channel tile (w & 3) and half of the input-channel tiles (or, when
//     there are <= 5 of them, all of them on every other pixel pair); MFMA operands are conflict-free ds_read_b32;
//     every byte of dY and X is read from HBM exactly once per launch;
//   * each workgroup stores its partial [128][CW] tile with plain coalesced stores into part[split][n][c]; one
//     deterministic reduce kernel per gradient bucket sums the splits into the gradient arena (no atomics, bitwise
//     reproducible dW).
// Roofline: 2*128*CW flops per (128 + CW)*4 bytes = 45.7 flop/B at CW = 320 > the fp32 ridge (~25): MFMA-bound.
#include <type_traits>
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

namespace cunet {

constexpr int WG3_P = 32;            // pixels per chunk
constexpr int WG3_THREADS = 512;
// Minimum waves per SIMD the fp32 kernels are compiled for: 2 = up to 256 VGPRs (the 320-channel instantiation uses 220, i.e. two of
// its waves fill a SIMD's register file and no data-gradient wave can sit next to them); a probe build with 3 (<= 168 VGPRs) measures
// whether leaving room for a co-resident wave of the caller's stream is worth the spills (-DCUNET_WG3_MIN_WAVES=3).
#ifndef CUNET_WG3_MIN_WAVES
#define CUNET_WG3_MIN_WAVES 2
#endif
constexpr int WG3_MIN_WAVES = CUNET_WG3_MIN_WAVES;
constexpr int WG3_NOUT = 128;        // output channels (4 tiles): the bottleneck / adapter convs of the network
constexpr int WG3_MAXCW = 320;

// EMU (planner option f32_split, fp32 storage only): the same staging, the contraction on the bf16 matrix pipe -- a lane takes its
// operands as 8 consecutive pixels of one channel (eight strided ds_read_b32, as many as the eight fp32 k-steps they replace), cuts
// each value into three bf16 pieces and issues six v_mfma_f32_32x32x16_bf16 per tile pair (conv_common.h, conv_body's XBG = 6).
template <int CTW, bool SPLITK, int XB, bool EMU = false>
__global__ __launch_bounds__(WG3_THREADS, WG3_MIN_WAVES) void wgrad3_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgradArgs& p = q.w;
    const int CW = q.CW;                                   // channels of this launch's slice (multiple of 32)
    const int ct = CW >> 5;
    float* sc = reinterpret_cast<float*>(smem);            // [CW] BatchNorm scale
    float* sh = sc + WG3_MAXCW;                            // [CW] BatchNorm shift
    float* buf0 = sh + WG3_MAXCW;                          // 2 x { dY [P][128], X [P][LDX] }
    constexpr int LDX = (SPLITK ? CTW : 2 * CTW) * 32;     // compile-time row pitch of X (>= CW): every LDS offset of the MFMA loop is an immediate
    constexpr int bufsz = WG3_P * (WG3_NOUT + LDX);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;

    // ---- BatchNorm scale / shift of this slice (batch statistics of the segment tensors, as every consumer derives them)
    for (int c = tid; c < CW; c += WG3_THREADS) {
        const int cc = q.c0 + c;
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        const int lc = cc - sg.choff;
        const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[cc] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[cc] - mean * scale);
    }

    // ---- staging plan of this thread (chunk-invariant): 2 float4 of dY, up to NX float4 of X
    constexpr int NX = SPLITK ? (CTW + 1) / 2 : CTW;       // ceil(ct / 2) float4 per thread: P * CW / 4 / 512
    const int cw4 = CW >> 2;
    const int nx4 = WG3_P * cw4;                           // float4 items of X per chunk
    int xp[NX], xc[NX];                                    // pixel within the chunk, first channel within the slice
    const float* xbase[NX];
    int xld[NX], xups[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        int idx = tid + WG3_THREADS * j;
        if (idx >= nx4) idx = nx4 - 1;                     // duplicates of the last item: same value to the same LDS address
        xp[j] = idx / cw4;
        xc[j] = (idx - xp[j] * cw4) << 2;
        const int cc = q.c0 + xc[j];
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        xbase[j] = xadv<XB>(sg.x, (size_t)(cc - sg.choff));
        xld[j] = sg.ld;
        xups[j] = sg.ups;
    }
    const int ap0 = tid >> 5, ac0 = (tid & 31) << 2;       // dY item 0: pixel tid/32, channels 4*(tid%32); item 1: pixel + 16

    const int row_begin = blockIdx.x * q.rows_per_split;
    int row_end = row_begin + q.rows_per_split;
    if (row_end > p.M) row_end = p.M;
    const int nchunks = (row_end - row_begin + WG3_P - 1) / WG3_P;

    float4 av[2], xv[NX];
    float4 s4[NX], h4[NX];                                 // BatchNorm scale / shift of this thread's channels (chunk-invariant, filled below)
    bool aok[2], xok[NX];
    auto issue = [&](int chunk) {                          // raw global loads of one chunk into registers (clamped addresses)
        const int m0 = row_begin + chunk * WG3_P;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ap0 + 16 * j;
            aok[j] = m < row_end;
            const int mc = aok[j] ? m : row_begin;
            av[j] = ldg4(p.dy + (size_t)mc * p.lddy + ac0);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int m = m0 + xp[j];
            xok[j] = m < row_end;
            const int mc = xok[j] ? m : row_begin;
            int row = mc;
            if (q.any_ups) {                               // nearest-upsample index map (models/cu_net.py:250,265): (y >> 1, x >> 1)
                const int nimg = mc / HW;
                const int rem = mc - nimg * HW;
                const int py = rem / p.W;
                const int px = rem - py * p.W;
                const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
                row = xups[j] ? rowU : mc;
            }
            xv[j] = ldx4<XB>(xbase[j], (size_t)row * xld[j]);
        }
    };
    auto commit = [&](float* buf) {                        // registers -> LDS, BatchNorm + ReLU on X, zeros beyond the range
        float* A = buf;
        float* X = buf + WG3_P * WG3_NOUT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = aok[j] ? av[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(A + (ap0 + 16 * j) * WG3_NOUT + ac0) = v;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            float4 v;
            v.x = fmaxf(fmaf(xv[j].x, s4[j].x, h4[j].x), 0.f);
            v.y = fmaxf(fmaf(xv[j].y, s4[j].y, h4[j].y), 0.f);
            v.z = fmaxf(fmaf(xv[j].z, s4[j].z, h4[j].z), 0.f);
            v.w = fmaxf(fmaf(xv[j].w, s4[j].w, h4[j].w), 0.f);
            if (!xok[j]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(X + xp[j] * LDX + xc[j]) = v;
        }
    };

    // ---- tile ownership: n tile = wave & 3; c tiles: split-K -> all ct on pixel pairs of parity (wave >> 2),
    //      otherwise half (wave >> 2) owns tiles [half * CTW, half * CTW + CTW) (the last one clamped when ct is odd)
    const int nt = wave & 3;
    const int half = wave >> 2;
    const int cb = SPLITK ? 0 : half * CTW;
    int ctile[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t) ctile[t] = (cb + t < ct) ? cb + t : ct - 1;

    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    issue(0);
    __syncthreads();                                       // sc / sh visible
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        s4[j] = *reinterpret_cast<const float4*>(sc + xc[j]);
        h4[j] = *reinterpret_cast<const float4*>(sh + xc[j]);
    }
    commit(buf0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float* cur = buf0 + (chunk & 1) * bufsz;
        const bool more = chunk + 1 < nchunks;
        if (more) issue(chunk + 1);                        // in flight across the MFMA loop
        // k-step i of this wave contracts pixel pair KS*i + (SPLITK ? half : 0); operands are fetched PF steps ahead of
        // the MFMAs that use them (left to itself hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per tile, which
        // exposes the whole LDS latency on every MFMA: measured 47 % of the matrix pipe)
        if constexpr (EMU) {
            // split-K: half h of the waves takes pixels 16 h .. 16 h + 15 of the chunk, otherwise both k-steps
            constexpr int NS = SPLITK ? 1 : 2;
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                const int p0 = 16 * (SPLITK ? half : ks) + 8 * hi;
                const float* A = cur + p0 * WG3_NOUT + nt * 32 + li;
                const float* X = cur + WG3_P * WG3_NOUT + p0 * LDX + li;
                float af[8], xf[CTW][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = A[j * WG3_NOUT];
#pragma unroll
                for (int t = 0; t < CTW; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[t][j] = X[j * LDX + ctile[t] * 32];
                u32x4 ah, am, al;
                split_bf16x3(af, ah, am, al);
#pragma unroll
                for (int t = 0; t < CTW; ++t) {
                    u32x4 xh, xm, xl;
                    split_bf16x3(xf[t], xh, xm, xl);
                    acc[t] = mfma_split6(ah, am, al, xh, xm, xl, acc[t]);
                }
            }
            if (more) commit(buf0 + ((chunk + 1) & 1) * bufsz);
            __syncthreads();
            continue;
        }
        constexpr int NK = SPLITK ? WG3_P / 4 : WG3_P / 2;
        constexpr int KS = SPLITK ? 2 : 1;
        constexpr int PF = 2;
        const float* A = cur + nt * 32 + li + (2 * (SPLITK ? half : 0) + hi) * WG3_NOUT;
        const float* X = cur + WG3_P * WG3_NOUT + li + (2 * (SPLITK ? half : 0) + hi) * LDX;
        float ar[PF + 1], xr[PF + 1][CTW];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            ar[i] = A[2 * KS * i * WG3_NOUT];
#pragma unroll
            for (int t = 0; t < CTW; ++t) xr[i][t] = X[2 * KS * i * LDX + ctile[t] * 32];
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if (i + PF < NK) {
                ar[(i + PF) % (PF + 1)] = A[2 * KS * (i + PF) * WG3_NOUT];
#pragma unroll
                for (int t = 0; t < CTW; ++t) xr[(i + PF) % (PF + 1)][t] = X[2 * KS * (i + PF) * LDX + ctile[t] * 32];
            }
#pragma unroll
            for (int t = 0; t < CTW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i % (PF + 1)], xr[i % (PF + 1)][t], acc[t], 0, 0, 0);
            // pin the software pipeline: this step's LDS reads (for step i + PF) BEFORE this step's MFMAs -- without it the
            // scheduler sinks the reads next to their uses and both waves of a SIMD stall on the LDS latency in phase
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + CTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CTW, 0);
        }
        if (more) commit(buf0 + ((chunk + 1) & 1) * bufsz);
        __syncthreads();
    }

    // ---- split-K: the odd-pair half hands its tiles to the even-pair half through LDS
    if (SPLITK) {
        float* red = buf0;                                  // [4 waves][CTW][1024] (<= 80 KB, the chunk buffers are free now)
        if (half == 1) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((nt * CTW + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((nt * CTW + t) * 16 + r) * 64 + lane];
        }
        if (half == 1) return;
    }
    // ---- partial tile -> part[split][n][c]: MFMA C layout, lane = input channel (32 consecutive floats per row)
    float* out = q.part + (size_t)blockIdx.x * WG3_NOUT * p.Ccat;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        if (cb + t >= ct) continue;                         // the clamped duplicate of an odd tile count
        const int c = q.c0 + (cb + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * p.Ccat + c] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same design on bf16 MFMA (bf16 storage of activations AND gradient tensors, FusedTrainer(bf16_grads=True)):
// v_mfma_f32_32x32x16_bf16 contracts 16 pixels per instruction and wants each lane's 8 k-values (pixels) contiguous,
// while NHWC keeps a pixel's CHANNELS contiguous.  The transpose happens on the way into LDS: a lane loads 4 consecutive
// pixels x 8 channels (four 16-byte loads), applies BatchNorm + ReLU in fp32 (X only), re-rounds with v_cvt_pk_bf16_f32 --
// whose two inputs are the SAME channel of two neighbouring pixels, so the conversion itself produces the pixel-major
// packing -- and writes eight 8-byte pieces into a channel-major image  T[channel][64 pixels (+8 pad)].  MFMA fragments
// are then single ds_read_b128 (row pitch 144 B: conflict-free).  Chunk = 64 pixels = 4 MFMA k-steps; 57 KB of HBM data
// per chunk against 40 MFMAs of 32 cycles: HBM-bound (AI = 91 flop/B << the bf16 ridge ~310), which is the point.
constexpr int WG3B_P = 64;           // pixels per chunk
constexpr int WG3B_LDP = 72;         // bf16 elements per LDS row (64 + 8: 144 B pitch)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int CTW, bool SPLITK>
__global__ __launch_bounds__(WG3_THREADS, 2) void wgrad3_bf16_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgradArgs& p = q.w;
    const int CW = q.CW;
    const int ct = CW >> 5;
    float* sc = reinterpret_cast<float*>(smem);            // [CW]
    float* sh = sc + WG3_MAXCW;
    unsigned short* buf0 = reinterpret_cast<unsigned short*>(sh + WG3_MAXCW);      // 2 x T[128 + CW][LDP] bf16
    const int bufsz = (WG3_NOUT + CW) * WG3B_LDP;                                  // elements per buffer

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;
    const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(p.dy);

    for (int c = tid; c < CW; c += WG3_THREADS) {
        const int cc = q.c0 + c;
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        const int lc = cc - sg.choff;
        const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[cc] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[cc] - mean * scale);
    }

    // ---- staging: the image has 4 + ct row groups of 32 channels (0..3 = dY, 4.. = X tiles); wave w stages groups w, w + 8.
    // Inside a group a lane owns pixels 4*q4 .. 4*q4+3 (q4 = lane & 15) of channels 8*c8 .. 8*c8+7 (c8 = lane >> 4).
    const int q4 = lane & 15, c8 = lane >> 4;
    const int nrg = 4 + ct;
    const unsigned short* gbase[2];
    int gld[2], gups[2], grow[2], gsc[2];
    bool gx[2], gon[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = wave + 8 * j;
        gon[j] = g < nrg;
        const int gg = gon[j] ? g : 0;
        gx[j] = gg >= 4;
        grow[j] = gg * 32 + 8 * c8;                        // first LDS row of this lane's 8 channels
        gsc[j] = gx[j] ? (gg - 4) * 32 + 8 * c8 : 0;       // their position in sc / sh
        if (gx[j]) {
            const int cc = q.c0 + (gg - 4) * 32 + 8 * c8;
            int s = 0;
            for (int t = 1; t < p.nseg; ++t)
                if (cc >= p.seg[t].choff) s = t;
            const Seg& sg = p.seg[s];
            gbase[j] = reinterpret_cast<const unsigned short*>(sg.x) + (cc - sg.choff);
            gld[j] = sg.ld; gups[j] = sg.ups;
        } else {
            gbase[j] = dy16 + gg * 32 + 8 * c8;
            gld[j] = p.lddy; gups[j] = 0;
        }
    }

    const int row_begin = blockIdx.x * q.rows_per_split;
    int row_end = row_begin + q.rows_per_split;
    if (row_end > p.M) row_end = p.M;
    const int nchunks = (row_end - row_begin + WG3B_P - 1) / WG3B_P;

    uint4 ld[2][4];
    unsigned okm[2];
    auto issue = [&](int chunk) {
        const int m0 = row_begin + chunk * WG3B_P + 4 * q4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            okm[j] = 0;
            if (!gon[j]) continue;                          // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + i;
                const bool ok = m < row_end;
                okm[j] |= (unsigned)ok << i;
                const int mc = ok ? m : row_begin;
                int row = mc;
                if (q.any_ups && gups[j]) {
                    const int nimg = mc / HW;
                    const int rem = mc - nimg * HW;
                    const int py = rem / p.W;
                    const int px = rem - py * p.W;
                    row = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
                }
                ld[j][i] = *reinterpret_cast<const uint4*>(gbase[j] + (size_t)row * gld[j]);
            }
        }
    };
    auto commit = [&](unsigned short* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!gon[j]) continue;
            unsigned r[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (okm[j] >> i) & 1;
                r[i][0] = ok ? ld[j][i].x : 0u; r[i][1] = ok ? ld[j][i].y : 0u;
                r[i][2] = ok ? ld[j][i].z : 0u; r[i][3] = ok ? ld[j][i].w : 0u;
            }
            uint2* dst = reinterpret_cast<uint2*>(buf + (size_t)grow[j] * WG3B_LDP + 4 * q4);      // + e * LDP elements per channel
            if (gx[j]) {
                float s8[8], h8[8];
                *reinterpret_cast<float4*>(s8) = *reinterpret_cast<const float4*>(sc + gsc[j]);
                *reinterpret_cast<float4*>(s8 + 4) = *reinterpret_cast<const float4*>(sc + gsc[j] + 4);
                *reinterpret_cast<float4*>(h8) = *reinterpret_cast<const float4*>(sh + gsc[j]);
                *reinterpret_cast<float4*>(h8 + 4) = *reinterpret_cast<const float4*>(sh + gsc[j] + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {              // channel pair (2k, 2k+1)
                    float lo[4], hh[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = (okm[j] >> i) & 1;
                        lo[i] = fmaxf(fmaf(bf16_bits_lo(r[i][k]), s8[2 * k], h8[2 * k]), 0.f);
                        hh[i] = fmaxf(fmaf(bf16_bits_hi(r[i][k]), s8[2 * k + 1], h8[2 * k + 1]), 0.f);
                        if (!ok) { lo[i] = 0.f; hh[i] = 0.f; }     // relu(shift) of a row beyond the range must not leak in
                    }
                    dst[(2 * k) * (WG3B_LDP / 4)] = make_uint2(cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]));
                    dst[(2 * k + 1) * (WG3B_LDP / 4)] = make_uint2(cvt_pk_bf16(hh[0], hh[1]), cvt_pk_bf16(hh[2], hh[3]));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned a0 = (r[0][k] & 0xffffu) | (r[1][k] << 16), a1 = (r[2][k] & 0xffffu) | (r[3][k] << 16);
                    const unsigned b0 = (r[0][k] >> 16) | (r[1][k] & 0xffff0000u), b1 = (r[2][k] >> 16) | (r[3][k] & 0xffff0000u);
                    dst[(2 * k) * (WG3B_LDP / 4)] = make_uint2(a0, a1);
                    dst[(2 * k + 1) * (WG3B_LDP / 4)] = make_uint2(b0, b1);
                }
            }
        }
    };

    const int nt = wave & 3;
    const int half = wave >> 2;
    const int cb = SPLITK ? 0 : half * CTW;
    int ctile[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t) ctile[t] = (cb + t < ct) ? cb + t : ct - 1;

    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    issue(0);
    __syncthreads();
    commit(buf0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned short* cur = buf0 + (size_t)(chunk & 1) * bufsz;
        const bool more = chunk + 1 < nchunks;
        if (more) issue(chunk + 1);
        constexpr int NK = SPLITK ? WG3B_P / 32 : WG3B_P / 16;
        constexpr int KS = SPLITK ? 2 : 1;
        const unsigned short* A = cur + (size_t)(nt * 32 + li) * WG3B_LDP + 8 * hi + (SPLITK ? 16 * half : 0);
        const unsigned short* X[CTW];
#pragma unroll
        for (int t = 0; t < CTW; ++t) X[t] = cur + (size_t)(WG3_NOUT + ctile[t] * 32 + li) * WG3B_LDP + 8 * hi + (SPLITK ? 16 * half : 0);
        bf16x8_t ar[2], xr[2][CTW];                          // operands one k-step ahead of the MFMAs
        ar[0] = *reinterpret_cast<const bf16x8_t*>(A);
#pragma unroll
        for (int t = 0; t < CTW; ++t) xr[0][t] = *reinterpret_cast<const bf16x8_t*>(X[t]);
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if (i + 1 < NK) {
                ar[(i + 1) & 1] = *reinterpret_cast<const bf16x8_t*>(A + 16 * KS * (i + 1));
#pragma unroll
                for (int t = 0; t < CTW; ++t) xr[(i + 1) & 1][t] = *reinterpret_cast<const bf16x8_t*>(X[t] + 16 * KS * (i + 1));
            }
#pragma unroll
            for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i & 1], xr[i & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + CTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CTW, 0);
        }
        if (more) commit(buf0 + (size_t)((chunk + 1) & 1) * bufsz);
        __syncthreads();
    }

    if (SPLITK) {
        float* red = reinterpret_cast<float*>(buf0);
        if (half == 1) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((nt * CTW + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((nt * CTW + t) * 16 + r) * 64 + lane];
        }
        if (half == 1) return;
    }
    float* out = q.part + (size_t)blockIdx.x * WG3_NOUT * p.Ccat;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        if (cb + t >= ct) continue;
        const int c = q.c0 + (cb + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * p.Ccat + c] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The bf16 weight gradient on CDNA4's data-movement instructions (round 4).  The kernel above moves every byte through
// VGPRs twice (global -> registers -> pixel-major LDS image, eight 8-byte ds_writes per 64 bytes) and keeps ONE 64-pixel
// chunk in flight per workgroup: alone on the GPU it reached 1.1 TB/s, 0.14 of the HBM peak (profiles/r03_bf16_*), parked
// on its loads.  Here
//   * the raw NHWC rows of dY [P][128] and of the concat X [P][CW] go from HBM straight into an LDS ring by LDS-DMA
//     (global_load_lds_dwordx4: 16 bytes per lane, lane l lands at M0 + 16 l, any source address per lane -- so a ring row
//     is [dY | segment 0 | segment 1 | ...] gathered from the tensors' own rows, the nearest-upsample map included); no
//     VGPR holds data, and D - 1 of the D ring slots (4 ... 7 x 18 ... 30 KB per workgroup) are in flight while one is used;
//   * MFMA operands are taken from those channel-minor rows by ds_read_b64_tr_b16, the LDS transpose read: in a 16-lane group lane 4 j + c passes the address of (pixel j, 4-channel piece c) of a 4 x 16 block
//     and lane i receives channel i of the four pixels (tools/probes/cdna4_lds_probe.hip, gpurun_out of round 4) -- two
//     reads give the 8 consecutive pixels v_mfma_f32_32x32x16_bf16 wants per lane;
//   * BatchNorm + ReLU run on the X fragment in registers: after the transpose a lane holds eight pixels of ONE channel,
//     i.e. one scale / shift pair per (lane, channel tile), applied in fp32 and re-rounded with v_cvt_pk_bf16_f32 exactly
//     as the staging path above does -- the two kernels are bit-identical (tests/test_gpu_exact.py).
// Ring row pitch: 256 + 64 CT bytes, + 64 when CT is even, so that pitch = 64 or 192 (mod 256): the four pixel rows a
// 32-lane half touches in one transpose read start 16 banks apart (conflict-free).  One barrier per 32-pixel slot.
// Preconditions (else the staging kernel runs): every pixel range a multiple of 32 (M and rows_per_split), power-of-two
// image geometry when a segment is read through the up-sample map.
constexpr int WG4_P = 32;                  // pixels per ring slot = two MFMA k-steps
constexpr int WG4_RING0 = 3072;            // LDS byte offset of the ring (sc / sh tables in front of it)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int wg4_pitch(int ct) { return 256 + 64 * ct + ((ct & 1) ? 0 : 64); }      // bytes per pixel row
__host__ __device__ constexpr int wg4_slots(int ct) { return (160 * 1024 - WG4_RING0) / (WG4_P * wg4_pitch(ct)); }

// one LDS-DMA request: 64 lanes x 16 bytes from per-lane global addresses to LDS [dst, dst + 1024)
__device__ __forceinline__ void wg4_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wg4_wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// bf16 pair (two pixels of one channel) -> relu(bn(.)) in fp32 -> bf16 pair
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wg4_bn_relu(unsigned v, float sc, float sh) {
    const f32x2_t y = {fmaxf(fmaf(bf16_bits_lo(v), sc, sh), 0.f), fmaxf(fmaf(bf16_bits_hi(v), sc, sh), 0.f)};
    // (the compiler's own v_cvt_pk_bf16_f32 -- round to nearest even, as the asm helper above -- so that it also places the wait
    // states between this VALU write and the MFMA that reads the register)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(y, bf16x2_t));
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct Wg4Frag { u32x2_t h[2]; };          // (plain registers: the two halves a transpose read pair delivers)
__device__ __forceinline__ bf16x8_t wg4_operand(const Wg4Frag& f) {
    const u32x4_t w = {f.h[0].x, f.h[0].y, f.h[1].x, f.h[1].y};
    return __builtin_bit_cast(bf16x8_t, w);
}

// the transpose reads of one k-step: fragment 0 = dY (this wave's output-channel tile), fragments 1 .. CTW = X tiles; all
// requests, then ONE wait -- issued behind the previous step's MFMAs, whose execution covers the LDS round trip
template <int CTW, int PITCH>
__device__ __forceinline__ void wg4_read_step(Wg4Frag (&f)[CTW + 1], unsigned addr_a, unsigned addr_x) {
#define WG4_RD(F, ADDR, OFF) \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" \
                 : "=&v"((F).h[0]), "=&v"((F).h[1]) : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * PITCH) : "memory")
    WG4_RD(f[0], addr_a, 0);
    WG4_RD(f[1], addr_x, 0);
    if constexpr (CTW > 1) WG4_RD(f[CTW > 1 ? 2 : 0], addr_x, 64);
    if constexpr (CTW > 2) WG4_RD(f[CTW > 2 ? 3 : 0], addr_x, 128);
    if constexpr (CTW > 3) WG4_RD(f[CTW > 3 ? 4 : 0], addr_x, 192);
    if constexpr (CTW > 4) WG4_RD(f[CTW > 4 ? 5 : 0], addr_x, 256);
#undef WG4_RD
    // the wait names every destination: nothing reads (or moves) them before the data has landed
    if constexpr (CTW == 5)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].h[0]), "+v"(f[0].h[1]), "+v"(f[1].h[0]), "+v"(f[1].h[1]), "+v"(f[2].h[0]), "+v"(f[2].h[1]),
                     "+v"(f[3].h[0]), "+v"(f[3].h[1]), "+v"(f[4].h[0]), "+v"(f[4].h[1]), "+v"(f[5].h[0]), "+v"(f[5].h[1]) :: "memory");
    else if constexpr (CTW == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].h[0]), "+v"(f[0].h[1]), "+v"(f[1].h[0]), "+v"(f[1].h[1]), "+v"(f[2].h[0]), "+v"(f[2].h[1]),
                     "+v"(f[3].h[0]), "+v"(f[3].h[1]), "+v"(f[4].h[0]), "+v"(f[4].h[1]) :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].h[0]), "+v"(f[0].h[1]), "+v"(f[1].h[0]), "+v"(f[1].h[1]), "+v"(f[2].h[0]), "+v"(f[2].h[1]),
                     "+v"(f[3].h[0]), "+v"(f[3].h[1]) :: "memory");
}

template <int CT>      // channel tiles of the slice: CW = 32 CT; CT <= 5: split-K over the two k-steps of a slot
__global__ __launch_bounds__(WG3_THREADS, 2) void wgrad4_bf16_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool SPLITK = CT <= 5;
    constexpr int CTW = SPLITK ? CT : (CT + 1) / 2;
    static_assert(CTW >= 3 && CTW <= 5, "tile ownerships of 3 .. 5 channel tiles");
    constexpr int CW = 32 * CT;
    constexpr int PITCH = wg4_pitch(CT);                   // bytes
    constexpr int PPR = PITCH / 16;                        // 16-byte pieces per ring row (incl. the pad)
    constexpr int NI = WG4_P * PPR / 64;                   // DMA requests per slot
    constexpr int IPC = (NI + 7) / 8;                      // ... per wave
    constexpr int D = wg4_slots(CT);                       // ring slots
    constexpr int SLOT = WG4_P * PITCH;                    // bytes
    static_assert(D >= 3 && (D - 1) * IPC < 64, "ring depth / vmcnt range");
    const WgradArgs& p = q.w;
    float* sc = reinterpret_cast<float*>(smem);            // [CW]
    float* sh = sc + WG3_MAXCW;
    const unsigned ring0 = (unsigned)(size_t)smem + WG4_RING0;      // LDS byte address of slot 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;
    const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(p.dy);

    for (int c = tid; c < CW; c += WG3_THREADS) {
        const int cc = q.c0 + c;
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        const int lc = cc - sg.choff;
        const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[cc] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[cc] - mean * scale);
    }

    // ---- loader plan of this lane (slot-invariant): request t of this wave is request I = wave + 8 t of the slot (the last ones
    // of a slot whose count does not divide by 8 are issued twice: same bytes to the same place); lane l moves piece 64 I + l
    const unsigned short* gptr[IPC];
    int gld[IPC], gpix[IPC], gups[IPC];
    unsigned gdst[IPC];
#pragma unroll
    for (int t = 0; t < IPC; ++t) {
        int I = wave + 8 * t;
        if (I >= NI) I = NI - 1;
        gdst[t] = (unsigned)I * 1024u;
        const int piece = I * 64 + lane;
        const int row = piece / PPR, pr = piece - row * PPR;
        gpix[t] = row;
        if (pr >= 16 && pr < 16 + CW / 8) {                // X: 8 channels of the slice
            const int cc = q.c0 + 8 * (pr - 16);
            int s = 0;
            for (int u = 1; u < p.nseg; ++u)
                if (cc >= p.seg[u].choff) s = u;
            const Seg& sg = p.seg[s];
            gptr[t] = reinterpret_cast<const unsigned short*>(sg.x) + (cc - sg.choff);
            gld[t] = sg.ld; gups[t] = sg.ups;
        } else {                                           // dY: 8 output channels (a pad piece repeats piece 0)
            gptr[t] = dy16 + 8 * (pr < 16 ? pr : 0);
            gld[t] = p.lddy; gups[t] = 0;
        }
    }
    const int row_begin = blockIdx.x * q.rows_per_split;
    int row_end = row_begin + q.rows_per_split;
    if (row_end > p.M) row_end = p.M;
    const int nchunks = (row_end - row_begin) / WG4_P;     // (whole slots: launcher precondition)
    auto issue = [&](int chunk) {                          // the IPC requests of this wave for ring slot chunk % D
        const int m0 = row_begin + chunk * WG4_P;
        const unsigned slot = ring0 + (unsigned)(chunk % D) * (unsigned)SLOT;
#pragma unroll
        for (int t = 0; t < IPC; ++t) {
            const int m = m0 + gpix[t];
            int row = m;
            if (q.any_ups && gups[t]) {                    // nearest-upsample index map (models/cu_net.py:250,265), power-of-two geometry
                const int nimg = m >> q.hwshift;
                const int rem = m & (HW - 1);
                row = nimg * (HW >> 2) + ((rem >> q.wshift) >> 1) * (p.W >> 1) + ((rem & (p.W - 1)) >> 1);
            }
            wg4_dma16(gptr[t] + (size_t)row * gld[t], __builtin_amdgcn_readfirstlane(slot + gdst[t]));
        }
    };

    // ---- tile ownership as in the staging kernel: output-channel tile wave & 3; split-K: every X tile, k-step (wave >> 2) of a slot;
    // otherwise half (wave >> 2) owns CTW consecutive X tiles -- the second half the LAST CTW (an odd CT: its first tile repeats the
    // first half's last one and is not stored)
    const int nt = wave & 3;
    const int half = wave >> 2;
    const int cb = SPLITK ? 0 : (half ? CT - CTW : 0);
    const bool dup_first = !SPLITK && half && (CT & 1);
    float bsc[CTW], bsh[CTW];
    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // transpose-read address of this lane inside a slot: pixel 8 (l >> 5) + ((l & 15) >> 2) (+ 4 for the second read, + 16 for the second
    // k-step), channel piece 16 ((l >> 4) & 1) + 4 (l & 3) of the tile
    const unsigned lane_off = (unsigned)((8 * hi + ((lane & 15) >> 2)) * PITCH + 32 * ((lane >> 4) & 1) + 8 * (lane & 3));
    const unsigned off_a = lane_off + (unsigned)(nt * 64);
    const unsigned off_x = lane_off + (unsigned)(256 + cb * 64);

#pragma unroll 1
    for (int c = 0; c < D - 1; ++c)
        if (c < nchunks) issue(c);
    __syncthreads();                                       // sc / sh visible
#pragma unroll
    for (int t = 0; t < CTW; ++t) { bsc[t] = sc[(cb + t) * 32 + li]; bsh[t] = sh[(cb + t) * 32 + li]; }

#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // this wave's requests of slot `chunk` have landed when at most the (D - 2) younger slots' are outstanding (the tail of the
        // loop issues nothing: wait for everything there)
        if (chunk + (D - 2) < nchunks) wg4_wait_dma<(D - 2) * IPC>(); else wg4_wait_dma<0>();
        __syncthreads();                                   // every wave's pieces of this slot are in LDS; slot chunk - 1 is free
        if (chunk + D - 1 < nchunks) issue(chunk + D - 1);
        const unsigned slot = ring0 + (unsigned)(chunk % D) * (unsigned)SLOT;
        constexpr int NK = SPLITK ? 1 : 2;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned kst = (unsigned)((SPLITK ? half : ks) * 16 * PITCH);
            Wg4Frag f[CTW + 1];
            wg4_read_step<CTW, PITCH>(f, slot + off_a + kst, slot + off_x + kst);
#pragma unroll
            for (int t = 0; t < CTW; ++t) {
                f[t + 1].h[0].x = wg4_bn_relu(f[t + 1].h[0].x, bsc[t], bsh[t]);
                f[t + 1].h[0].y = wg4_bn_relu(f[t + 1].h[0].y, bsc[t], bsh[t]);
                f[t + 1].h[1].x = wg4_bn_relu(f[t + 1].h[1].x, bsc[t], bsh[t]);
                f[t + 1].h[1].y = wg4_bn_relu(f[t + 1].h[1].y, bsc[t], bsh[t]);
            }
            const bf16x8_t av = wg4_operand(f[0]);
#pragma unroll
            for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wg4_operand(f[t + 1]), acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                       // (the ring is reused below)

    if (SPLITK) {
        float* red = reinterpret_cast<float*>(smem + WG4_RING0);
        if (half == 1) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((nt * CTW + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((nt * CTW + t) * 16 + r) * 64 + lane];
        }
        if (half == 1) return;
    }
    float* out = q.part + (size_t)blockIdx.x * WG3_NOUT * p.Ccat;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        if (t == 0 && dup_first) continue;                  // the tile both halves computed
        const int c = q.c0 + (cb + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * p.Ccat + c] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6 (planner option wgrad_split_planes): the fp32 1x1 weight gradient on the split contraction with the operands cut ONCE.
// wgrad3_kernel<..., EMU> stages fp32 tiles and every wave cuts the fragments it reads -- the dY fragment by the two waves that share an
// output-channel tile's k-step, every X fragment by the four waves of the four output-channel tiles: 49 152 element cuts per 32-pixel
// chunk for 14 336 elements (CW = 320), ~216 VALU instructions per wave and k-step next to 30 MFMAs: the two pipes of a SIMD are loaded
// about equally, and whatever of the cutting does not hide behind the other wave's MFMAs is lost (38 us per 64 x 64 launch alone against
// a matrix-pipe bound of 26 at 192 workgroups).  Here the element is cut on its way INTO LDS -- once, behind BatchNorm + ReLU -- into three
// bf16 planes laid out as wgrad4_bf16_kernel's ring rows ([pixel][dY 128 | X CW] bf16, the same conflict-free pitch), and the MFMA
// fragments come out of each plane by ds_read_b64_tr_b16 (two reads per fragment: lane = channel, 8 consecutive pixels), through the
// compiler's own builtin (its waits are the compiler's).  Same pieces, same six products per pair in the same k order as the staging
// kernel: bit-identical partial tiles (tests/test_gpu_exact.py).  Two 32-pixel buffers of 3 x 32 x pitch bytes: CW <= 288 (9 tiles) fit
// 160 KB; a 320-channel slice keeps wgrad3_kernel.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 wg5_frag(const char* p, int step) {      // p: this lane's transpose-read address, step = 4 pixel rows
    typedef __attribute__((address_space(3))) v4i16_t lds_v4;
    const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
    const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + step));
    const u32x2w ua = __builtin_bit_cast(u32x2w, a), ub = __builtin_bit_cast(u32x2w, b);
    return u32x4{ua.x, ua.y, ub.x, ub.y};
}
// four fp32 values (4 channels of one pixel) -> their three bf16 pieces, 8 bytes per plane
__device__ __forceinline__ void wg5_cut4(const float4 v, u32x2w& h, u32x2w& m, u32x2w& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_bf16x3_pair(f32x2_op{v.x, v.y}, h0, m0, l0);
    split_bf16x3_pair(f32x2_op{v.z, v.w}, h1, m1, l1);
    h = u32x2w{h0, h1}; m = u32x2w{m0, m1}; l = u32x2w{l0, l1};
}

template <int CT>      // channel tiles of the slice: CW = 32 CT, 4 <= CT <= 9; CT <= 5: split-K over the two k-steps of a chunk
__global__ __launch_bounds__(WG3_THREADS, 2) void wgrad5_split_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool SPLITK = CT <= 5;
    constexpr int CTW = SPLITK ? CT : (CT + 1) / 2;
    static_assert(CTW >= 3 && CTW <= 5, "tile ownerships of 3 .. 5 channel tiles");
    constexpr int CW = 32 * CT;
    constexpr int PITCH = wg4_pitch(CT);                   // bytes per pixel row of a plane
    constexpr int PLANE = WG3_P * PITCH;
    constexpr int BUF = 3 * PLANE;
    const WgradArgs& p = q.w;
    float* sc = reinterpret_cast<float*>(smem);            // [CW]
    float* sh = sc + WG3_MAXCW;
    char* buf0 = smem + WG4_RING0;                         // 2 x { plane h, plane m, plane l }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;

    for (int c = tid; c < CW; c += WG3_THREADS) {
        const int cc = q.c0 + c;
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        const int lc = cc - sg.choff;
        const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[cc] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[cc] - mean * scale);
    }

    // ---- staging plan of this thread (chunk-invariant), as wgrad3_kernel: 2 float4 of dY, up to NX float4 of X
    constexpr int NX = (CT + 1) / 2;                       // ceil(32 * CW / 4 / 512)
    constexpr int cw4 = CW >> 2;
    constexpr int nx4 = WG3_P * cw4;
    int xp[NX], xc[NX];
    const float* xbase[NX];
    int xld[NX], xups[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        int idx = tid + WG3_THREADS * j;
        if (idx >= nx4) idx = nx4 - 1;                     // duplicates of the last item: same value to the same LDS address
        xp[j] = idx / cw4;
        xc[j] = (idx - xp[j] * cw4) << 2;
        const int cc = q.c0 + xc[j];
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cc >= p.seg[t].choff) s = t;
        const Seg& sg = p.seg[s];
        xbase[j] = sg.x + (cc - sg.choff);
        xld[j] = sg.ld;
        xups[j] = sg.ups;
    }
    const int ap0 = tid >> 5, ac0 = (tid & 31) << 2;       // dY item 0: pixel tid/32, channels 4*(tid%32); item 1: pixel + 16

    const int row_begin = blockIdx.x * q.rows_per_split;
    int row_end = row_begin + q.rows_per_split;
    if (row_end > p.M) row_end = p.M;
    const int nchunks = (row_end - row_begin + WG3_P - 1) / WG3_P;

    // (two chunks of raw loads in flight per thread -- stages by chunk parity, the loop unrolled by two, exact counted waits -- measured the same on
    // the 64 x 64 launches, 27.9 vs 28.7 us, and slower on the small ones: a workgroup streams ~21 GB/s from its CU either way, which at 192
    // workgroups IS the launch's share of the HBM rate; one chunk ahead it stays)
    float4 av[2], xv[NX];
    float4 s4[NX], h4[NX];
    bool aok[2], xok[NX];
    auto issue = [&](int chunk) {                          // raw global loads of one chunk into registers (clamped addresses)
        const int m0 = row_begin + chunk * WG3_P;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ap0 + 16 * j;
            aok[j] = m < row_end;
            const int mc = aok[j] ? m : row_begin;
            av[j] = ldg4(p.dy + (size_t)mc * p.lddy + ac0);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int m = m0 + xp[j];
            xok[j] = m < row_end;
            const int mc = xok[j] ? m : row_begin;
            int row = mc;
            if (q.any_ups) {                               // nearest-upsample index map (models/cu_net.py:250,265): (y >> 1, x >> 1)
                const int nimg = mc / HW;
                const int rem = mc - nimg * HW;
                const int py = rem / p.W;
                const int px = rem - py * p.W;
                const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
                row = xups[j] ? rowU : mc;
            }
            xv[j] = ldg4(xbase[j] + (size_t)row * xld[j]);
        }
    };
    auto commit = [&](char* buf) {                         // registers -> three bf16 planes, BatchNorm + ReLU on X, zeros beyond the range
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = aok[j] ? av[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            u32x2w h, m, l;
            wg5_cut4(v, h, m, l);
            char* d = buf + (ap0 + 16 * j) * PITCH + 2 * ac0;
            *reinterpret_cast<u32x2w*>(d) = h;
            *reinterpret_cast<u32x2w*>(d + PLANE) = m;
            *reinterpret_cast<u32x2w*>(d + 2 * PLANE) = l;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            float4 v;
            v.x = fmaxf(fmaf(xv[j].x, s4[j].x, h4[j].x), 0.f);
            v.y = fmaxf(fmaf(xv[j].y, s4[j].y, h4[j].y), 0.f);
            v.z = fmaxf(fmaf(xv[j].z, s4[j].z, h4[j].z), 0.f);
            v.w = fmaxf(fmaf(xv[j].w, s4[j].w, h4[j].w), 0.f);
            if (!xok[j]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            u32x2w h, m, l;
            wg5_cut4(v, h, m, l);
            char* d = buf + xp[j] * PITCH + 256 + 2 * xc[j];
            *reinterpret_cast<u32x2w*>(d) = h;
            *reinterpret_cast<u32x2w*>(d + PLANE) = m;
            *reinterpret_cast<u32x2w*>(d + 2 * PLANE) = l;
        }
    };

    // ---- tile ownership as wgrad4_bf16_kernel: output-channel tile wave & 3; split-K: every X tile, k-step (wave >> 2) of a chunk; otherwise
    // half (wave >> 2) owns CTW consecutive X tiles -- the second half the LAST CTW (an odd CT: its first tile repeats the first half's
    // last one and is not stored)
    const int nt = wave & 3;
    const int half = wave >> 2;
    const int cb = SPLITK ? 0 : (half ? CT - CTW : 0);
    const bool dup_first = !SPLITK && half && (CT & 1);
    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // transpose-read address of this lane inside a plane: pixel 8 (l >> 5) + ((l & 15) >> 2) (+ 4 for the second read, + 16 for the second
    // k-step), channel piece 16 ((l >> 4) & 1) + 4 (l & 3) of the tile
    const int lane_off = (8 * hi + ((lane & 15) >> 2)) * PITCH + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    const int off_a = lane_off + nt * 64;
    const int off_x = lane_off + 256 + cb * 64;

    issue(0);
    __syncthreads();                                       // sc / sh visible
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        s4[j] = *reinterpret_cast<const float4*>(sc + xc[j]);
        h4[j] = *reinterpret_cast<const float4*>(sh + xc[j]);
    }
    commit(buf0);
    __syncthreads();
    auto contract = [&](const char* cur) {                 // the chunk in `cur`: this wave's k-step(s)
        constexpr int NS = SPLITK ? 1 : 2;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const int kst = (SPLITK ? half : ks) * 16 * PITCH;
            // every fragment of the k-step is requested before its first MFMA (left alone hipcc reads a tile's three X fragments and waits
            // for them right in front of that tile's six MFMAs: one exposed LDS round trip per tile)
            const u32x4 ah = wg5_frag(cur + off_a + kst, 4 * PITCH);
            const u32x4 am = wg5_frag(cur + PLANE + off_a + kst, 4 * PITCH);
            const u32x4 al = wg5_frag(cur + 2 * PLANE + off_a + kst, 4 * PITCH);
            u32x4 xh[CTW], xm[CTW], xl[CTW];
#pragma unroll
            for (int t = 0; t < CTW; ++t) {
                xh[t] = wg5_frag(cur + off_x + kst + 64 * t, 4 * PITCH);
                xm[t] = wg5_frag(cur + PLANE + off_x + kst + 64 * t, 4 * PITCH);
                xl[t] = wg5_frag(cur + 2 * PLANE + off_x + kst + 64 * t, 4 * PITCH);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < CTW; ++t) acc[t] = mfma_split6(ah, am, al, xh[t], xm[t], xl[t], acc[t]);
        }
    };
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) issue(chunk + 1);                        // in flight across the MFMA loop
        contract(buf0 + (chunk & 1) * BUF);
        if (more) commit(buf0 + ((chunk + 1) & 1) * BUF);
        __syncthreads();
    }

    if (SPLITK) {                                          // the second k-step's half hands its tiles over through LDS
        float* red = reinterpret_cast<float*>(smem + WG4_RING0);
        if (half == 1) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((nt * CTW + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int t = 0; t < CTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((nt * CTW + t) * 16 + r) * 64 + lane];
        }
        if (half == 1) return;
    }
    float* out = q.part + (size_t)blockIdx.x * WG3_NOUT * p.Ccat;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        if (t == 0 && dup_first) continue;                  // the tile both halves computed
        const int c = q.c0 + (cb + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * p.Ccat + c] = acc[t][r];
        }
    }
}

static hipError_t launch_wg5_split(const Wg3Args& q, int ct, dim3 grid, hipStream_t s) {
    const size_t smem = (size_t)WG4_RING0 + (size_t)2 * 3 * WG3_P * wg4_pitch(ct);
#define CUNET_WG5(CT_) hipLaunchKernelGGL((wgrad5_split_kernel<CT_>), grid, dim3(WG3_THREADS), smem, s, q)
    switch (ct) {
        case 4: CUNET_WG5(4); break;
        case 5: CUNET_WG5(5); break;
        case 6: CUNET_WG5(6); break;
        case 7: CUNET_WG5(7); break;
        case 8: CUNET_WG5(8); break;
        case 9: CUNET_WG5(9); break;
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG5
    return hipGetLastError();
}

// dst[i] = sum_s part[s][i]: fixed summation order (bitwise reproducible).  blockIdx.y = table entry.
// A block covers 64 float4 of the output; its four 64-thread groups take every fourth split and meet in LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgReduceEntry* __restrict__ tab, const float* __restrict__ ws,
                                                           float* __restrict__ grads) {
    const WgReduceEntry e = tab[blockIdx.y];
    if (e.S <= 0) return;                               // this node ran on an atomic kernel in the current mode
    const int n4 = e.numel >> 2;
    const int i4 = blockIdx.x * 64 + (threadIdx.x & 63);
    if (blockIdx.x * 64 >= n4) return;
    __shared__ float4 red[256];
    const int sg = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
        const float* src = ws + e.part + (size_t)i4 * 4;
        for (int s = sg; s < e.S; s += 4) {
            const float4 v = ldg4(src + (size_t)s * e.numel);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && i4 < n4) {
        const float4 b = red[64 + threadIdx.x], c = red[128 + threadIdx.x], d = red[192 + threadIdx.x];
        acc.x = (acc.x + b.x) + (c.x + d.x); acc.y = (acc.y + b.y) + (c.y + d.y);
        acc.z = (acc.z + b.z) + (c.z + d.z); acc.w = (acc.w + b.w) + (c.w + d.w);
        if (e.taps > 1) {                              // [tap][n*C + c] -> [n*C + c][tap]
            const int per = e.numel / e.taps;
            const int i = i4 * 4;
            const int tap = i / per, nc = i - tap * per;
            float* d0 = grads + e.dst + (size_t)nc * e.taps + tap;
            d0[0] = acc.x; d0[e.taps] = acc.y; d0[2 * e.taps] = acc.z; d0[3 * e.taps] = acc.w;
        } else {
            *reinterpret_cast<float4*>(grads + e.dst + (size_t)i4 * 4) = acc;
        }
    }
}

bool wgrad3_supported(const WgradArgs& a) {
    if (a.taps != 1 || a.Cout != WG3_NOUT || a.lddy != WG3_NOUT || a.img != nullptr) return false;
    if (a.Ccat % 32 || a.Ccat < 128) return false;
    const int al = a.xbf16 == 2 ? 8 : 4;                  // 16-byte pieces: 8 bf16 / 4 fp32 channels
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % al || a.seg[i].ld % al) return false;
    return true;
}

static hipError_t launch_wg4_bf16(const Wg3Args& q, int ct, dim3 grid, hipStream_t s) {
    const size_t smem = (size_t)WG4_RING0 + (size_t)wg4_slots(ct) * WG4_P * wg4_pitch(ct);
#define CUNET_WG4(CT_) hipLaunchKernelGGL((wgrad4_bf16_kernel<CT_>), grid, dim3(WG3_THREADS), smem, s, q)
    switch (ct) {
        case 4: CUNET_WG4(4); break;
        case 5: CUNET_WG4(5); break;
        case 6: CUNET_WG4(6); break;
        case 7: CUNET_WG4(7); break;
        case 8: CUNET_WG4(8); break;
        case 9: CUNET_WG4(9); break;
        case 10: CUNET_WG4(10); break;
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG4
    return hipGetLastError();
}

static hipError_t launch_wg3_bf16(const Wg3Args& q, int ct, dim3 grid, size_t smem, hipStream_t s) {
#define CUNET_WG3B(CTW_, SK_) hipLaunchKernelGGL((wgrad3_bf16_kernel<CTW_, SK_>), grid, dim3(WG3_THREADS), smem, s, q)
    switch (ct) {
        case 4: CUNET_WG3B(4, true); break;
        case 5: CUNET_WG3B(5, true); break;
        case 6: CUNET_WG3B(3, false); break;
        case 7: case 8: CUNET_WG3B(4, false); break;
        case 9: case 10: CUNET_WG3B(5, false); break;
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG3B
    return hipGetLastError();
}

template <int XB, bool EMU = false>
static hipError_t launch_wg3_x(const Wg3Args& q, int ct, dim3 grid, size_t smem, hipStream_t s) {
#define CUNET_WG3(CTW_, SK_) hipLaunchKernelGGL((wgrad3_kernel<CTW_, SK_, XB, EMU>), grid, dim3(WG3_THREADS), smem, s, q)
    switch (ct) {
        case 4: CUNET_WG3(4, true); break;
        case 5: CUNET_WG3(5, true); break;
        case 6: CUNET_WG3(3, false); break;
        case 7: case 8: CUNET_WG3(4, false); break;
        case 9: case 10: CUNET_WG3(5, false); break;
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG3
    return hipGetLastError();
}

// part: [S][128][Ccat] floats.  Slices of at most 320 channels (more: several launches, dY is then re-read per slice).
hipError_t launch_wgrad3(const WgradArgs& a, float* part, int S, int rows_per_split, hipStream_t s) {
    if (!wgrad3_supported(a) || S < 1 || rows_per_split % (a.xbf16 == 2 ? WG3B_P : WG3_P)) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[] = {
            (const void*)&wgrad3_kernel<4, true, 0>, (const void*)&wgrad3_kernel<5, true, 0>, (const void*)&wgrad3_kernel<3, false, 0>,
            (const void*)&wgrad3_kernel<4, false, 0>, (const void*)&wgrad3_kernel<5, false, 0>,
            (const void*)&wgrad3_kernel<4, true, 1>, (const void*)&wgrad3_kernel<5, true, 1>, (const void*)&wgrad3_kernel<3, false, 1>,
            (const void*)&wgrad3_kernel<4, false, 1>, (const void*)&wgrad3_kernel<5, false, 1>,
            (const void*)&wgrad3_kernel<4, true, 0, true>, (const void*)&wgrad3_kernel<5, true, 0, true>, (const void*)&wgrad3_kernel<3, false, 0, true>,
            (const void*)&wgrad3_kernel<4, false, 0, true>, (const void*)&wgrad3_kernel<5, false, 0, true>,
            (const void*)&wgrad3_bf16_kernel<4, true>, (const void*)&wgrad3_bf16_kernel<5, true>, (const void*)&wgrad3_bf16_kernel<3, false>,
            (const void*)&wgrad3_bf16_kernel<4, false>, (const void*)&wgrad3_bf16_kernel<5, false>,
            (const void*)&wgrad4_bf16_kernel<4>, (const void*)&wgrad4_bf16_kernel<5>, (const void*)&wgrad4_bf16_kernel<6>, (const void*)&wgrad4_bf16_kernel<7>,
            (const void*)&wgrad4_bf16_kernel<8>, (const void*)&wgrad4_bf16_kernel<9>, (const void*)&wgrad4_bf16_kernel<10>,
            (const void*)&wgrad5_split_kernel<4>, (const void*)&wgrad5_split_kernel<5>, (const void*)&wgrad5_split_kernel<6>, (const void*)&wgrad5_split_kernel<7>,
            (const void*)&wgrad5_split_kernel<8>, (const void*)&wgrad5_split_kernel<9>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    Wg3Args q{};
    q.w = a;
    q.part = part;
    q.rows_per_split = rows_per_split;
    q.any_ups = 0;
    for (int i = 0; i < a.nseg; ++i) q.any_ups |= a.seg[i].ups;
    {
        auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (v > 0 && (1 << l) == v) ? l : -1; };
        const int lw = lg2(a.W), lhw = lg2(a.H * a.W);
        q.wshift = (lw >= 0 && lhw >= 0) ? lw : -1;
        q.hwshift = (lw >= 0 && lhw >= 0) ? lhw : -1;
    }
    // bf16 x and dY: the LDS-DMA ring kernel (wgrad4_bf16_kernel) when every pixel range is whole ring slots and the up-sample map is
    // shifts; the register-staged kernel otherwise (ragged ranges: N * H * W not a multiple of 32 at the bottom of a small batch)
    const bool dma = a.xbf16 == 2 && a.bf16_dma && a.M % WG4_P == 0 && rows_per_split % WG4_P == 0 && (!q.any_ups || q.wshift >= 0);
    const int ct_all = a.Ccat / 32;
    const int nslices = (ct_all + 9) / 10;
    const int per = (ct_all + nslices - 1) / nslices;
    for (int c0t = 0; c0t < ct_all; c0t += per) {
        int ct = ct_all - c0t < per ? ct_all - c0t : per;
        if (ct < 4) { c0t -= 4 - ct; ct = 4; }          // a short tail slice overlaps its predecessor (same values written twice)
        q.c0 = c0t * 32;
        q.CW = ct * 32;
        const int ldx = (ct <= 5 ? ct : 2 * ((ct + 1) / 2)) * 32;          // the fp32 kernel's compile-time X pitch
        size_t buf_bytes = a.xbf16 == 2 ? (size_t)2 * (WG3_NOUT + q.CW) * WG3B_LDP * 2 : (size_t)2 * WG3_P * (WG3_NOUT + ldx) * 4;
        if (ct <= 5 && buf_bytes < (size_t)4 * ct * 4096) buf_bytes = (size_t)4 * ct * 4096;      // split-K hand-over area
        const size_t smem = (size_t)2 * WG3_MAXCW * 4 + buf_bytes;
        // fp32 on the split contraction, operands cut once on the way into LDS (planner option wgrad_split_planes): slices of <= 8 tiles
        const bool planes = a.split && a.xbf16 == 0 && a.split_planes && ct <= 8 &&      // (9 tiles: 256 VGPRs and 11 spilled)
                            (size_t)WG4_RING0 + (size_t)2 * 3 * WG3_P * wg4_pitch(ct) <= (size_t)160 * 1024 && (ct > 5 || (size_t)4 * ct * 4096 <= (size_t)2 * 3 * WG3_P * wg4_pitch(ct));
        const hipError_t e = planes ? launch_wg5_split(q, ct, dim3(S), s)
                           : dma ? launch_wg4_bf16(q, ct, dim3(S), s) : a.xbf16 == 2 ? launch_wg3_bf16(q, ct, dim3(S), smem, s)
                           : a.xbf16 ? launch_wg3_x<1>(q, ct, dim3(S), smem, s)
                           : a.split ? launch_wg3_x<0, true>(q, ct, dim3(S), smem, s) : launch_wg3_x<0>(q, ct, dim3(S), smem, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_wgrad_reduce(const WgReduceEntry* tab, int n, int max_numel, const float* ws, float* grads, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int gx = (max_numel / 4 + 63) / 64;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, n), dim3(256), 0, s, tab, ws, grads);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 weight gradient on the same principles (autograd wgrad of models/cu_net.py:47, 128 -> 32 channels, pad 1):
//     dW[n][c][dy][dx] = sum_{img,y,x} dY[img][y][x][n] * relu(bn(X))[img][y+dy][x+dx][c]
// The first-generation kernel (wgrad_kernel<WG_3X3>) issues 10 per-lane global loads for every 9 MFMAs and re-reads X
// nine times and dY four times: 23 % of the fp32 MFMA peak.  Here a 512-thread workgroup walks a range of image ROWS:
// the activated rows y-1, y, y+1 live in an LDS ring of three [W+2 pixels][128 channels] slots (zero columns left and
// right, an all-zero slot for rows outside the image), so a tap is nothing but an LDS offset; row y+2 and the next dY row
// are fetched from HBM while the MFMAs of row y run and replace the slot of row y-1 afterwards.  Every element of X and
// dY is read from HBM once (plus one halo row per workgroup).  Wave w owns input-channel tile (w & 3) and taps 0..4
// (w < 4) or 5..8: <= 5 accumulators, A fragment shared by its taps.  Partial tiles go to part[split][tap][n][c]; the
// bucket's reduce kernel sums the splits and transposes into torch's [n][c][tap].
constexpr int WG3C_C = 128, WG3C_N = 32;

// EMU (planner option f32_split, XBG = 0, W a multiple of 16): the contraction on the bf16 matrix pipe as in wgrad3_kernel -- a lane
// takes 8 consecutive pixels of its channel, cuts them into three bf16 pieces, six MFMAs per (dY tile, tap).
template <int XBG, bool EMU = false>      // 0: fp32 x and dY; 1: bf16 x; 2: bf16 x and bf16 dY (both widened to fp32 on the way into LDS; fp32 MFMA)
__global__ __launch_bounds__(WG3_THREADS, WG3_MIN_WAVES) void wgrad3_3x3_kernel(const Wg3Args q) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgradArgs& p = q.w;
    const int W = p.W, H = p.H;
    const int SLOT = (W + 2) * WG3C_C;                     // floats per ring slot
    float* sc = reinterpret_cast<float*>(smem);            // [128]
    float* sh = sc + WG3C_C;
    float* ring = sh + WG3C_C;                             // 3 slots + the zero slot
    float* zslot = ring + 3 * SLOT;
    float* dyb = zslot + SLOT;                             // 2 x [W][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg& sg = p.seg[0];

    for (int c = tid; c < WG3C_C; c += WG3_THREADS) {
        const double sum = sg.stats[c], sq = sg.stats[sg.C + c];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
    }
    for (int i = tid; i < 4 * SLOT / 4; i += WG3_THREADS)   // zero everything once: pad columns and the zero slot stay zero
        reinterpret_cast<float4*>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int NH = p.M / W;                                // image rows in the batch
    const int g_begin = blockIdx.x * q.rows_per_split;     // rows_per_split counts IMAGE ROWS here
    int g_end = g_begin + q.rows_per_split;
    if (g_end > NH) g_end = NH;

    // staging plan: a row of X is W * 32 float4 (4 per thread at W = 64), a row of dY W * 8 float4
    constexpr int NXR = 4;
    const int nx4 = W * (WG3C_C / 4), na4 = W * (WG3C_N / 4);
    float4 xv[NXR], av;
    float4 s4[NXR], h4[NXR];
    int xpix[NXR], xc4[NXR];
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
        int idx = tid + WG3_THREADS * j;
        if (idx >= nx4) idx = nx4 - 1;
        xpix[j] = idx >> 5;                                // 32 float4 per pixel
        xc4[j] = (idx & 31) << 2;
        s4[j] = *reinterpret_cast<const float4*>(sc + xc4[j]);
        h4[j] = *reinterpret_cast<const float4*>(sh + xc4[j]);
    }
    const int aidx = tid < na4 ? tid : na4 - 1;
    bool xrow_ok = false, arow_ok = false;
    auto issue_x = [&](int g) {                            // row g of X (global image-row index) -> registers
        xrow_ok = g >= 0 && g < NH;
        const size_t base = (size_t)(xrow_ok ? g : 0) * W;
#pragma unroll
        for (int j = 0; j < NXR; ++j) xv[j] = ldx4<XB>(sg.x, (base + xpix[j]) * sg.ld + xc4[j]);
    };
    auto commit_x = [&](int g) {                           // registers -> slot (g mod 3), pixels 1 .. W, BatchNorm + ReLU
        if (!xrow_ok) return;
        float* slot = ring + (size_t)(((g % 3) + 3) % 3) * SLOT;
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            float4 v;
            v.x = fmaxf(fmaf(xv[j].x, s4[j].x, h4[j].x), 0.f);
            v.y = fmaxf(fmaf(xv[j].y, s4[j].y, h4[j].y), 0.f);
            v.z = fmaxf(fmaf(xv[j].z, s4[j].z, h4[j].z), 0.f);
            v.w = fmaxf(fmaf(xv[j].w, s4[j].w, h4[j].w), 0.f);
            if (p.qin_bits) {
                v.x = quan_input_act(v.x, p.qin_bits); v.y = quan_input_act(v.y, p.qin_bits);
                v.z = quan_input_act(v.z, p.qin_bits); v.w = quan_input_act(v.w, p.qin_bits);
            }
            *reinterpret_cast<float4*>(slot + (xpix[j] + 1) * WG3C_C + xc4[j]) = v;
        }
    };
    auto issue_a = [&](int g) {
        arow_ok = g < g_end;
        av = ldx4<GB>(p.dy, ((size_t)(arow_ok ? g : g_begin) * W + (aidx >> 3)) * p.lddy + ((aidx & 7) << 2));
    };
    auto commit_a = [&](int g) {
        if (!arow_ok) return;
        *reinterpret_cast<float4*>(dyb + (size_t)(g & 1) * W * WG3C_N + (aidx >> 3) * WG3C_N + ((aidx & 7) << 2)) = av;
    };

    // ---- tile ownership
    const int ctile = wave & 3;
    const int half = wave >> 2;
    constexpr int CTW = 5;
    int tdy[CTW], tdx[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        int tap = half * 5 + t;
        if (tap > 8) tap = 8;                              // the second half owns 4 taps: its 5th slot repeats tap 8 and is not stored
        tdy[t] = tap / 3 - 1;
        tdx[t] = tap - (tap / 3) * 3 - 1;
    }
    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- prologue: rows g_begin-1, g_begin, g_begin+1 of X and row g_begin of dY
    for (int g = g_begin - 1; g <= g_begin + 1; ++g) { issue_x(g); commit_x(g); }
    issue_a(g_begin); commit_a(g_begin);
    __syncthreads();

    for (int g = g_begin; g < g_end; ++g) {
        const int y = g % H;
        issue_x(g + 2);                                    // in flight across this row's MFMAs
        issue_a(g + 1);
        // LDS operands are addressed as INTEGER offsets from the one shared base: a pointer selected at run time between
        // ring slots and the zero slot loses its address space, hipcc then emits flat_load and waits vmcnt(0) -- i.e. for
        // the next row's HBM loads -- in front of every MFMA (measured: 18 us per image row instead of 9)
        float* lds = reinterpret_cast<float*>(smem);
        const int ring0 = 2 * WG3C_C, zoff = ring0 + 3 * SLOT, dyoff = zoff + SLOT;
        const int aoff = dyoff + (g & 1) * W * WG3C_N + hi * WG3C_N + li;
        int rowoff[3];
        rowoff[0] = (y > 0) ? ring0 + ((g + 2) % 3) * SLOT : zoff;                // slot of row g-1
        rowoff[1] = ring0 + (g % 3) * SLOT;
        rowoff[2] = (y < H - 1) ? ring0 + ((g + 1) % 3) * SLOT : zoff;
        int boff[CTW];
#pragma unroll
        for (int t = 0; t < CTW; ++t) boff[t] = rowoff[tdy[t] + 1] + (hi + tdx[t] + 1) * WG3C_C + ctile * 32 + li;
        if constexpr (EMU) {
            const int ns = W >> 4;                         // k-steps of 16 pixels
            for (int ks = 0; ks < ns; ++ks) {
                // this lane's pixels 16 ks + 8 hi .. + 7 (aoff / boff carry hi pixels already: 7 hi more)
                const int pa = aoff + (16 * ks + 7 * hi) * WG3C_N;
                float af[8], bf[CTW][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = lds[pa + j * WG3C_N];
#pragma unroll
                for (int t = 0; t < CTW; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[t][j] = lds[boff[t] + (16 * ks + 7 * hi + j) * WG3C_C];
                u32x4 ah, am, al;
                split_bf16x3(af, ah, am, al);
#pragma unroll
                for (int t = 0; t < CTW; ++t) {
                    u32x4 xh, xm, xl;
                    split_bf16x3(bf[t], xh, xm, xl);
                    acc[t] = mfma_split6(ah, am, al, xh, xm, xl, acc[t]);
                }
            }
            __syncthreads();
            commit_x(g + 2);
            commit_a(g + 1);
            __syncthreads();
            continue;
        }
        const int nk = W >> 1;
        float a_cur = lds[aoff], b_cur[CTW];
#pragma unroll
        for (int t = 0; t < CTW; ++t) b_cur[t] = lds[boff[t]];
        for (int kk = 0; kk < nk; ++kk) {
            float a_nxt, b_nxt[CTW];
            const int kn = (kk + 1 < nk) ? kk + 1 : kk;
            a_nxt = lds[aoff + 2 * kn * WG3C_N];
#pragma unroll
            for (int t = 0; t < CTW; ++t) b_nxt[t] = lds[boff[t] + 2 * kn * WG3C_C];
#pragma unroll
            for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + CTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CTW, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < CTW; ++t) b_cur[t] = b_nxt[t];
        }
        __syncthreads();                                   // everyone is done with row g-1's slot and this dY buffer's twin
        commit_x(g + 2);
        commit_a(g + 1);
        __syncthreads();
    }

    // ---- partial tiles: part[split][tap][n][c]
    float* out = q.part + (size_t)blockIdx.x * 9 * WG3C_N * WG3C_C;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        const int tap = half * 5 + t;
        if (tap > 8) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[((size_t)tap * WG3C_N + n) * WG3C_C + ctile * 32 + li] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The 3x3 weight gradient on bf16 MFMA (bf16 x AND bf16 dY: FusedTrainer(bf16_grads=True)).  v_mfma_f32_32x32x16_bf16
// contracts 16 PIXELS per instruction and wants a lane's 8 pixels contiguous, so the rows live in LDS channel-major:
//     Xs[slot][c][W (+8 pad)]   activated (BatchNorm + ReLU in fp32, re-rounded: the operand the bf16 forward multiplied)
//     dYs[buf][dx][n][W (+8)]   THREE copies of the dY row, shifted by dx = -1, 0, +1 pixels with zeros shifted in
// so that   dW[dy][dx][n][c] = sum_q dYs[dx][n][q] * Xs[row y+dy][c][q]   is a contraction over ALIGNED 16-byte fragments for
// every tap: the x-shift is paid once per dY row (32 channels, 16 lane shuffles) instead of per tap, the y-shift is a ring
// slot, rows outside the image are an all-zero slot.  The transposition happens on the way into LDS exactly as in
// wgrad3_bf16_kernel (a lane owns 4 pixels x 8 channels; v_cvt_pk_bf16_f32 of the same channel of two neighbouring pixels
// is the pixel-major packing).  Ring of FOUR X slots: row g+2 is written while rows g-1, g, g+1 are read -> one barrier per
// image row; global loads run two rows ahead of their LDS commit (two register sets).  Per image row a workgroup moves
// 20 KB from HBM for 20 MFMAs of 32 cycles per wave: HBM-bound by a wide margin (AI = 230 flop/B at bf16 ridge ~310), the
// fp32-MFMA ring it replaces in this mode is MFMA-bound at 1/16 of the rate.
// Wave w: input-channel tile (w & 3), taps 0..4 (w < 4; rows y-1, y) or 5..8 (rows y, y+1).  Partials as above.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// The register sets of wgrad3_3x3_bf16_kernel are loaded with inline asm and retired with a COUNTED s_waitcnt: hipcc's own
// waitcnt insertion gives up on this loop (per-role exec branches) and waits vmcnt(0) before every commit, i.e. also for the
// set that was issued one row ago -- which halves the distance between a load and its use.
__device__ __forceinline__ u32x4 gload16_asm(const unsigned short* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void vm_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int LOGQG>     // log2(W / 4): W = 16, 32, 64
__global__ __launch_bounds__(WG3_THREADS, 2) void wgrad3_3x3_bf16_kernel(const Wg3Args q) {
    constexpr int QG = 1 << LOGQG;            // 4-pixel groups per image row
    constexpr int W = 4 * QG;
    constexpr int LDPX = W + 8;               // bf16 elements per LDS row: pitch W*2 + 16 bytes (conflict-free ds_read_b128)
    constexpr int NXW = QG / 4 > 0 ? QG / 4 : 1;      // waves that stage an X row (a wave covers 64 / QG channel groups of 8)
    constexpr int SLOT = WG3C_C * LDPX;       // elements per X slot
    constexpr int DYB = 3 * WG3C_N * LDPX;    // elements per dY buffer (3 shifted copies)
    constexpr int NK = W / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgradArgs& p = q.w;
    const int H = p.H;
    float* sc = reinterpret_cast<float*>(smem);            // [128]
    float* sh = sc + WG3C_C;
    unsigned short* lds16 = reinterpret_cast<unsigned short*>(smem);
    constexpr int RING0 = 2 * WG3C_C * 2;                  // element offset of slot 0 (after sc / sh: 1 KB)
    constexpr int ZOFF = RING0 + 4 * SLOT;
    constexpr int DYOFF = ZOFF + SLOT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg& sg = p.seg[0];

    for (int c = tid; c < WG3C_C; c += WG3_THREADS) {
        const double sum = sg.stats[c], sq = sg.stats[sg.C + c];
        const double mean = sum / sg.count;
        double var = sq / sg.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
    }
    for (int i = tid; i < SLOT / 8; i += WG3_THREADS)       // the zero slot
        reinterpret_cast<uint4*>(lds16 + ZOFF)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const int NH = p.M / W;                                // image rows in the batch
    const int g_begin = blockIdx.x * q.rows_per_split;
    int g_end = g_begin + q.rows_per_split;
    if (g_end > NH) g_end = NH;

    // ---- staging roles.  X: waves 0 .. NXW-1, lane = (pixel group q4, channel group); dY: wave NXW, channel groups 0..3
    const int q4 = lane & (QG - 1);
    const int cgl = lane >> LOGQG;                         // channel group inside the wave
    const bool is_x = wave < NXW;
    const int cgx = wave * (64 / QG) + cgl;                // 0 .. 15 (8 channels each)
    const bool is_a = wave == NXW && cgl < 4;
    const unsigned short* x16 = reinterpret_cast<const unsigned short*>(sg.x);
    const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(p.dy);
    float s8[8], h8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s8[e] = sc[(is_x ? cgx * 8 : 0) + e]; h8[e] = sh[(is_x ? cgx * 8 : 0) + e]; }

    // two register sets (compile-time selected): loads run two rows ahead of their commit
    u32x4 ld0[4], ld1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ld0[i] = u32x4{0u, 0u, 0u, 0u}; ld1[i] = ld0[i]; }
    bool ok0 = false, ok1 = false;
    // one load site for both roles (a pending-load state that differs between paths makes hipcc wait for the NEWER set too)
    const bool stager = is_x || is_a;
    const unsigned short* gsrc = is_x ? x16 + 8 * cgx : dy16 + 8 * cgl;
    const size_t gld = is_x ? (size_t)sg.ld : (size_t)p.lddy;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto issue = [&](auto SET, int gx, int ga) {           // X row gx (waves < NXW) / dY row ga (wave NXW)
        constexpr int ST = decltype(SET)::value;
        u32x4 (&L)[4] = *(ST ? &ld1 : &ld0);
        bool& okv = *(ST ? &ok1 : &ok0);
        const int g = is_x ? gx : ga;
        okv = is_x ? (gx >= 0 && gx < NH) : (is_a && ga < g_end);
        if (stager) {
            const unsigned short* src = gsrc + ((size_t)(okv ? g : g_begin) * W + 4 * q4) * gld;
#pragma unroll
            for (int i = 0; i < 4; ++i) L[i] = gload16_asm(src + (size_t)i * gld);
        }
    };
    auto commit = [&](auto SET, auto NEWER, int gx, int ga) {      // NEWER: loads issued after this set's that may stay in flight
        constexpr int ST = decltype(SET)::value;
        u32x4 (&L)[4] = *(ST ? &ld1 : &ld0);
        const bool okv = ST ? ok1 : ok0;
        if (!stager) return;
        vm_wait4<decltype(NEWER)::value>(L[0], L[1], L[2], L[3]);
        unsigned r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[i][0] = L[i][0]; r[i][1] = L[i][1]; r[i][2] = L[i][2]; r[i][3] = L[i][3]; }
        if (!okv) return;                                  // (after the registers were consumed: the loads are retired on every path)
        if (is_x) {
            uint2* dst = reinterpret_cast<uint2*>(lds16 + RING0 + (gx & 3) * SLOT + (8 * cgx) * LDPX + 4 * q4);
            float v[8][4];                                 // [channel][pixel]
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * k][i] = fmaxf(fmaf(bf16_bits_lo(r[i][k]), s8[2 * k], h8[2 * k]), 0.f);
                    v[2 * k + 1][i] = fmaxf(fmaf(bf16_bits_hi(r[i][k]), s8[2 * k + 1], h8[2 * k + 1]), 0.f);
                }
            if (p.qin_bits) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[e][i] = quan_input_act(v[e][i], p.qin_bits);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e * (LDPX / 4)] = make_uint2(cvt_pk_bf16(v[e][0], v[e][1]), cvt_pk_bf16(v[e][2], v[e][3]));
        } else {
            uint2* dst = reinterpret_cast<uint2*>(lds16 + DYOFF + (ga & 1) * DYB + (8 * cgl) * LDPX + 4 * q4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int par = 0; par < 2; ++par) {        // channel 2k + par: pixels 0..3 as two dwords
                    const unsigned a0 = par ? ((r[0][k] >> 16) | (r[1][k] & 0xffff0000u)) : ((r[0][k] & 0xffffu) | (r[1][k] << 16));
                    const unsigned a1 = par ? ((r[2][k] >> 16) | (r[3][k] & 0xffff0000u)) : ((r[2][k] & 0xffffu) | (r[3][k] << 16));
                    unsigned left = __shfl_up(a1, 1), right = __shfl_down(a0, 1);      // neighbouring pixel groups are neighbouring lanes
                    if (q4 == 0) left = 0u;
                    if (q4 == QG - 1) right = 0u;
                    const int e = 2 * k + par;
                    // copy dxi holds dY[q - dx], dx = dxi - 1
                    dst[(0 * WG3C_N + e) * (LDPX / 4)] = make_uint2((a0 >> 16) | (a1 << 16), (a1 >> 16) | (right << 16));      // dx = -1: dY[q + 1]
                    dst[(1 * WG3C_N + e) * (LDPX / 4)] = make_uint2(a0, a1);
                    dst[(2 * WG3C_N + e) * (LDPX / 4)] = make_uint2((left >> 16) | (a0 << 16), (a0 >> 16) | (a1 << 16));        // dx = +1: dY[q - 1]
                }
            }
        }
    };

    const int ctile = wave & 3;
    const int half = wave >> 2;
    constexpr int CTW = 5;
    f32x16 acc[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- prologue: X rows g_begin-1 .. g_begin+1 and dY row g_begin committed; X row g_begin+2 / dY row g_begin+1 in set 1
    issue(S0{}, g_begin - 1, g_begin);
    issue(S1{}, g_begin, g_end);                           // (no dY row: ok = false)
    using N0 = std::integral_constant<int, 0>;
    using N4 = std::integral_constant<int, 4>;
    commit(S0{}, N4{}, g_begin - 1, g_begin);
    issue(S0{}, g_begin + 1, g_end);
    commit(S1{}, N4{}, g_begin, 0);
    commit(S0{}, N0{}, g_begin + 1, 0);
    issue(S1{}, g_begin + 2, g_begin + 1);
    __syncthreads();

    const int frag = li * LDPX + 8 * hi;                   // this lane's fragment inside a [32][LDPX] tile
    auto row = [&](auto HALF, int g) {
        constexpr int HF = decltype(HALF)::value;
        const int y = g % H;
        const int abase = DYOFF + (g & 1) * DYB + frag;
        // B rows this half needs: HF = 0 -> y-1 (or zero), y;  HF = 1 -> y, y+1 (or zero)
        int b0, b1;
        if (HF == 0) { b0 = (y > 0) ? RING0 + ((g - 1) & 3) * SLOT : ZOFF; b1 = RING0 + (g & 3) * SLOT; }
        else { b0 = RING0 + (g & 3) * SLOT; b1 = (y < H - 1) ? RING0 + ((g + 1) & 3) * SLOT : ZOFF; }
        b0 += ctile * 32 * LDPX + frag;
        b1 += ctile * 32 * LDPX + frag;
        bf16x8_t a[2][3], b[2][2];
        auto fetch = [&](int st, int j) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a[st][d] = *reinterpret_cast<const bf16x8_t*>(lds16 + abase + d * WG3C_N * LDPX + 16 * j);
            b[st][0] = *reinterpret_cast<const bf16x8_t*>(lds16 + b0 + 16 * j);
            b[st][1] = *reinterpret_cast<const bf16x8_t*>(lds16 + b1 + 16 * j);
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            if (j + 1 < NK) fetch((j + 1) & 1, j + 1);
            const int st = j & 1;
            if (HF == 0) {                                 // taps 0..4 = (dy -1: dx -1, 0, +1), (dy 0: dx -1, 0)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][0], b[st][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][1], b[st][0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][2], b[st][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][0], b[st][1], acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][1], b[st][1], acc[4], 0, 0, 0);
            } else {                                       // taps 5..8 = (dy 0: dx +1), (dy +1: dx -1, 0, +1)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][2], b[st][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][0], b[st][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][1], b[st][1], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st][2], b[st][1], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, HF == 0 ? 5 : 4, 0);
        }
    };
    auto step = [&](auto PAR, int g) {                     // PAR = g - g_begin parity: set PAR^1 holds X row g+2 / dY row g+1
        using P1 = std::integral_constant<int, decltype(PAR)::value ^ 1>;
        commit(P1{}, N4{}, g + 2, g + 1);                  // slot (g+2)&3 and dY buffer (g+1)&1 are not read during this row
        issue(P1{}, g + 4, g + 3);
        if (half == 0) row(S0{}, g);
        else row(S1{}, g);
        __syncthreads();
    };
    // register-set schedule: at row g (relative index i = g - g_begin) set (i&1)^1 holds (X g+2, dY g+1), set (i&1) holds (X g+3, dY g+2)
    // after the prologue: set 1 = (X g_begin+2, dY g_begin+1); set 0 must get (X g_begin+3, dY g_begin+2)
    issue(S0{}, g_begin + 3, g_begin + 2);
    for (int g = g_begin; g < g_end; g += 2) {
        step(S0{}, g);
        step(S1{}, g + 1);                                 // (the launcher guarantees an even number of rows per workgroup)
    }

    // ---- partial tiles: part[split][tap][n][c]
    float* out = q.part + (size_t)blockIdx.x * 9 * WG3C_N * WG3C_C;
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        const int tap = half * 5 + t;
        if (tap > 8) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[((size_t)tap * WG3C_N + n) * WG3C_C + ctile * 32 + li] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The stem's 7x7 / stride-2 weight gradient (autograd wgrad of models/cu_net.py:300; K = 3*7*7 = 147, Cout = 128):
//     dW[n][c][ky][kx] = sum_{img,oy,ox} dY[img][oy][ox][n] * X[img][c][2 oy + ky - 3][2 ox + kx - 3]
// It is the LAST kernel of a step and runs alone, so its duration is on the critical path.  wgrad2_stem_kernel gathers
// the im2col operand per lane from the NCHW image (one 4-byte load per MFMA, dY read twice, atomic commit: 292 us,
// 51 TFLOP/s).  Here a 512-thread workgroup owns `rows` consecutive output rows of ONE image: the 2 rows + 5 input rows
// they touch are staged ONCE in LDS (zero borders written as zeros, so a tap is an LDS offset), dY is streamed through
// LDS in chunks of 64 output pixels (next chunk's global loads in flight across the MFMAs), wave w owns output-channel tile
// (w & 3) and k-tiles {0,1,2} (w < 4) or {3,4}; the im2col operand of lane k is ONE conflict-free ds_read_b32 (stem_cp /
// stem_rp in common.h).  Partial tiles part[split][n][147], summed by the stem bucket's reduce: no atomics.
template <int HF> struct StemTiles { static constexpr int N = HF == 0 ? 3 : 2; static constexpr int K0 = HF == 0 ? 0 : 3; };

// FUSE (round 5, planner option stem_fuse_dz): dY -- the gradient of conv0's output behind BatchNorm, ReLU and the 2 x 2 max-pool -- is not
// read from a tensor but computed while the chunk is staged: a thread owns one 4-channel piece and, per chunk, two horizontally adjacent
// output pixels twice (the two columns of one pooling window of row oy): it requests the window's four x pieces and the pooled gradient's
// piece (10 requests of 16 bytes instead of 4), finds the window's first arg-max of relu(bn(x)) as the pool did, and writes
//     dz = A * (own the max and it is positive ? g : 0) + E - D * x        (the arithmetic of stem_bwd_kernel<1>, operation for operation)
// into the LDS chunk.  The 200 MB dz tensor is neither written nor read, stem_bwd_kernel<1> (the last kernel of the step on the caller's
// stream but one) is not launched; bit-identical weight gradient.
// EMU (round 5, planner option stem_wgrad_split with f32_split): the contraction on the bf16 matrix pipe as in wgrad3_kernel -- a lane takes
// its operands as 8 consecutive output pixels (dY: eight ds_read_b32 a pixel row apart; im2col: eight ds_read_b32 two input pixels apart --
// as many LDS reads as the fp32 k-steps they replace), cuts each into three bf16 pieces and issues six v_mfma_f32_32x32x16_bf16 per k-tile:
// 4 x CT x 6 x 32 = 2304 matrix-pipe cycles per chunk instead of 32 x CT x 64 = 6144 (CT = 3).
template <bool FUSE, bool EMU>
__global__ __launch_bounds__(WG3_THREADS, 1) void wgrad3_stem_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const WgradArgs& p = q.w;
    const int OH = p.H, OW = p.W, IH = p.IH, IW = p.IW;
    const int CP = stem_cp(IW), RP = stem_rp(IW);
    const int rows = q.rows_per_split;                     // output rows per workgroup
    const int NR = 2 * rows + 5;                           // staged input rows; slot NR is all zeros (k >= 147 lanes)
    const int DYOFF = (NR + 1) * RP;                       // [64][128] floats (16-byte aligned: rounded up below)
    const int dyoff = (DYOFF + 3) & ~3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int wpi = q.c0;                                  // workgroups per image
    const int img = blockIdx.x / wpi;
    const int r0 = (blockIdx.x - img * wpi) * rows;
    int r1 = r0 + rows;
    if (r1 > OH) r1 = OH;

    // ---- input rows 2 r0 - 3 ... (zeros outside the image and in the 3 + 3 border columns).  All global loads of the
    // prologue are issued before anything waits for one (a load-then-store loop pays the HBM latency per iteration).
    typedef float f32x4n __attribute__((ext_vector_type(4)));      // (a native vector: arrays of HIP's float4 struct end up in scratch here)
    constexpr int MAXIN = 16;                              // float4 per thread: (2*13+5) rows * 3 * 64 float4 / 512 threads = 11.6
    const int per_row = 3 * (IW >> 2);                     // float4 per input row
    const int total = NR * per_row;
    f32x4n iv[MAXIN];
    int idst[MAXIN];
#pragma unroll
    for (int u = 0; u < MAXIN; ++u) {
        const int i = tid + WG3_THREADS * u;
        idst[u] = -1;
        iv[u] = f32x4n{0.f, 0.f, 0.f, 0.f};
        if (i < total) {
            const int j = i / per_row;
            const int rem = i - j * per_row;
            const int c = rem / (IW >> 2);
            const int x4 = rem - c * (IW >> 2);
            const int ir = 2 * r0 - 3 + j;
            if (ir >= 0 && ir < IH) {
                iv[u] = *reinterpret_cast<const f32x4n*>(p.img + (((size_t)img * 3 + c) * IH + ir) * IW + 4 * x4);
                idst[u] = j * RP + c * CP + 3 + 4 * x4;
            }
        }
    }
    for (int i = tid; i < dyoff; i += WG3_THREADS) lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAXIN; ++u)
        if (idst[u] >= 0) {
            float* d = lds + idst[u];
            d[0] = iv[u][0]; d[1] = iv[u][1]; d[2] = iv[u][2]; d[3] = iv[u][3];
        }

    // ---- dY chunks: 64 pixels x 128 channels = 2048 float4, 4 per thread
    const int cpr = OW / STEM_CHUNK;                       // chunks per output row
    const int nchunks = (r1 - r0) * cpr;
    f32x4n dv[FUSE ? 10 : 4];
    // FUSE: this thread's channel piece is 4 * (tid & 31) in every item (512 = 16 x 32 threads); its BatchNorm tables in registers
    f32x4n tS = {0.f, 0.f, 0.f, 0.f}, tH = tS, tE = tS, tD = tS;
    int f_row = 0;                                         // parity of the chunk's output row inside its pooling window
    if (FUSE) {
        const double invM = 1.0 / p.scount;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (tid & 31) + e;
            const double mean = p.sstats[c] / p.scount;
            double var = p.sstats[128 + c] / p.scount - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const double scale = (double)p.gamma[c] * istd;
            const double c1 = p.sred[c] * invM, c2 = p.sred[128 + c] * invM;
            const double D = scale * c2 * istd;
            tS[e] = (float)scale;
            tH[e] = (float)((double)p.beta[c] - mean * scale);
            tD[e] = (float)D;
            tE[e] = (float)(D * mean - scale * c1);
        }
    }
    auto issue = [&](int ci) {
        const int oy = r0 + ci / cpr;
        const int x0 = STEM_CHUNK * (ci - (ci / cpr) * cpr);
        if (FUSE) {
            f_row = oy & 1;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {               // window jj: output columns x0 + 2 pw, + 1 with pw = (tid >> 5) + 16 jj
                const int pw = (tid >> 5) + 16 * jj;
                const size_t w00 = (((size_t)img * OH + (oy & ~1)) * OW + x0 + 2 * pw) * 128 + 4 * (tid & 31);
                dv[5 * jj + 0] = *reinterpret_cast<const f32x4n*>(p.sx + w00);
                dv[5 * jj + 1] = *reinterpret_cast<const f32x4n*>(p.sx + w00 + 128);
                dv[5 * jj + 2] = *reinterpret_cast<const f32x4n*>(p.sx + w00 + (size_t)OW * 128);
                dv[5 * jj + 3] = *reinterpret_cast<const f32x4n*>(p.sx + w00 + (size_t)OW * 128 + 128);
                dv[5 * jj + 4] = *reinterpret_cast<const f32x4n*>(p.sgy + (((size_t)img * (OH >> 1) + (oy >> 1)) * (OW >> 1) + (x0 >> 1) + pw) * 128 + 4 * (tid & 31));
            }
        } else {
            const float* src = p.dy + (((size_t)img * OH + oy) * OW + x0) * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j) dv[j] = *reinterpret_cast<const f32x4n*>(src + 4 * (tid + WG3_THREADS * j));
        }
    };
    auto commit = [&]() {
        if (FUSE) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int pw = (tid >> 5) + 16 * jj;
                f32x4n o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int am = 0;                            // first arg-max of the window, as the forward's pool and stem_bwd_kernel take it
                    float best = fmaxf(fmaf(dv[5 * jj + 0][e], tS[e], tH[e]), 0.f);
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const float a = fmaxf(fmaf(dv[5 * jj + k][e], tS[e], tH[e]), 0.f);
                        if (a > best) { best = a; am = k; }
                    }
                    const float g = best > 0.f ? dv[5 * jj + 4][e] : 0.f;
                    const float xa = f_row ? dv[5 * jj + 2][e] : dv[5 * jj + 0][e];      // this row's two pixels of the window
                    const float xb = f_row ? dv[5 * jj + 3][e] : dv[5 * jj + 1][e];
                    o0[e] = fmaf(-tD[e], xa, fmaf(tS[e], (am == 2 * f_row) ? g : 0.f, tE[e]));
                    o1[e] = fmaf(-tD[e], xb, fmaf(tS[e], (am == 2 * f_row + 1) ? g : 0.f, tE[e]));
                }
                *reinterpret_cast<f32x4n*>(lds + dyoff + (2 * pw) * 128 + 4 * (tid & 31)) = o0;
                *reinterpret_cast<f32x4n*>(lds + dyoff + (2 * pw + 1) * 128 + 4 * (tid & 31)) = o1;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4n*>(lds + dyoff + 4 * (tid + WG3_THREADS * j)) = dv[j];
        }
    };

    const int nt = wave & 3;
    const int half = wave >> 2;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // this lane's im2col column per k-tile: offset inside the staged rows, and whether it exists (k < 147)
    int koff[3], kmul[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = (half * 3 + t) * 32 + li;
        const bool valid = k < STEM_K && (half == 0 || t < 2);
        const int c = k / 49, rem = k - (k / 49) * 49;
        const int ky = rem / 7, kx = rem - (rem / 7) * 7;
        koff[t] = valid ? ky * RP + c * CP + kx : NR * RP;
        kmul[t] = valid ? 1 : 0;
    }

    issue(0);
    commit();
    __syncthreads();
    auto chunk_mma = [&](auto HF, int ci) {
        constexpr int CT = StemTiles<decltype(HF)::value>::N;
        const int oy = r0 + ci / cpr;
        const int x0 = STEM_CHUNK * (ci - (ci / cpr) * cpr);
        const int aoff = dyoff + hi * 128 + nt * 32 + li;
        int boff[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) boff[t] = koff[t] + kmul[t] * (2 * (oy - r0)) * RP + 2 * (x0 + hi);
        float a_cur = lds[aoff], b_cur[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) b_cur[t] = lds[boff[t]];
#pragma unroll 4
        for (int pp = 0; pp < STEM_CHUNK / 2; ++pp) {
            const int pn = pp + 1 < STEM_CHUNK / 2 ? pp + 1 : pp;
            const float a_nxt = lds[aoff + 2 * pn * 128];
            float b_nxt[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) b_nxt[t] = lds[boff[t] + 4 * pn];
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + CT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < CT; ++t) b_cur[t] = b_nxt[t];
        }
    };
    auto chunk_mma_emu = [&](auto HF, int ci) {
        constexpr int CT = StemTiles<decltype(HF)::value>::N;
        const int oy = r0 + ci / cpr;
        const int x0 = STEM_CHUNK * (ci - (ci / cpr) * cpr);
#pragma unroll
        for (int ks = 0; ks < STEM_CHUNK / 16; ++ks) {
            const int p0 = 16 * ks + 8 * hi;                    // this lane's 8 pixels of the 16-pixel k-step
            const float* A = lds + dyoff + p0 * 128 + nt * 32 + li;
            float af[8], bf[CT][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = A[j * 128];
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const float* B = lds + koff[t] + kmul[t] * (2 * (oy - r0)) * RP + 2 * (x0 + p0);
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[t][j] = B[2 * j];
            }
            u32x4 ah, am, al;
            split_bf16x3(af, ah, am, al);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                u32x4 bh, bm, bl;
                split_bf16x3(bf[t], bh, bm, bl);
                acc[t] = mfma_split6(ah, am, al, bh, bm, bl, acc[t]);
            }
        }
    };
    for (int ci = 0; ci < nchunks; ++ci) {
        const bool more = ci + 1 < nchunks;
        if (more) issue(ci + 1);
        if constexpr (EMU) {
            if (half == 0) chunk_mma_emu(std::integral_constant<int, 0>{}, ci);
            else chunk_mma_emu(std::integral_constant<int, 1>{}, ci);
        } else {
            if (half == 0) chunk_mma(std::integral_constant<int, 0>{}, ci);
            else chunk_mma(std::integral_constant<int, 1>{}, ci);
        }
        __syncthreads();
        if (more) commit();
        __syncthreads();
    }

    float* out = q.part + (size_t)blockIdx.x * 128 * STEM_K;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = (half * 3 + t) * 32 + li;
        if (k >= STEM_K || (half == 1 && t == 2)) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * STEM_K + k] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6 (planner option stem_wgrad_planes, acts with f32_split + stem_wgrad_split): the stem's weight gradient with BOTH operands cut once and
// the work of a workgroup split by ROLE.  wgrad3_stem_kernel<*, EMU> keeps the image rows and the dY chunk as fp32 in LDS and every wave cuts the
// fragments it reads (an image element ends up cut ~49 times: 12 (tap, pixel) pairs x 4 output-channel waves; ~256 VALU instructions per wave and
// 16-pixel k-step next to 18 MFMAs), and all eight waves walk the same phases in lock-step -- requests, cut, MFMAs, barrier -- so the matrix pipe
// of a SIMD waits while its two waves cut and the VALU waits while they multiply: 150-160 us, alone on the GPU at the end of every step, against
// 42 us of matrix pipe.  Here
//   * the image rows live in a RING of 16 row slots as three bf16 planes, de-interleaved by column parity (a stride-2 tap walks consecutive
//     elements of one parity line): line (c, kx & 1) holds element idx = ox + (kx >> 1) of the 3 + IW + 3 padded row at byte 2 idx.  A lane's
//     8 pixels of one plane are 5 dwords (ds_read2_b32 x 2 + ds_read_b32 at byte 2 (x0 + p0) + 4 (kx >> 2)) and four v_alignbyte_b32 by
//     2 ((kx >> 1) & 1) bytes; line pitch IW + 8 bytes (= 2 mod 4 dwords: the 32 im2col columns of a tile fall into different banks: SQ_LDS_BANK_CONFLICT
//     5 % of the LDS cycles); zero rows / borders are zeros in the planes;
//   * the dY chunk (32 pixels, two buffers, ONE barrier per chunk) is cut on its way in, behind the fused BatchNorm / ReLU / pool backward, into three
//     planes of wgrad4's row layout and read by ds_read_b64_tr_b16 (wg5_frag);
//   * waves 0-3 (one per SIMD) are CONSUMERS: wave w owns output-channel tile w and all five im2col tiles, and does nothing but fragment reads and
//     MFMAs -- the matrix pipe of its SIMD has one client that keeps it fed back to back; waves 4-7 are PRODUCERS: global loads, the dz arithmetic, the
//     cuts and the LDS stores of the NEXT chunk and of the image rows of the next output-row pair -- pure VALU / memory work beside the other wave's MFMAs;
//   * chunks are walked by pooling-window row PAIR (row 2m and 2m + 1 of one 32-column block back to back): conv0's output and the pooled gradient are
//     requested once per window instead of once per output row (490 -> ~270 MB per launch), two blocks ahead of their cut.
// Same pieces and the same six products per pair as wgrad3_stem_kernel<*, true>, the same rows per workgroup and partial-tile layout; the order in which
// a workgroup's pixels enter the fp32 accumulators differs (row pairs): agreement to fp32 summation order, not bit for bit (tests/test_gpu_exact.py).
constexpr int SPL_NS = 16;                                 // ring slots: 9 rows under contraction + 4 being staged, a power of two
constexpr int SPL_P = 32;                                  // output pixels per dY chunk
constexpr int SPL_APITCH = 320;                            // bytes per pixel row of a dY plane (128 bf16 + 64: an odd multiple of 64)
constexpr int SPL_APLANE = SPL_P * SPL_APITCH;
constexpr int SPL_ABUF = 3 * SPL_APLANE;
constexpr int SPL_THREADS = 512;                           // 8 waves, two per SIMD: a consumer and a producer
constexpr int SPL_CONS = 256;                              // consumer threads (waves 0-3)
constexpr int SPL_PROD = SPL_THREADS - SPL_CONS;           // producer threads (waves 4-7)
#ifndef SPL_CONS_PRIO
#define SPL_CONS_PRIO 2
#endif
typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

struct SplCursor { int m, lo, nr, xb, j; };                // output-row pair, its first row / row count inside the workgroup's range, column block, row index

template <bool FUSE>
__global__ __launch_bounds__(SPL_THREADS, 1) void wgrad3_stem_planes_kernel(const Wg3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    const WgradArgs& p = q.w;
    const int OH = p.H, OW = p.W, IH = p.IH, IW = p.IW;
    const int LP = IW + 8;                                 // bytes per (channel, parity) line of a plane
    const int PL = 6 * LP;                                 // bytes per plane of a row slot
    const int SLOT = 3 * PL;
    char* abuf0 = smem + (SPL_NS + 1) * SLOT;              // slot SPL_NS stays all zeros (im2col columns >= 147)
    const int rows = q.rows_per_split;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int wpi = q.c0;
    const int img = blockIdx.x / wpi;
    const int r0 = (blockIdx.x - img * wpi) * rows;
    int r1 = r0 + rows;
    if (r1 > OH) r1 = OH;
    const int cpr = OW / SPL_P;
    const int nchunks = r1 > r0 ? (r1 - r0) * cpr : 0;
    const bool producer = wave >= SPL_CONS / 64;

    for (int i = tid * 16; i < (SPL_NS + 1) * SLOT; i += SPL_THREADS * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0u, 0u, 0u, 0u);

    // ---- the chunk sequence: pairs m = r0 >> 1 ..., of a pair the rows inside [r0, r1), column blocks of 32 pixels; per block the rows back to back
    auto cur_init = [&](SplCursor& c) {
        c.m = r0 >> 1; c.lo = r0; c.nr = (r1 < 2 * c.m + 2 ? r1 : 2 * c.m + 2) - c.lo; c.xb = 0; c.j = 0;
    };
    auto cur_next = [&](SplCursor& c) -> bool {            // true: the next chunk opens a new block
        if (++c.j < c.nr) return false;
        c.j = 0;
        if (++c.xb == cpr) { c.xb = 0; ++c.m; c.lo = 2 * c.m; c.nr = (r1 < 2 * c.m + 2 ? r1 : 2 * c.m + 2) - c.lo; }
        return true;
    };

    // ---- image rows: item i of a row PAIR is one float4 (2 rows x 3 channels x IW / 4 items); pixel ix = 4 sx4 + e sits at element ix + 3 of the
    // padded row: parity (ix + 3) & 1, idx (ix + 3) >> 1
    const int per_row = 3 * (IW >> 2);
    struct Item { bool ok; int sj, so0, so1; unsigned g; };
    auto make_item = [&](int i) {
        Item it;
        it.ok = i < 2 * per_row;
        const int ii = it.ok ? i : 0;
        it.sj = ii / per_row;
        const int sc = (ii - it.sj * per_row) / (IW >> 2);
        const int sx4 = (ii - it.sj * per_row) - sc * (IW >> 2);
        it.so1 = (2 * sc + 1) * LP + 2 * (2 * sx4 + 1);     // e = 0, 2: parity 1, idx 2 sx4 + 1, + 2 (two 2-byte stores)
        it.so0 = (2 * sc) * LP + 2 * (2 * sx4 + 2);         // e = 1, 3: parity 0, idx 2 sx4 + 2, + 3 (one 4-byte store)
        it.g = 4u * (unsigned)((sc * IH + it.sj) * IW + 4 * sx4);      // bytes from the image's row `first` (uniform base + 32-bit lane offset)
        return it;
    };
    const char* img_base = reinterpret_cast<const char*>(p.img + (size_t)img * 3 * IH * IW);
    auto img_issue = [&](const Item& it, f32x4n& v, int& row, int first) {
        if (it.ok) {
            const int iy = first + it.sj;
            row = iy;
            v = f32x4n{0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < IH) v = *reinterpret_cast<const f32x4n*>(img_base + (ptrdiff_t)first * IW * 4 + it.g);
        }
    };
    auto img_commit = [&](const Item& it, const f32x4n& v, int row) {
        if (it.ok) {
            char* d = smem + ((row + 4) & (SPL_NS - 1)) * SLOT;
            unsigned h0, m0, l0, h1, m1, l1;
            split_bf16x3_pair(f32x2_op{v[1], v[3]}, h0, m0, l0);
            split_bf16x3_pair(f32x2_op{v[0], v[2]}, h1, m1, l1);
            *reinterpret_cast<unsigned*>(d + it.so0) = h0;
            *reinterpret_cast<unsigned*>(d + PL + it.so0) = m0;
            *reinterpret_cast<unsigned*>(d + 2 * PL + it.so0) = l0;
            *reinterpret_cast<unsigned short*>(d + it.so1) = (unsigned short)h1;
            *reinterpret_cast<unsigned short*>(d + it.so1 + 2) = (unsigned short)(h1 >> 16);
            *reinterpret_cast<unsigned short*>(d + PL + it.so1) = (unsigned short)m1;
            *reinterpret_cast<unsigned short*>(d + PL + it.so1 + 2) = (unsigned short)(m1 >> 16);
            *reinterpret_cast<unsigned short*>(d + 2 * PL + it.so1) = (unsigned short)l1;
            *reinterpret_cast<unsigned short*>(d + 2 * PL + it.so1 + 2) = (unsigned short)(l1 >> 16);
        }
    };
    // prologue, every thread: rows 4 m0 - 3 ... 4 m0 + 6 (five pairs: what the first output-row pair reads), all requested before the first is cut
    if (nchunks > 0) {
        const Item it = make_item(tid);
        f32x4n iv[5];
        int ir[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) img_issue(it, iv[u], ir[u], 4 * (r0 >> 1) - 3 + 2 * u);
        __syncthreads();                                   // the zeros are in place
#pragma unroll
        for (int u = 0; u < 5; ++u) img_commit(it, iv[u], ir[u]);
    } else {
        __syncthreads();
    }

    if (producer) {
        // ================= producers: loads, dz arithmetic, cuts, LDS stores -- one chunk ahead of the consumers =================
        const int tp = tid - SPL_CONS;
        const int pw = tp >> 5, pc = 4 * (tp & 31);        // FUSE: pooling windows pw, pw + 8 of a block; else pixels pw + 8 i of a row
        f32x4n tS = {0.f, 0.f, 0.f, 0.f}, tH = tS, tE = tS, tD = tS;
        if (FUSE) {                                        // tables of this thread's channel piece (wgrad3_stem_kernel<true>, operation for operation)
            const double invM = 1.0 / p.scount;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = pc + e;
                const double mean = p.sstats[c] / p.scount;
                double var = p.sstats[128 + c] / p.scount - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const double istd = 1.0 / sqrt(var + (double)BN_EPS);
                const double scale = (double)p.gamma[c] * istd;
                const double c1 = p.sred[c] * invM, c2 = p.sred[128 + c] * invM;
                const double D = scale * c2 * istd;
                tS[e] = (float)scale;
                tH[e] = (float)((double)p.beta[c] - mean * scale);
                tD[e] = (float)D;
                tE[e] = (float)(D * mean - scale * c1);
            }
        }
        constexpr int NDV = FUSE ? 5 : 4;
        f32x4n ds[2][2][NDV];                              // [set][window | row of the block][FUSE: 4 x pieces + pooled gradient | pixel pw + 8 i]: two blocks in flight
        f32x4n tc[2][2];                                   // FUSE: [window][pixel] S * (routed gradient) + E of the window's SECOND row, kept from the first row's pass
        // lane offsets (bytes, constant over the kernel) from the uniform base of a block: FUSE: window pw + 8 w of conv0's output row 2 m and of the pooled
        // gradient's row m; else pixel pw of a dY row
        unsigned offx[2], offg[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            offx[w] = 4u * (unsigned)(2 * (pw + 8 * w) * 128 + pc);
            offg[w] = 4u * (unsigned)((pw + 8 * w) * 128 + pc);
        }
        auto load_block = [&](auto SET, int m, int xb) {
            constexpr int S = decltype(SET)::value;
            const int x0 = SPL_P * xb;
#ifdef CUNET_SPL_NO_LOADS      // probe builds (timing only, wrong results)
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int u = 0; u < NDV; ++u) ds[S][w][u] = f32x4n{0.25f * (float)(m + u), 1.f, -0.5f, 0.125f * (float)x0};
            return;
#endif
            if (FUSE) {
                const char* r0p = reinterpret_cast<const char*>(p.sx + (((size_t)img * OH + 2 * m) * OW + x0) * 128);
                const char* r1p = r0p + (size_t)OW * 512;
                const char* gp = reinterpret_cast<const char*>(p.sgy + (((size_t)img * (OH >> 1) + m) * (OW >> 1) + (x0 >> 1)) * 128);
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    ds[S][w][0] = *reinterpret_cast<const f32x4n*>(r0p + offx[w]);
                    ds[S][w][1] = *reinterpret_cast<const f32x4n*>(r0p + offx[w] + 512);
                    ds[S][w][2] = *reinterpret_cast<const f32x4n*>(r1p + offx[w]);
                    ds[S][w][3] = *reinterpret_cast<const f32x4n*>(r1p + offx[w] + 512);
                    ds[S][w][NDV - 1] = *reinterpret_cast<const f32x4n*>(gp + offg[w]);
                }
            } else {
                const int lo = 2 * m < r0 ? r0 : 2 * m;
                const int hiq = 2 * m + 2 > r1 ? r1 : 2 * m + 2;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int oy = lo + w < hiq ? lo + w : lo;     // (a block of one row: the same row twice)
                    const char* src = reinterpret_cast<const char*>(p.dy + (((size_t)img * OH + oy) * OW + x0) * 128);
#pragma unroll
                    for (int u = 0; u < NDV; ++u) ds[S][w][u] = *reinterpret_cast<const f32x4n*>(src + offg[0] + 4096u * u);
                }
            }
        };
        auto put = [&](char* buf, int pixel, const f32x4n& o) {
            u32x2w h, m, l;
            wg5_cut4(make_float4(o[0], o[1], o[2], o[3]), h, m, l);
            char* d = buf + pixel * SPL_APITCH + 2 * pc;
            *reinterpret_cast<u32x2w*>(d) = h;
            *reinterpret_cast<u32x2w*>(d + SPL_APLANE) = m;
            *reinterpret_cast<u32x2w*>(d + 2 * SPL_APLANE) = l;
        };
        // the FIRST row of a block inside the workgroup's range (row parity F inside its pooling windows; F = 1: a range that starts on an odd row).
        // dz = A * (own the max and it is positive ? g : 0) + E - D * x, the arithmetic of stem_bwd_kernel<1> operation for operation; F = 0 also leaves
        // S * (routed gradient) + E of the windows' second row for commit_second.
        auto commit_first = [&](auto SET, auto FR, char* buf) {
            constexpr int S = decltype(SET)::value;
            constexpr int F = decltype(FR)::value;
#ifdef CUNET_SPL_NO_COMMIT     // probe builds (timing only, wrong results)
            return;
#endif
            if (FUSE) {
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    f32x4n o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int am = 0;                        // first arg-max of the window, as the forward's pool and stem_bwd_kernel take it
                        float best = fmaxf(fmaf(ds[S][w][0][e], tS[e], tH[e]), 0.f);
#pragma unroll
                        for (int k = 1; k < 4; ++k) {
                            const float a = fmaxf(fmaf(ds[S][w][k][e], tS[e], tH[e]), 0.f);
                            if (a > best) { best = a; am = k; }
                        }
                        const float g = best > 0.f ? ds[S][w][NDV - 1][e] : 0.f;
                        o0[e] = fmaf(-tD[e], ds[S][w][2 * F][e], fmaf(tS[e], (am == 2 * F) ? g : 0.f, tE[e]));
                        o1[e] = fmaf(-tD[e], ds[S][w][2 * F + 1][e], fmaf(tS[e], (am == 2 * F + 1) ? g : 0.f, tE[e]));
                        if (F == 0) {
                            tc[w][0][e] = fmaf(tS[e], (am == 2) ? g : 0.f, tE[e]);
                            tc[w][1][e] = fmaf(tS[e], (am == 3) ? g : 0.f, tE[e]);
                        }
                    }
                    put(buf, 2 * (pw + 8 * w), o0);
                    put(buf, 2 * (pw + 8 * w) + 1, o1);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NDV; ++u) put(buf, pw + 8 * u, ds[S][0][u]);
            }
        };
        auto commit_second = [&](auto SET, char* buf) {     // the second row of a block of two
            constexpr int S = decltype(SET)::value;
#ifdef CUNET_SPL_NO_COMMIT     // probe builds (timing only, wrong results)
            return;
#endif
            if (FUSE) {
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    f32x4n o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = fmaf(-tD[e], ds[S][w][2][e], tc[w][0][e]);
                        o1[e] = fmaf(-tD[e], ds[S][w][3][e], tc[w][1][e]);
                    }
                    put(buf, 2 * (pw + 8 * w), o0);
                    put(buf, 2 * (pw + 8 * w) + 1, o1);
                }
            } else {
#pragma unroll
                for (int u = 0; u < NDV; ++u) put(buf, pw + 8 * u, ds[S][1][u]);
            }
        };
        // the image rows of the NEXT output-row pair: rows 4 m + 7 ... 4 m + 10 beyond what the prologue / the last pair staged (items tp, tp + 256 of
        // two row pairs), requested at a pair's first step and cut at its last
        const Item itA = make_item(tp), itB = make_item(tp + SPL_PROD);
        f32x4n nv[4] = {};
        int nrow[4] = {0, 0, 0, 0};
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;

        if (nchunks > 0) {
            int m = r0 >> 1, xb = 0;                       // the block whose chunks the consumers walk
            int mL = m, xbL = 0;                           // block cursor of the loads
            auto blk_next = [&](int& mm, int& xx) { if (++xx == cpr) { xx = 0; ++mm; } };
            load_block(S0{}, mL, xbL);
            blk_next(mL, xbL);
            if (2 * mL < r1) load_block(S1{}, mL, xbL);
            blk_next(mL, xbL);                             // (mL, xbL) = block 2: the next one to request
            if (r0 & 1) commit_first(S0{}, S1{}, abuf0);
            else commit_first(S0{}, S0{}, abuf0);
            __syncthreads();
            int wb = 1;                                    // dY buffer the next cut goes to
            // one block: its first chunk is being contracted when the body starts; SET holds its loads, SET ^ 1 the next block's
            auto block = [&](auto SET) {
                using SN = std::integral_constant<int, decltype(SET)::value ^ 1>;
                const int lo = 2 * m < r0 ? r0 : 2 * m;
                const int nr = (2 * m + 2 > r1 ? r1 : 2 * m + 2) - lo;
                const bool stage = 2 * (m + 1) < r1;       // another pair follows: its four new rows 4 m + 7 ... 4 m + 10
                if (stage && xb == 0) {
                    img_issue(itA, nv[0], nrow[0], 4 * m + 7);
                    img_issue(itB, nv[1], nrow[1], 4 * m + 7);
                    img_issue(itA, nv[2], nrow[2], 4 * m + 9);
                    img_issue(itB, nv[3], nrow[3], 4 * m + 9);
                }
                if (nr == 2) {
                    commit_second(SET, abuf0 + wb * SPL_ABUF);
                    wb ^= 1;
                    __syncthreads();                       // the block's second chunk is being contracted now
                }
                int mn = m, xn = xb;
                blk_next(mn, xn);
                const bool have_next = 2 * mn < r1;
                if (2 * mL < r1) load_block(SET, mL, xbL);     // SET has served its block: the block after the next one
                blk_next(mL, xbL);
                if (have_next) {
                    if ((2 * mn < r0 ? r0 : 2 * mn) & 1) commit_first(SN{}, S1{}, abuf0 + wb * SPL_ABUF);
                    else commit_first(SN{}, S0{}, abuf0 + wb * SPL_ABUF);
                    wb ^= 1;
                }
                if (stage && xb == cpr - 1) {
                    img_commit(itA, nv[0], nrow[0]);
                    img_commit(itB, nv[1], nrow[1]);
                    img_commit(itA, nv[2], nrow[2]);
                    img_commit(itB, nv[3], nrow[3]);
                }
                m = mn; xb = xn;
                __syncthreads();
                return have_next;
            };
            for (;;) {
                if (!block(S0{})) break;
                if (!block(S1{})) break;
            }
        }
        return;
    }

    // ================= consumers: wave w owns output-channel tile w and the five im2col tiles =================
    // (the consumer is the pole of its SIMD: its instructions go first, the producer's VALU work takes the slots it leaves)
    __builtin_amdgcn_s_setprio(SPL_CONS_PRIO);
    const int nt = wave;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // this lane's im2col column per k-tile: its line inside a plane (+ the dword the five-dword window starts at), its row offset ky and the byte
    // shift of the window; columns >= 147 read the zero slot
    int kline[5], kky[5], ksh[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = t * 32 + li;
        const bool valid = k < STEM_K;
        const int c = k / 49, rem = k - (k / 49) * 49;
        const int ky = rem / 7, kx = rem - (rem / 7) * 7;
        kline[t] = valid ? (2 * c + (kx & 1)) * LP + 4 * (kx >> 2) : 0;
        kky[t] = valid ? ky : -1;
        ksh[t] = valid ? 2 * ((kx >> 1) & 1) : 0;
    }
    // transpose-read address of this lane inside a dY plane (wgrad5_split_kernel): pixel 8 hi + ((l & 15) >> 2) (+ 4 for the second read),
    // channel piece 16 ((l >> 4) & 1) + 4 (l & 3) of output-channel tile nt
    const int off_a = (8 * hi + ((lane & 15) >> 2)) * SPL_APITCH + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) + nt * 64;
    if (nchunks > 0) {
        SplCursor cc;
        cur_init(cc);
        __syncthreads();                                   // chunk 0 is cut
        for (int ci = 0; ci < nchunks; ++ci) {
            const int oy = cc.lo + cc.j;
            const int x0 = SPL_P * cc.xb;
            const char* cur = abuf0 + (ci & 1) * SPL_ABUF;
#ifndef CUNET_SPL_NO_MMA       // probe builds (timing only, wrong results)
            const int rb = 2 * oy + 1;                     // input row 2 oy - 3 + ky lives in slot (2 oy + 1 + ky) & 15
            const char* bp[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int sl = kky[t] >= 0 ? ((rb + kky[t]) & (SPL_NS - 1)) : SPL_NS;
                bp[t] = smem + sl * SLOT + kline[t] + (kky[t] >= 0 ? 2 * (x0 + 8 * hi) : 0);
            }
            // ten stages (k-step, tile) per chunk, software-pipelined by hand: the dwords of stage i + 1 are requested BEFORE stage i's v_alignbyte /
            // MFMAs (left alone hipcc requests a tile's dwords right in front of its own MFMAs: an LDS round trip exposed per tile)
            unsigned raw[2][3][5];
            u32x4 af[2][3];
            auto rd = [&](int ks, int t, unsigned (&r)[3][5]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const char* s = bp[t] + pl * PL + 32 * ks;
                    const u32x2_a4 d01 = *reinterpret_cast<const u32x2_a4*>(s);
                    const u32x2_a4 d23 = *reinterpret_cast<const u32x2_a4*>(s + 8);
                    r[pl][0] = d01.x; r[pl][1] = d01.y; r[pl][2] = d23.x; r[pl][3] = d23.y;
                    r[pl][4] = *reinterpret_cast<const unsigned*>(s + 16);
                }
            };
            auto rda = [&](int ks, u32x4 (&a)[3]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = wg5_frag(cur + pl * SPL_APLANE + off_a + ks * 16 * SPL_APITCH, 4 * SPL_APITCH);
            };
            rda(0, af[0]);
            rd(0, 0, raw[0]);
#pragma unroll
            for (int st = 0; st < 10; ++st) {
                const int ks = st / 5, t = st - 5 * (st / 5);
                if (st + 1 < 10) rd((st + 1) / 5, (st + 1) - 5 * ((st + 1) / 5), raw[(st + 1) & 1]);
                if (st == 3) rda(1, af[1]);
                __builtin_amdgcn_sched_barrier(0);
                u32x4 b[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned (&r)[5] = raw[st & 1][pl];
                    b[pl] = u32x4{__builtin_amdgcn_alignbyte(r[1], r[0], ksh[t]), __builtin_amdgcn_alignbyte(r[2], r[1], ksh[t]),
                                  __builtin_amdgcn_alignbyte(r[3], r[2], ksh[t]), __builtin_amdgcn_alignbyte(r[4], r[3], ksh[t])};
                }
                acc[t] = mfma_split6(af[ks][0], af[ks][1], af[ks][2], b[0], b[1], b[2], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            (void)cur_next(cc);
            __syncthreads();
        }
    }

    float* out = q.part + (size_t)blockIdx.x * 128 * STEM_K;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = t * 32 + li;
        if (k >= STEM_K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)n * STEM_K + k] = acc[t][r];
        }
    }
}
size_t wgrad3_stem_planes_lds_bytes(int IW) { return (size_t)(SPL_NS + 1) * 18 * (IW + 8) + 2 * SPL_ABUF; }
bool wgrad3_stem_planes_supported(const WgradArgs& a) {
    if (a.img == nullptr || a.Cout != 128 || a.lddy != 128 || a.Ccat != STEM_K) return false;
    if (a.W % SPL_P || a.IW % 8 || a.IW != 2 * a.W || a.IH != 2 * a.H || 6 * (a.IW / 4) > SPL_PROD * 2) return false;
    return wgrad3_stem_planes_lds_bytes(a.IW) <= 160 * 1024;
}

// rows: output rows per workgroup; wpi: workgroups per image (wpi * rows >= OH); part: [N * wpi][128][147]
size_t wgrad3_stem_lds_bytes(int IW, int rows) {
    return (size_t)((((2 * rows + 6) * stem_rp(IW) + 3) & ~3) + STEM_CHUNK * 128) * 4;
}
bool wgrad3_stem_supported(const WgradArgs& a, int rows) {
    if (a.img == nullptr || a.Cout != 128 || a.lddy != 128 || a.Ccat != STEM_K) return false;
    if (a.W % STEM_CHUNK || a.IW % 8 || a.IW != 2 * a.W || a.IH != 2 * a.H || rows < 1) return false;
    if ((long)(2 * rows + 5) * 3 * (a.IW / 4) > 16 * WG3_THREADS) return false;      // prologue registers (MAXIN)
    return wgrad3_stem_lds_bytes(a.IW, rows) <= 160 * 1024;
}
hipError_t launch_wgrad3_stem(const WgradArgs& a, float* part, int wpi, int rows, hipStream_t s) {
    if (!wgrad3_stem_supported(a, rows) || wpi < 1 || (long)wpi * rows < a.H) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[4] = {(const void*)&wgrad3_stem_kernel<false, false>, (const void*)&wgrad3_stem_kernel<true, false>,
                              (const void*)&wgrad3_stem_kernel<false, true>, (const void*)&wgrad3_stem_kernel<true, true>};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    Wg3Args q{};
    q.w = a;
    q.part = part;
    q.rows_per_split = rows;
    q.c0 = wpi;
    const int N = a.M / (a.H * a.W);
    if (a.sx != nullptr && (a.sgy == nullptr || a.sstats == nullptr || a.sred == nullptr || a.gamma == nullptr || a.beta == nullptr || (a.H & 1))) return hipErrorInvalidValue;
    if (a.split && a.split_planes && wgrad3_stem_planes_supported(a)) {      // both operands cut once (planner option stem_wgrad_planes)
        static bool planes_attr_done = false;
        if (!planes_attr_done) {
            const void* pf[2] = {(const void*)&wgrad3_stem_planes_kernel<false>, (const void*)&wgrad3_stem_planes_kernel<true>};
            for (const void* f : pf) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
            planes_attr_done = true;
        }
        if (a.sx != nullptr) hipLaunchKernelGGL((wgrad3_stem_planes_kernel<true>), dim3(N * wpi), dim3(SPL_THREADS), wgrad3_stem_planes_lds_bytes(a.IW), s, q);
        else hipLaunchKernelGGL((wgrad3_stem_planes_kernel<false>), dim3(N * wpi), dim3(SPL_THREADS), wgrad3_stem_planes_lds_bytes(a.IW), s, q);
        return hipGetLastError();
    }
    if (a.sx != nullptr) {
        if (a.split) hipLaunchKernelGGL((wgrad3_stem_kernel<true, true>), dim3(N * wpi), dim3(WG3_THREADS), wgrad3_stem_lds_bytes(a.IW, rows), s, q);
        else hipLaunchKernelGGL((wgrad3_stem_kernel<true, false>), dim3(N * wpi), dim3(WG3_THREADS), wgrad3_stem_lds_bytes(a.IW, rows), s, q);
    } else {
        if (a.split) hipLaunchKernelGGL((wgrad3_stem_kernel<false, true>), dim3(N * wpi), dim3(WG3_THREADS), wgrad3_stem_lds_bytes(a.IW, rows), s, q);
        else hipLaunchKernelGGL((wgrad3_stem_kernel<false, false>), dim3(N * wpi), dim3(WG3_THREADS), wgrad3_stem_lds_bytes(a.IW, rows), s, q);
    }
    return hipGetLastError();
}

bool wgrad3_3x3_supported(const WgradArgs& a) {
    if (a.taps != 9 || a.Cout != WG3C_N || a.lddy != WG3C_N || a.Ccat != WG3C_C || a.nseg != 1) return false;
    if (a.seg[0].ups || a.seg[0].C != WG3C_C || a.seg[0].ld % 4) return false;
    if (a.W < 2 || a.W > 64 || (a.W & 1) || a.M % a.W) return false;
    return true;
}

// bf16 x and bf16 dY at W = 16 / 32 / 64: the contraction runs on bf16 MFMA (wgrad3_3x3_bf16_kernel)
bool wgrad3_3x3_on_bf16_mfma(const WgradArgs& a) {
    return wgrad3_3x3_supported(a) && a.xbf16 == 2 && (a.W == 16 || a.W == 32 || a.W == 64) && a.seg[0].ld % 8 == 0;
}

// part: [S][9][32][128] floats; rows_per_split counts image rows (N*H of them in total)
hipError_t launch_wgrad3_3x3(const WgradArgs& a, float* part, int S, int rows_per_split, hipStream_t s) {
    if (!wgrad3_3x3_supported(a) || S < 1 || rows_per_split < 1) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)&wgrad3_3x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)&wgrad3_3x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)&wgrad3_3x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)&wgrad3_3x3_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    Wg3Args q{};
    q.w = a;
    q.part = part;
    q.rows_per_split = rows_per_split;
    if (wgrad3_3x3_on_bf16_mfma(a) && rows_per_split % 2 == 0 && (a.M / a.W) % 2 == 0) {
        static bool attr16_done = false;
        if (!attr16_done) {
            hipError_t e = hipFuncSetAttribute((const void*)&wgrad3_3x3_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)&wgrad3_3x3_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)&wgrad3_3x3_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr16_done = true;
        }
        const size_t ldpx = a.W + 8;
        const size_t smem16 = 1024 + (5 * WG3C_C * ldpx + 2 * 3 * WG3C_N * ldpx) * 2;
        if (a.W == 64) hipLaunchKernelGGL(wgrad3_3x3_bf16_kernel<4>, dim3(S), dim3(WG3_THREADS), smem16, s, q);
        else if (a.W == 32) hipLaunchKernelGGL(wgrad3_3x3_bf16_kernel<3>, dim3(S), dim3(WG3_THREADS), smem16, s, q);
        else hipLaunchKernelGGL(wgrad3_3x3_bf16_kernel<2>, dim3(S), dim3(WG3_THREADS), smem16, s, q);
        return hipGetLastError();
    }
    const size_t smem = ((size_t)2 * WG3C_C + (size_t)4 * (a.W + 2) * WG3C_C + (size_t)2 * a.W * WG3C_N) * 4;
    if (a.xbf16 == 2) hipLaunchKernelGGL(wgrad3_3x3_kernel<2>, dim3(S), dim3(WG3_THREADS), smem, s, q);
    else if (a.xbf16) hipLaunchKernelGGL(wgrad3_3x3_kernel<1>, dim3(S), dim3(WG3_THREADS), smem, s, q);
    else if (a.split && a.W % 16 == 0) hipLaunchKernelGGL((wgrad3_3x3_kernel<0, true>), dim3(S), dim3(WG3_THREADS), smem, s, q);
    else hipLaunchKernelGGL(wgrad3_3x3_kernel<0>, dim3(S), dim3(WG3_THREADS), smem, s, q);
    return hipGetLastError();
}

}  // namespace cunet
