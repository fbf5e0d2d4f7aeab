// Fused [concat -> BatchNorm -> ReLU] -> convolution as an MFMA GEMM, forward and backward-data.
//
// Replaces, for the reference's hot path:
//   models/cu_net.py:11-17   cat -> norm -> relu -> conv  (1x1 bottlenecks, adapters, heads)
//   models/cu_net.py:45-48,62 norm2 -> relu2 -> conv2     (3x3, pad 1)
//   models/cu_net.py:300     conv0 7x7/2                  (stem, im2col gather)
//   and the data-gradient half of autograd for those nodes (EP_BWD).
//
// Design (gfx950, fp32): WEIGHT-STATIONARY and barrier-free.  The weights of a node are tiny
// (<= 160 KB fp32) next to its activations (tens of MB), so a block first copies its whole B operand
// (pre-packed [tap][k/4][n][4], all taps, all K, NT*32 output channels) into the CU's 160 KB LDS
// ONCE; after that single barrier its (up to 12) waves never synchronise again: each wave walks its
// own 32-row tiles, reads the A operand (activations) straight from HBM/L2 as one 16-byte NHWC piece
// per lane -- lane (row = l&31, half = l>>5) takes channels 8q+4*half..+3 of the current 32-channel
// chunk, one chunk ahead in registers -- applies BatchNorm+ReLU in registers (a concat is never
// materialised), fetches B fragments with conflict-free ds_read_b128 and contracts with
// v_mfma_f32_32x32x2_f32.  Independent waves de-synchronise, so one wave's load latency is covered
// by the other waves' MFMAs (the first, double-buffered-B version of this kernel had a barrier per
// 32-channel chunk and reached only 37 % MFMA utilisation: its phases serialised).
// Per-channel batch statistics of the OUTPUT (sum, sum of squares) are produced in the
// epilogue in fp64 so every consumer BatchNorm reuses them (a BN over a concat is per channel).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_common.h"
#include "plan.h"

namespace cunet {

#ifdef CUNET_TUNING
// Phase clocks of the 1x1 data gradient's tile loop (tuning builds, CUNET_CONV_DBG & 512): shader cycles (s_memtime) summed over every
// wave and tile -- [0] tiles, [1] requests + MFMA issue of a tile, [2] x pieces into the LDS tile (waits for the x loads),
// [3] column pass (waits for the MFMA chain), [4] LDS fp64 atomics of the reductions, [5] dz pieces out (LDS read + global stores),
// [6] block set-up (operand copy, tables) x waves, [7] whole kernel x waves.  tools/dgrad_phase_clocks.py reads them.
__device__ unsigned long long g_conv_phase[8];
#define CUNET_STAMP(var) const unsigned long long var = CUNET_DBG(p, 512) ? __builtin_amdgcn_s_memtime() : 0ull
#else
#define CUNET_STAMP(var) const unsigned long long var = 0ull; (void)var
#endif
constexpr int CONV_MAX_WAVES = 12;       // 3 waves per SIMD (VGPR budget 168)
#ifndef CUNET_TEPI_WAVES
#define CUNET_TEPI_WAVES 12              // the fp32 1x1 / 3x3 data gradient with the LDS-tile epilogue (118 VGPRs: 16 would fit; probe builds)
#endif
template <int LD, int EP, int NT, bool FAST, int XBG>
constexpr int conv_max_waves() {
    // the fp32 data gradient with the LDS-tile epilogue: 3 or 4 channel tiles per wave hold 48 / 64 accumulators next to the A chunks in
    // flight and the epilogue's pieces -- two waves per SIMD (256 VGPRs) instead of three
    // XBG = 4 (the data gradient with the node's weight gradient fused in, round 4): 64 more accumulators per wave -- two waves per SIMD
    return (EP == EP_BWD && FAST && XBG == 4) ? 8 : (EP == EP_BWD && FAST && (XBG == 0 || XBG == 5 || XBG == 6)) ? (NT <= 2 ? CUNET_TEPI_WAVES : 8) : CONV_MAX_WAVES;
}

// (the body is a device function of (arguments, block coordinates): conv_kernel runs it on one problem, conv_pair_kernel on the
// problem blockIdx.z selects out of two)
template <int LD, int EP, int NT, bool FAST, int XBG>           // XBG, EP_BWD only: 1 = x of the concat is bf16, 2 = x, dY and dz are bf16
__device__ __forceinline__ void conv_body(const ConvArgs& p, const int bidx, const int bidy, const int gdimx) {
    constexpr int XB = (XBG == 1 || XBG == 2);       // storage of x
    // XBG = 3 (EP_FWD only): the heat-map heads' fused MSE epilogue.  Its 16 target values per channel tile were live next to the
    // accumulators in EVERY forward instantiation and cost the 128-column one 20 spilled registers, reloaded from scratch one by
    // one in its store epilogue (rocm ISA, round 3); only the heads' kernels carry it now.
    constexpr bool MSE = (EP == EP_FWD && (XBG == 3 || XBG == 7));
    // XBG = 6 (XBG = 7: with the fused MSE epilogue), round 4: the fp32 contraction on the bf16 matrix pipe.  Every fp32 operand value is
    // cut into three bf16 pieces a = h + m + l (8 + 8 + 8 significand bits: exact), and a b ~ hh' + hm' + mh' + mm' + hl' + lh' -- six
    // v_mfma_f32_32x32x16_bf16 (6 x 32 cycles) where the fp32 pipe needs eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles) for the same
    // 16 k.  The dropped terms (ml', lm', ll') are below 2^-24 of |a b|; measured against fp64 the result is as close as the fp32
    // MFMA's (tools/probes/split_bf16_probe.hip, profiles/r04_split_bf16_probe.txt).  B is cut once per block while it is copied into
    // LDS (three planes: 1.5 x the bytes), A in registers behind the BatchNorm + ReLU; loaders and epilogues are unchanged.
    constexpr bool EMU = (XBG == 6 || XBG == 7);
    constexpr int GB = (XBG == 2 && (LD == LD_PLAIN || LD == LD_PLAIN3)) ? 1 : 0;      // storage of the gradient tensors
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = NT * 32;                      // output channels per block
    const int kq4 = p.Kpad >> 2;
    const int brows = p.taps * kq4;                  // float4 rows of B
    float4* Bs = reinterpret_cast<float4*>(smem);    // [taps * Kpad/4][NB]   (EMU: [chunk][step 0..1][plane h, m, l][NT][64 lanes] x 16 bytes)
    const u32x4* Bp = reinterpret_cast<const u32x4*>(smem);
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(Bs + (size_t)brows * NB * (EMU ? 3 : 2) / 2);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat);   // [NB][2]
    // fp32 data gradient, nothing ragged: the epilogue's x loads and dz stores go through a wave-private LDS tile, one 32-column
    // tile of the slice after the other (see below).  The other instantiations keep the element-wise epilogue.
    constexpr bool TEPI = (EP == EP_BWD && FAST && (XBG == 0 || XBG == 4 || XBG == 5 || XBG == 6));
    // XBG = 5: the 1x1 data gradient over K = 128 with TWO 32-channel chunks of dY in flight per wave (see the PF2 tile loop below)
    constexpr bool PF2 = (TEPI && XBG == 5 && NT == 1 && LD == LD_PLAIN);
    static_assert(XBG != 5 || PF2, "the two-chunks-ahead loop exists for the fast fp32 1x1 data gradient, one channel tile per wave");
    // XBG = 4: the 1x1 data gradient that also computes the node's WEIGHT gradient from the dY and x tiles it has in hand (see the fused
    // tile loop below): one pass over dY and x per node instead of two kernels on two streams
    constexpr bool FUSEW = (TEPI && XBG == 4 && NT == 1 && LD == LD_PLAIN);
    static_assert(XBG != 4 || FUSEW, "the fused weight gradient exists for the fast fp32 1x1 data gradient, one channel tile per wave");
    constexpr int TEPI_PITCH = 36;                             // floats per tile row (16-byte aligned rows)
    float* tileT = reinterpret_cast<float*>(redbuf + NB * 2);  // TEPI: [waves][32][36] (FUSEW: a second tile per wave behind them)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    CUNET_STAMP(tk0);
    // block -> (row block bx of gxd, column slice by).  With xcd_gx > 0 the launch is 1-D and the slices of a row block sit on one XCD.
    int bx = bidx, by = bidy, gxd = gdimx;
    if (TEPI && p.sl_gx_small > 0) {
        // 1-D grid of the tiled-epilogue data gradient: eight consecutive workgroups = row blocks 8 g .. 8 g + 7 of one slice; the
        // first sl_rem slices (one channel tile more) have sl_gx_big row blocks, the others sl_gx_small
        const int xcd = bidx & 7, slot = bidx >> 3;
        const int gb = (p.sl_gx_big + 7) >> 3, gs = (p.sl_gx_small + 7) >> 3;
        if (slot < p.sl_rem * gb) {
            by = slot / gb;
            bx = (slot - by * gb) * 8 + xcd;
            gxd = p.sl_gx_big;
        } else {
            const int s2 = slot - p.sl_rem * gb;
            const int q = s2 / gs;
            by = p.sl_rem + q;
            bx = (s2 - q * gs) * 8 + xcd;
            gxd = p.sl_gx_small;
        }
        if (bx >= gxd) return;                         // (padding blocks of a group of eight; before any barrier)
    } else if (p.xcd_gx > 0) {
        const int L = bidx, xcd = L & 7, slot = L >> 3;
        by = slot % p.xcd_gy;
        bx = (slot / p.xcd_gy) * 8 + xcd;
        gxd = p.xcd_gx;
        if (bx >= gxd) return;                         // (padding blocks of the last group of eight; before any barrier)
    }
    // Column slice of this block.  TEPI (round 4): the 32-column tiles of the output are dealt to the col_slices slices as evenly as
    // they go -- slice `by` owns base or base + 1 consecutive tiles (5 tiles on 2 slices: 3 + 2; NT = the larger count) -- so a
    // slice never contracts an all-zero tile; `ntl` = this block's live tiles (NT or NT - 1, block-uniform).
    int n0 = by * NB, ntl = NT;
    if (TEPI && p.col_slices > 0) {
        const int nc32 = (p.Nout + 31) >> 5;
        const int base = nc32 / p.col_slices, rem = nc32 - base * p.col_slices;
        n0 = (by * base + (by < rem ? by : rem)) * 32;
        ntl = base + (by < rem ? 1 : 0);
    }

    // ---- one-time block setup: B operand -> LDS, BN scale/shift tables ------------------------
    if constexpr (EMU) {
        // operand slot (chunk ch, step t, channel tile nt, lane) = the 8 k-values lane (column nt * 32 + (lane & 31), half lane >> 5) feeds
        // step t of chunk ch with: float4 rows 8 ch + 4 t + half and + 2 of the packed operand (the channels 8 q + 4 half .. + 3 of
        // q = 2 t, 2 t + 1 the A side holds), cut into the three planes
        u32x4* Bw = reinterpret_cast<u32x4*>(smem);
        const int total = (brows >> 3) * 2 * NT * 64;
        for (int base = tid; base < total; base += blockDim.x * 4) {
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * blockDim.x;
                const int ic = idx < total ? idx : 0;
                const int l = ic & 63, nt = (ic >> 6) % NT, ct = (ic >> 6) / NT;      // ct = ch * 2 + t
                const int row = (ct >> 1) * 8 + (ct & 1) * 4 + (l >> 5);
                const int n = n0 + nt * 32 + (l & 31);
                const int nn = n < p.Npad ? n : 0;
                v[u][0] = ldg4(p.wB + ((size_t)row * p.Npad + nn) * 4);
                v[u][1] = ldg4(p.wB + ((size_t)(row + 2) * p.Npad + nn) * 4);
                if (n >= p.Npad) { v[u][0] = make_float4(0.f, 0.f, 0.f, 0.f); v[u][1] = v[u][0]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * blockDim.x;
                if (idx < total) {
                    const int l = idx & 63, nt = (idx >> 6) % NT, ct = (idx >> 6) / NT;
                    const float f[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
                    u32x4 h, m, lo;
                    split_bf16x3(f, h, m, lo);
                    u32x4* d = Bw + ((size_t)ct * 3 * NT + nt) * 64 + l;
                    d[0] = h; d[NT * 64] = m; d[2 * NT * 64] = lo;
                }
            }
        }
    } else if (!CUNET_DBG(p, 32)) {
        // 8 independent 16-byte loads in flight per thread (a load-then-store loop would expose one full
        // L2/HBM round trip per iteration: the copy of a 147 KB 3x3 operand dominated the small launches)
        const int total = brows * NB;
        for (int base = tid; base < total; base += blockDim.x * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                const int ic = idx < total ? idx : 0;
                const int row = ic / NB;
                const int n = ic - row * NB;
                const int nn = (n0 + n < p.Npad) ? n0 + n : 0;
                v[u] = ldg4(p.wB + ((size_t)row * p.Npad + nn) * 4);
                if (n0 + n >= p.Npad) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                if (idx < total) Bs[idx] = v[u];
            }
        }
    }
    constexpr bool HAS_CONCAT = (LD == LD_SEG || LD == LD_3X3 || EP == EP_BWD);
    if (HAS_CONCAT && !CUNET_DBG(p, 64)) setup_concat<EP == EP_BWD, XB>(p, grp, sc, sh, mu, is);
    for (int i = tid; i < NB * 2; i += blockDim.x) redbuf[i] = 0.0;
    __syncthreads();

    const int HW = p.H * p.W;
    const int nck = p.Kpad >> 5;                     // 32-channel chunks per tap
    const int nchunks = p.taps * nck;
    const int ntiles = (p.M + 31) >> 5;              // 32-row tiles, one per wave at a time

    double dsum[NT], dsq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dsum[nt] = 0.0; dsq[nt] = 0.0; }

    if (!CUNET_DBG(p, 128))
    {
    // Loader state of the tile whose A operand is being REQUESTED.  On the fast path the next tile's first chunk is requested
    // before the current tile's epilogue: vmcnt counts loads and stores in one in-order queue, so a load issued after the
    // epilogue's 16 (x NT) stores can only be waited for together with them -- every tile of a multi-tile wave paid the
    // store-acknowledge latency before its first MFMA.
    int mc = 0, nimg = 0, py = 0, px = 0, rowU = 0;
    auto set_tile = [&](int t) {
        const int m = t * 32 + li;                   // this lane's A row
        mc = m < p.M ? m : p.M - 1;
        if (p.wshift >= 0) {                         // power-of-two geometry (uniform): shifts instead of emulated divisions
            nimg = mc >> p.hwshift;
            const int rem = mc & (HW - 1);
            py = rem >> p.wshift;
            px = rem & (p.W - 1);
        } else {
            nimg = mc / HW;
            const int rem = mc - nimg * HW;
            py = rem / p.W;
            px = rem - py * p.W;
        }
        rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
    };
    // ---- loaders -------------------------------------------------------------------
    auto tap_row = [&](int t, bool& valid) -> int {
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        const int yy = py + dy, xx = px + dx;
        valid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        return valid ? mc + dy * p.W + dx : mc;
    };
    // fast-path loader state (see below)
    int sidx = 0, cl = 0, ncs = 0, tap = 0;          // uniform loop state
    const float* rowptr = nullptr;                    // per lane: &X[row][4*hi] of the current segment/tap
    bool tvalid = true;
    auto enter = [&]() {                              // (re)compute rowptr for (sidx | tap)
        if (LD == LD_SEG) {
            const Seg sg = p.seg[sidx];
            rowptr = sg.x + (size_t)(sg.ups ? rowU : mc) * sg.ld + 4 * hi;
            ncs = sg.C >> 5;
        } else if (LD == LD_3X3 || LD == LD_PLAIN3) {
            const int row = tap_row(tap, tvalid);
            const float* base = (LD == LD_3X3) ? p.seg[0].x : p.a;
            const int ld = (LD == LD_3X3) ? p.seg[0].ld : p.lda;
            rowptr = xadv<GB>(base, (size_t)row * ld + 4 * hi);
            ncs = nck;
        } else {
            rowptr = xadv<GB>(p.a, (size_t)mc * p.lda + 4 * hi);
            ncs = nck;
        }
        cl = 0;
    };
    auto fetch = [&](float4 (&a)[4]) {                // loads chunk (sidx|tap, cl)
        if (LD == LD_PLAIN && CUNET_DBG(p, 256)) {
            // tuning builds, TIMING ONLY (the operand is wrong): the same 4 KB of the chunk requested as 8 rows x 128 contiguous bytes per
            // instruction instead of 32 rows x 32 bytes -- what a request pattern that is kind to the vector L1 would buy
            const float* base = p.a + (size_t)(mc - li + (lane >> 3)) * p.lda + (lane & 7) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = ldx4<GB>(base + (size_t)(8 * q) * p.lda, cl * 32);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = ldx4<GB>(rowptr, cl * 32 + q * 8);
        if (LD == LD_PLAIN3 && !tvalid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 anext[4];
    bool vcur = true;
    auto begin_tile = [&](int t) {                    // geometry of tile t, its first chunk on the way
        set_tile(t);
        sidx = 0; tap = 0;
        enter();
        vcur = tvalid;
        fetch(anext);
    };
    constexpr bool EARLY_NEXT = NT <= 2;              // (NT = 3 / 4 hold 48 / 64 accumulators: 16 more live registers across their epilogue spill)
    const int tstride = gxd * nwaves;
    int tile = bx * nwaves + wave;
    if (FAST && EARLY_NEXT && !PF2 && tile < ntiles) begin_tile(tile);
    // TEPI: this lane's four 16-byte pieces of a 32 x 32 tile of x (see the epilogue): piece column 4 * pc4 of the channel tile,
    // rows pr0 + 8 j.  The group entry (segment pointer, pitch, up-sample flag) of a channel tile is re-read from LDS where it is needed
    // (four registers per tile otherwise).
    const int pc4 = lane & 7, pr0 = lane >> 3;
    float4 xp[TEPI ? 4 : 1];
    auto request_x = [&](const GrpEnt& g, int t, float4 (&o)[TEPI ? 4 : 1]) {      // row tile t, the channel tile g describes
        if constexpr (TEPI) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mm = t * 32 + pr0 + 8 * j;
                int row = mm;
                if (p.any_ups && g.ups) {                 // (a branch around arithmetic only: the load below stays unconditional)
                    int ni, yy, xx;
                    if (p.wshift >= 0) {
                        ni = mm >> p.hwshift;
                        const int rm = mm & (HW - 1);
                        yy = rm >> p.wshift;
                        xx = rm & (p.W - 1);
                    } else {
                        ni = mm / HW;
                        const int rm = mm - ni * HW;
                        yy = rm / p.W;
                        xx = rm - yy * p.W;
                    }
                    row = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
                }
                o[j] = ldg4(g.ptr + (size_t)row * g.ld);
            }
        }
    };
    if constexpr (FUSEW) {
        // ---- data gradient + weight gradient of a 1x1 node in one pass (round 4) --------------------------------------------------
        //     dz[m][c] = sum_n dY[m][n] W[n][c]  (masked)          dW[n][c] += sum_m dY[m][n] z[m][c],   z = relu(bn(x))
        // The wave that owns row tile m0 .. m0+31 for the block's 32 columns has both operands of the second contraction in hand:
        // the dY chunk (32 rows x 32 output channels) it is feeding to the data gradient, and the x tile of the epilogue.  Per chunk:
        // the dY pieces also go into a wave-private LDS tile, come back TRANSPOSED (lane = output channel; 16 row values, in the row
        // order rho(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi of the accumulator layout -- the order a contraction runs in is free), and
        // contract with z, which the lane already holds in exactly that layout from the epilogue's column read: 16 more MFMAs into one
        // of four persistent [32 channels x 32 columns] accumulators.  64 + 64 MFMAs per tile on two independent chains; dY and x are
        // read once per node instead of twice, and the wgrad3 launch of the node (low-priority stream, same CUs) disappears.  At the
        // end the block's waves add their accumulators through LDS in a fixed order and the block stores its partial tile
        // part[bx][128][32 columns]; the bucket's reduce kernel sums the row blocks (deterministic, like wgrad3).
        float* T2base = tileT + (size_t)nwaves * 32 * TEPI_PITCH;
        float* T = tileT + (size_t)wave * 32 * TEPI_PITCH;
        float* T2 = T2base + (size_t)wave * 32 * TEPI_PITCH;
        f32x16 accW[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[t][r] = 0.f;
        const int col = n0 + li;                               // (FAST: every column of the slice exists)
        const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
        GrpEnt pg0 = grp[(n0 + 4 * pc4) >> 2];
        if (tile < ntiles) request_x(pg0, tile, xp);
        for (; tile < ntiles; tile += tstride) {
            // x tile -> T -> this lane's column (16 rows rho(r, hi)): z for the weight gradient, x kept for the BatchNorm reductions
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * TEPI_PITCH + 4 * pc4) = xp[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float xv[16], zv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = T[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH + li];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zv[r] = fmaxf(fmaf(xv[r], csc, csh), 0.f);
                if (p.qin_bits) zv[r] = quan_input_act(zv[r], p.qin_bits);      // the conv's input is QuanInput(relu(bn(x)))
            }
            const bool more = tile + tstride < ntiles;
            if (more) request_x(pg0, tile + tstride, xp);       // the next tile's x, in flight across this tile's 128 MFMAs
            f32x16 accD;
#pragma unroll
            for (int r = 0; r < 16; ++r) accD[r] = 0.f;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {                  // K = 128 output channels = four 32-channel chunks (launcher precondition)
                float4 acur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acur[q] = anext[q];
                if (ch + 1 < 4) { ++cl; fetch(anext); }
                else if (more) begin_tile(tile + tstride);      // the next tile's first chunk
                // dY chunk -> T2[row li][channel 8 q + 4 hi ..]
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(T2 + li * TEPI_PITCH + 8 * q + 4 * hi) = acur[q];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float4* bb = Bs + (size_t)ch * 8 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = bb[(2 * q + hi) * NB + li];
                    accD = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].x, bv.x, accD, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].y, bv.y, accD, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].z, bv.z, accD, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].w, bv.w, accD, 0, 0, 0);
                }
                float at[16];                                  // dY^T: output channel 32 ch + li of rows rho(r, hi)
#pragma unroll
                for (int r = 0; r < 16; ++r) at[r] = T2[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH + li];
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[r], zv[r], accW[ch], 0, 0, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next chunk overwrites T2)
                __builtin_amdgcn_wave_barrier();
            }
            // BatchNorm / ReLU backward, first half, as the TEPI epilogue below
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? accD[r] : 0.f;
                s1 += dz;
                s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
                T[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH + li] = dz;
            }
            atomicAdd(&redbuf[li * 2 + 0], (double)s1);
            atomicAdd(&redbuf[li * 2 + 1], (double)s2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = pr0 + 8 * j;
                *reinterpret_cast<float4*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + n0 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * TEPI_PITCH + 4 * pc4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
            __builtin_amdgcn_wave_barrier();
        }
        // ---- the block's partial weight-gradient tile: waves added two at a time, in wave order, through LDS (B, the tables' neighbours
        // and the tiles are dead: [2][4][1024] floats from the start of the dynamic LDS ... behind the tables, which the statistics
        // epilogue below still needs -- so: in the tile area, 2 x 16 KB <= 2 tiles x 8 waves x 4.5 KB only from 4 waves up: the launcher
        // gives a fused block at least four waves)
        __syncthreads();
        float* wred = tileT;                                   // [2][4 * 1024]
        float tot[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) tot[k] = 0.f;
        const int nthr = blockDim.x;
        for (int round = 0; round < nwaves; round += 2) {
            if (wave == round || wave == round + 1) {
                float* dst = wred + (wave - round) * 4096;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[t * 1024 + r * 64 + lane] = accW[t][r];
            }
            __syncthreads();
            const bool two = round + 1 < nwaves;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = tid + k * nthr;
                if (e < 4096) tot[k] += two ? (wred[e] + wred[4096 + e]) : wred[e];
            }
            __syncthreads();
        }
        float* part = p.wg_part + (size_t)bx * p.K * p.Nout;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = tid + k * nthr;
            if (e < 4096) {
                const int t = e >> 10, r = (e >> 6) & 15, l = e & 63;
                const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                part[(size_t)n * p.Nout + n0 + (l & 31)] = tot[k];
            }
        }
    }
    if constexpr (PF2) {
        // ---- 1x1 data gradient, K = 128 = four chunks, two chunks ahead (round 4) -------------------------------------------------------
        // The generic loop keeps ONE chunk (4 KB per wave) on the way while the previous one is contracted: 16 MFMAs = 1024 matrix-pipe
        // cycles (x 3 waves sharing the SIMD) to cover a round trip that takes 2 - 3 us when the rows come from HBM.  Here the four chunks
        // of a tile have fixed register sets A0 .. A3 and chunk c + 2 is requested when chunk c is consumed -- across the tile boundary
        // too (the next tile's chunks 0 / 1 behind this tile's chunks 2 / 3, i.e. before its epilogue's stores), without a branch around
        // any request (the last tile re-requests its own rows), so the compiler's counted vmcnt waits stay exact.
        float* T = tileT + (size_t)wave * 32 * TEPI_PITCH;
        const int col = n0 + li;
        const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
        const GrpEnt pg0 = grp[(n0 + 4 * pc4) >> 2];
        float4 A0[4], A1[4], A2[4], A3[4];
        auto rowp = [&](int t) { return p.a + (size_t)(t * 32 + li) * p.lda + 4 * hi; };
        auto req = [&](const float* rp, int c, float4 (&a)[4]) {
            if (CUNET_DBG(p, 2048)) return;                // tuning builds: no dY requests (timing only)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = ldg4(rp + c * 32 + q * 8);
        };
        f32x16 acc1;
        auto mfma16 = [&](int ch, const float4 (&a)[4]) {
            if (CUNET_DBG(p, 4)) return;                   // tuning builds: no contraction
            const float4* bb = Bs + (size_t)ch * 8 * NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = bb[(2 * q + hi) * NB + li];
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bv.x, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bv.y, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bv.z, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bv.w, acc1, 0, 0, 0);
            }
        };
        const float* rp = rowp(tile < ntiles ? tile : 0);
        CUNET_STAMP(tk1);
        unsigned long long ph[5] = {0, 0, 0, 0, 0};
        unsigned ntl_done = 0;
        if (tile < ntiles) { req(rp, 0, A0); req(rp, 1, A1); }
        __builtin_amdgcn_sched_barrier(0);
        for (; tile < ntiles; tile += tstride) {
            CUNET_STAMP(ts0);
            if (!CUNET_DBG(p, 8)) request_x(pg0, tile, xp);      // (tuning builds: 8 = no x loads)
            const float* rpn = rowp(tile + tstride < ntiles ? tile + tstride : tile);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
            // (sched_barrier: left alone the scheduler sinks each request next to its first use -- two MFMA groups of cover instead of 32)
            req(rp, 2, A2);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(0, A0);
            __builtin_amdgcn_sched_barrier(0);
            req(rp, 3, A3);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(1, A1);
            __builtin_amdgcn_sched_barrier(0);
            req(rpn, 0, A0);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(2, A2);
            __builtin_amdgcn_sched_barrier(0);
            req(rpn, 1, A1);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(3, A3);
            __builtin_amdgcn_sched_barrier(0);
            rp = rpn;
            CUNET_STAMP(ts1);
            // the TEPI epilogue of one channel tile (see below)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * TEPI_PITCH + 4 * pc4) = xp[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            CUNET_STAMP(ts2);
            float s1 = 0.f, s2 = 0.f;
            float* tcol = T + li;
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc1[r] : 0.f;
                s1 += dz;
                s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
                xv[r] = dz;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH] = xv[r];
            CUNET_STAMP(ts3);
            atomicAdd(&redbuf[li * 2 + 0], (double)s1);
            atomicAdd(&redbuf[li * 2 + 1], (double)s2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            CUNET_STAMP(ts4);
            if (CUNET_DBG(p, 2)) {                             // tuning builds: no dz stores
            } else if (CUNET_DBG(p, 1024)) {                   // tuning builds: the dz pieces as non-temporal stores
                typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = pr0 + 8 * j;
                    __builtin_nontemporal_store(*reinterpret_cast<const f32x4v*>(T + rr * TEPI_PITCH + 4 * pc4),
                                                reinterpret_cast<f32x4v*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + n0 + 4 * pc4));
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = pr0 + 8 * j;
                *reinterpret_cast<float4*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + n0 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * TEPI_PITCH + 4 * pc4);
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
            __builtin_amdgcn_wave_barrier();
            CUNET_STAMP(ts5);
            ph[0] += ts1 - ts0; ph[1] += ts2 - ts1; ph[2] += ts3 - ts2; ph[3] += ts4 - ts3; ph[4] += ts5 - ts4;
            ++ntl_done;
        }
        (void)ntl_done;
#ifdef CUNET_TUNING
        if (CUNET_DBG(p, 512) && lane == 0) {
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_conv_phase[0], (unsigned long long)ntl_done);
            for (int i = 0; i < 5; ++i) atomicAdd(&g_conv_phase[1 + i], ph[i]);
            atomicAdd(&g_conv_phase[6], tk1 - tk0);
            atomicAdd(&g_conv_phase[7], te - tk0);
        }
#endif
    }
    if constexpr (!FUSEW && !PF2)
    for (; tile < ntiles; tile += tstride) {
        if (!FAST) set_tile(tile);
        if (FAST && !EARLY_NEXT) begin_tile(tile);
        if constexpr (TEPI) {
            // channel tile 0's pieces are requested HERE, in front of the row tile's MFMAs, not in the epilogue: x does not depend
            // on them, and a wave that asks after its MFMAs sits out a full memory round trip per tile with nothing of its own to do
            GrpEnt pg0;
            pg0.ptr = p.a; pg0.ld = 0; pg0.ups = 0;      // always a valid address
            if (n0 + 4 * pc4 < p.Nout) pg0 = grp[(n0 + 4 * pc4) >> 2];
            request_x(pg0, tile, xp);
        }

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        auto load_a = [&](int ch, float4 (&a)[4]) {
            const int t = ch / nck;
            const int c = ch - t * nck;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = c * 32 + q * 8 + hi * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < p.K) {
                    if (LD == LD_SEG) {
                        const GrpEnt g = grp[kk >> 2];
                        v = ldg4(g.ptr + (size_t)(g.ups ? rowU : mc) * g.ld);
                    } else if (LD == LD_3X3) {
                        bool valid;
                        const int row = tap_row(t, valid);
                        v = ldg4(p.seg[0].x + (size_t)row * p.seg[0].ld + kk);
                    } else if (LD == LD_PLAIN) {
                        v = ldx4<GB>(p.a, (size_t)mc * p.lda + kk);
                    } else if (LD == LD_PLAIN3) {
                        bool valid;
                        const int row = tap_row(t, valid);
                        v = ldx4<GB>(p.a, (size_t)row * p.lda + kk);
                        if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {  // LD_STEM: im2col gather of the NCHW image, 7x7 stride 2 pad 3
                        float e4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = kk + e;
                            const int ci = k / 49;
                            const int r = k - ci * 49;
                            const int ky = r / 7;
                            const int kx = r - ky * 7;
                            const int iy = 2 * py - 3 + ky, ix = 2 * px - 3 + kx;
                            const bool ok = (k < p.K) && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                            e4[e] = ok ? ldg1(p.img + ((size_t)(nimg * 3 + ci) * p.IH + iy) * p.IW + ix) : 0.f;
                        }
                        v = make_float4(e4[0], e4[1], e4[2], e4[3]);
                    }
                }
                a[q] = v;
            }
        };
        auto activate = [&](int ch, float4 (&a)[4]) {   // BN + ReLU in registers (forward loaders only)
            if (LD == LD_SEG || LD == LD_3X3) {
                const int t = ch / nck;
                const int c = ch - t * nck;
                bool valid = true;
                if (LD == LD_3X3) (void)tap_row(t, valid);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = c * 32 + q * 8 + hi * 4;
                    if (kk < p.K && valid) {
                        const float4 s = *reinterpret_cast<const float4*>(sc + kk);
                        const float4 h = *reinterpret_cast<const float4*>(sh + kk);
                        a[q].x = fmaxf(fmaf(a[q].x, s.x, h.x), 0.f);
                        a[q].y = fmaxf(fmaf(a[q].y, s.y, h.y), 0.f);
                        a[q].z = fmaxf(fmaf(a[q].z, s.z, h.z), 0.f);
                        a[q].w = fmaxf(fmaf(a[q].w, s.w, h.w), 0.f);
                        if (p.qin_bits) {                       // QuanInput2d between the ReLU and the conv (uniform branch)
                            a[q].x = quan_input_act(a[q].x, p.qin_bits); a[q].y = quan_input_act(a[q].y, p.qin_bits);
                            a[q].z = quan_input_act(a[q].z, p.qin_bits); a[q].w = quan_input_act(a[q].w, p.qin_bits);
                        }
                    } else {
                        a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        };

        auto mfma_live = [&](int ch, const float4 (&acur)[4], auto ltag) {      // the first L channel tiles of the slice
            constexpr int L = decltype(ltag)::value;
            if constexpr (EMU) {
                const u32x4* bb = Bp + (size_t)ch * 2 * 3 * NT * 64 + lane;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float f[8] = {acur[2 * t].x, acur[2 * t].y, acur[2 * t].z, acur[2 * t].w,
                                        acur[2 * t + 1].x, acur[2 * t + 1].y, acur[2 * t + 1].z, acur[2 * t + 1].w};
                    u32x4 ah, am, al;
                    split_bf16x3(f, ah, am, al);
#pragma unroll
                    for (int nt = 0; nt < L; ++nt) {
                        const u32x4 bh = bb[((t * 3 + 0) * NT + nt) * 64], bm = bb[((t * 3 + 1) * NT + nt) * 64], bl = bb[((t * 3 + 2) * NT + nt) * 64];
                        acc[nt] = mfma_split6(ah, am, al, bh, bm, bl, acc[nt]);
                    }
                    // (three or four accumulator tiles: the second step's pieces are not cut while the first step's are live)
                    if constexpr (L >= 3) __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            const float4* bb = Bs + (size_t)ch * 8 * NB;     // chunk ch = (tap, c): rows (t*kq4 + c*8) ..+7
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 bv[L];
#pragma unroll
                for (int nt = 0; nt < L; ++nt) bv[nt] = bb[(2 * q + hi) * NB + nt * 32 + li];
#pragma unroll
                for (int nt = 0; nt < L; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].x, bv[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < L; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].y, bv[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < L; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].z, bv[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < L; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].w, bv[nt].w, acc[nt], 0, 0, 0);
            }
        };
        auto mfma_chunk = [&](int ch, const float4 (&acur)[4]) {
            if constexpr (TEPI && NT > 1) {                  // (block-uniform: the slice that carries one tile fewer)
                if (ntl < NT) { mfma_live(ch, acur, std::integral_constant<int, NT - 1>{}); return; }
            }
            mfma_live(ch, acur, std::integral_constant<int, NT>{});
        };

        if (FAST) {
            // ---- fast path: every segment and K are multiples of 32 channels and M of 32 rows, so a
            // chunk never straddles a segment and nothing is predicated: straight-line code, the
            // segment descriptor is wave-uniform (scalar loads from the kernarg), the per-lane row
            // pointer is recomputed only when the segment (or tap) changes, loads take immediates.
            for (int ch = 0; ch < nchunks; ++ch) {
                float4 acur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acur[q] = anext[q];
                const bool vthis = vcur;
                if (ch + 1 < nchunks) {
                    if (++cl == ncs) { ++sidx; ++tap; enter(); }
                    vcur = tvalid;
                    fetch(anext);
                }
                if (LD == LD_SEG || LD == LD_3X3) {           // BN + ReLU in registers
                    const int cc = (LD == LD_3X3) ? (ch % nck) : ch;
                    const float* scp = sc + cc * 32 + 4 * hi;
                    const float* shp = sh + cc * 32 + 4 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 s4 = *reinterpret_cast<const float4*>(scp + q * 8);
                        const float4 h4 = *reinterpret_cast<const float4*>(shp + q * 8);
                        acur[q].x = fmaxf(fmaf(acur[q].x, s4.x, h4.x), 0.f);
                        acur[q].y = fmaxf(fmaf(acur[q].y, s4.y, h4.y), 0.f);
                        acur[q].z = fmaxf(fmaf(acur[q].z, s4.z, h4.z), 0.f);
                        acur[q].w = fmaxf(fmaf(acur[q].w, s4.w, h4.w), 0.f);
                        if (p.qin_bits) {
                            acur[q].x = quan_input_act(acur[q].x, p.qin_bits); acur[q].y = quan_input_act(acur[q].y, p.qin_bits);
                            acur[q].z = quan_input_act(acur[q].z, p.qin_bits); acur[q].w = quan_input_act(acur[q].w, p.qin_bits);
                        }
                        if (LD == LD_3X3 && !vthis) acur[q] = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding is post-activation
                    }
                }
                mfma_chunk(ch, acur);
            }
            if (!TEPI && EARLY_NEXT && tile + tstride < ntiles) begin_tile(tile + tstride);      // before this tile's epilogue (see above)
        } else {
            // ---- generic path: per-group table look-ups and predicates (odd channel counts, ragged M)
            load_a(0, anext);
            for (int ch = 0; ch < nchunks; ++ch) {
                float4 acur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acur[q] = anext[q];
                activate(ch, acur);
                if (ch + 1 < nchunks) load_a(ch + 1, anext);
                mfma_chunk(ch, acur);
            }
        }

        // ---- epilogue ---------------------------------------------------------------------
        // C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int mrow0 = tile * 32;
        if constexpr (TEPI) {
            // BatchNorm / ReLU backward, first half, with global memory seen in 16-byte pieces.  The element-wise version below
            // issues 16 four-byte x loads and 16 four-byte dz stores per lane and tile next to 64 MFMAs -- its cost was the
            // per-CU rate of those 32 instructions, not their bytes (without them the kernel ran 24 % faster, round 2).  Here a
            // lane requests FOUR 16-byte pieces of x (4 channels of one row; 8 lanes cover 128 contiguous bytes), the pieces go
            // into the wave's LDS tile T[32][36], the tile is read back in the accumulator layout, dz overwrites the x it came
            // from, and leaves as four 16-byte stores per lane.
            // Round 4: a wave owns up to NT channel tiles of its 32 rows (the A operand -- dY -- is read once per NT * 32 output
            // channels instead of once per 32, each A fragment feeds NT independent accumulator chains); the tiles go through T one
            // after the other, the next tile's x pieces requested while the current one is worked on.
            float* T = tileT + (size_t)wave * 32 * TEPI_PITCH;
            if (EARLY_NEXT && tile + tstride < ntiles) begin_tile(tile + tstride);      // next tile's first A chunk: behind the x requests, ahead of the stores
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt < ntl) {                                           // block-uniform
                    const int pcol = n0 + nt * 32 + 4 * pc4;
                    const bool pok = pcol < p.Nout;
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * TEPI_PITCH + 4 * pc4) = xp[j];
                    if (nt + 1 < NT && nt + 1 < ntl) {                    // the next channel tile's pieces, into the registers just emptied
                        GrpEnt g;
                        g.ptr = p.a; g.ld = 0; g.ups = 0;
                        int gi = (pcol + 32) >> 2;
                        asm volatile("" : "+v"(gi));      // (recomputed here: hoisted out of the tile loop the address costs a spilled register)
                        if (pcol + 32 < p.Nout) g = grp[gi];
                        request_x(g, tile, xp);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int col = n0 + nt * 32 + li;
                        const bool colok = col < p.Nout;
                        float csc = 0.f, csh = 0.f, cmu = 0.f, cis = 0.f;
                        if (colok) { csc = sc[col]; csh = sh[col]; cmu = mu[col]; cis = is[col]; }
                        float s1 = 0.f, s2 = 0.f;
                        float* tcol = T + li;
                        // (three passes -- reads, arithmetic, writes: a read next to a write of the same array is ordered by the compiler,
                        // i.e. 16 dependent LDS round trips per tile)
                        float xv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float z = fmaf(xv[r], csc, csh);
                            // ReLU mask; with a QuanInput behind the ReLU also its straight-through mask (no gradient where z >= 1)
                            const float dz = (colok && z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[nt][r] : 0.f;
                            s1 += dz;
                            s2 = fmaf(dz, colok ? (xv[r] - cmu) * cis : 0.f, s2);
                            xv[r] = dz;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * TEPI_PITCH] = xv[r];
                        // the BatchNorm reductions of this tile straight into the block's fp64 accumulators in LDS (no per-wave fp64
                        // registers: with NT tiles they were 4 NT VGPRs held across the whole launch)
                        if (colok) {
                            atomicAdd(&redbuf[(nt * 32 + li) * 2 + 0], (double)s1);
                            atomicAdd(&redbuf[(nt * 32 + li) * 2 + 1], (double)s2);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (pok) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int rr = pr0 + 8 * j;
                            *reinterpret_cast<float4*>(p.y + (size_t)(mrow0 + rr) * p.ldy + pcol) = *reinterpret_cast<const float4*>(T + rr * TEPI_PITCH + 4 * pc4);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
                    __builtin_amdgcn_wave_barrier();
                }
            }
            continue;
        }
        // EP_BWD: rows of this lane's 16 accumulator registers, plain and through the nearest-upsample map.
        // Everything below is branch-free per element: a branch around a load makes hipcc wait for that load
        // before issuing the next one (the first version of this epilogue spent 59 us of 177 in 64 serialised loads).
        int rowP[16], rowUp[16];
        if (EP == EP_BWD) {
            const bool w4 = (p.W & 3) == 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int m4 = mrow0 + 8 * k + 4 * hi;
                int base = 0;
                if (w4 && p.any_ups) {            // rows 4k..4k+3 share an image row: one split per group (uniform branches)
                    const int mq = m4 < p.M ? m4 : p.M - 1;
                    int ni, yy, xx;
                    if (p.wshift >= 0) {
                        ni = mq >> p.hwshift;
                        const int rm = mq & (HW - 1);
                        yy = rm >> p.wshift;
                        xx = rm & (p.W - 1);
                    } else {
                        ni = mq / HW;
                        const int rm = mq - ni * HW;
                        yy = rm / p.W;
                        xx = rm - yy * p.W;
                    }
                    base = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int mm = m4 + j;
                    mm = (FAST || mm < p.M) ? mm : p.M - 1;
                    rowP[4 * k + j] = mm;
                    if (!p.any_ups) {
                        rowUp[4 * k + j] = mm;    // never read through the up-sample map
                    } else if (w4) {
                        rowUp[4 * k + j] = base + (j >> 1);
                    } else {
                        const int ni = mm / HW;
                        const int rm = mm - ni * HW;
                        const int yy = rm / p.W;
                        const int xx = rm - yy * p.W;
                        rowUp[4 * k + j] = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
                    }
                }
            }
        }
        float xq[2][16];
        auto bwd_fetch = [&](int nt, float (&xo)[16]) {      // X[row][col] of this lane's 16 rows, channel tile nt
            const int col = n0 + nt * 32 + li;
            const bool colok = col < p.Nout;
            GrpEnt g;
            g.ptr = p.a; g.ld = 0; g.ups = 0;                 // always a valid address: loads stay branch-free
            if (colok) g = grp[col >> 2];
            const size_t xoff0 = colok ? (col & 3) : 0;
            const bool up = g.ups != 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = up ? rowUp[r] : rowP[r];
                if (CUNET_DBG(p, 8)) { xo[r] = 1.f; continue; }      // (tuning builds: timing without the x loads)
                xo[r] = ldx1<XB>(g.ptr, xoff0 + (size_t)row * g.ld);
            }
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            const bool colok = col < p.Nout;
            float s1 = 0.f, s2 = 0.f;
            if (MSE) {
                // heat-map head with the pixelwise MSE fused in (uniform branch): the target values are requested first
                const float ginv = (float)(2.0 * p.mse_inv);
                const bool padcol = !colok && col < p.mse_ldd;      // pad columns of d(loss)/d(out) must read as zero downstream
                float tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int mq = (FAST || mm < p.M) ? mm : p.M - 1;
                    tv[r] = ldg1(p.mse_tgt + (size_t)mq * p.ldy + (colok ? col : 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (FAST || mm < p.M) {
                        if (colok) {
                            const float v = acc[nt][r];
                            p.y[(size_t)mm * p.ldy + col] = v;
                            const float d = v - tv[r];
                            if (p.mse_gbf16) stx1<1>(p.mse_dout, (size_t)mm * p.mse_ldd + col, d * ginv);
                            else p.mse_dout[(size_t)mm * p.mse_ldd + col] = d * ginv;
                            s1 = fmaf(d, d, s1);
                        } else if (padcol) {
                            if (p.mse_gbf16) stx1<1>(p.mse_dout, (size_t)mm * p.mse_ldd + col, 0.f);
                            else p.mse_dout[(size_t)mm * p.mse_ldd + col] = 0.f;
                        }
                    }
                }
            } else if (EP == EP_FWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if ((FAST || mm < p.M) && colok) {
                        const float v = acc[nt][r];
                        p.y[(size_t)mm * p.ldy + col] = v;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
            } else {
                // BatchNorm/ReLU backward, first half: dz = relu'(z) * dA, reductions sum(dz), sum(dz*xhat).
                // The 16 X values of the NEXT channel tile are requested before this tile's are consumed.
                if (nt == 0) bwd_fetch(0, xq[0]);
                if (nt + 1 < NT) bwd_fetch(nt + 1, xq[(nt + 1) & 1]);
                float csc = 0.f, csh = 0.f, cmu = 0.f, cis = 0.f;
                if (colok) { csc = sc[col]; csh = sh[col]; cmu = mu[col]; cis = is[col]; }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if ((FAST || mm < p.M) && colok) {
                        const float xv = xq[nt & 1][r];
                        const float z = fmaf(xv, csc, csh);
                        // ReLU mask; with a QuanInput behind the ReLU also its straight-through mask (utils/quantize.py:58-63:
                        // no gradient where the activation is >= 1, i.e. z >= 1)
                        const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[nt][r] : 0.f;
                        if (!CUNET_DBG(p, 2)) stx1<GB>(p.y, (size_t)mm * p.ldy + col, dz);      // (tuning builds: timing without the dz stores)
                        s1 += dz;
                        s2 = fmaf(dz, (xv - cmu) * cis, s2);
                    }
                }
            }
            dsum[nt] += (double)s1;
            dsq[nt] += (double)s2;
        }
    }
    }

    if (MSE) {      // fused MSE: sum of squared errors of the block -> one fp64 atomic
        double t = 0.0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t += dsum[nt];
        for (int o = 32; o > 0; o >>= 1) t += shfl_xor_d(t, o);
        if (lane == 0) atomicAdd(&redbuf[0], t);
        __syncthreads();
        if (tid == 0) atomic_add_f64(p.mse_acc, redbuf[0] * p.mse_inv);
        return;
    }
    // ---- per-channel reductions: lanes (l, l+32) -> waves (serialised through LDS) -> one fp64
    //      atomic per channel per block
    if (p.ystats != nullptr && !CUNET_DBG(p, 1)) {
        if constexpr (!TEPI) {                          // (TEPI: every tile has added its sums already)
            double a[NT], b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                a[nt] = dsum[nt] + shfl_xor_d(dsum[nt], 32);
                b[nt] = dsq[nt] + shfl_xor_d(dsq[nt], 32);
            }
            if (hi == 0) {                              // LDS fp64 atomics: one barrier instead of one per wave
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    atomicAdd(&redbuf[(nt * 32 + li) * 2 + 0], a[nt]);
                    atomicAdd(&redbuf[(nt * 32 + li) * 2 + 1], b[nt]);
                }
            }
        }
        __syncthreads();
        if (tid < ntl * 32) {                           // (a slice's dead tile belongs to its neighbour)
            const int col = n0 + tid;
            if (col < p.Nout) {
                atomic_add_f64(p.ystats + col, redbuf[tid * 2 + 0]);
                atomic_add_f64(p.ystats + p.Nout + col, redbuf[tid * 2 + 1]);
            }
        }
    }
}

template <int LD, int EP, int NT, bool FAST, int XBG = 0>
__global__ __launch_bounds__((conv_max_waves<LD, EP, NT, FAST, XBG>() * 64)) void conv_kernel(const ConvArgs p) {
    conv_body<LD, EP, NT, FAST, XBG>(p, blockIdx.x, blockIdx.y, gridDim.x);
}

// Two convolutions of ONE shape in one launch -- the ahead and the skip adapter of a down block read the same concat
// (models/cu_net.py:139-142), and their data gradients are adjacent in the backward: gridDim.z = 2, blockIdx.z picks the problem
// (its own weights, BatchNorm, output and statistics), each on half of the workgroups a single launch would use.
template <int LD, int EP, int NT, bool FAST, int XBG = 0>
__global__ __launch_bounds__((conv_max_waves<LD, EP, NT, FAST, XBG>() * 64)) void conv_pair_kernel(const ConvPair q) {
    conv_body<LD, EP, NT, FAST, XBG>(q.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// 1x1 data gradient with dY staged ONCE per row tile (round 4): dz[M][Ccat] = dY[M][128] . W[128][Ccat], masked, + BatchNorm reductions.
// The weight-stationary kernel above cuts the Ccat output channels into 32-column slices on different workgroups: every slice re-reads
// the whole dY (5 - 10 times per node, from L2), and its phase clocks show a wave waiting ~40 k cycles per tile for 4 k cycles of MFMAs
// -- on requests landing and on stores being ACCEPTED, i.e. on the CU's vector-memory path (deeper prefetch, other request shapes and
// more accumulator chains per wave all measured +-1 %).  What goes through that path is what counts, so here
//   * a workgroup owns ALL Ccat channels of its row tiles: wave w computes columns 32 w .. 32 w + 31 (Ccat / 32 = 4 .. 10 waves);
//   * its slice of the weights (128 x 32) lives in 64 REGISTERS per lane for the whole launch -- no LDS operand, no per-chunk B reads;
//   * the 32 x 128 dY tile goes from HBM into an LDS ring ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4, no VGPRs), in MFMA
//     fragment order: request q of a tile = the 64 16-byte pieces (row l & 31, piece 2 q + (l >> 5)) step q of the contraction wants, so
//     every wave takes its A fragment with ONE conflict-free ds_read_b128 at slot + 1024 q + 16 l.  Three slots: the requests of tile
//     t + 2 are issued when tile t starts, one barrier per tile;
//   * epilogue per wave as in the kernel above (x pieces -> wave-private LDS tile -> column pass -> dz pieces out, fp64 sums in LDS).
// dY, x and dz cross the CU's memory path once each: 1/3 ... 1/2 of the bytes the sliced kernel moves.
constexpr int DR_SLOT = 32 * 128 * 4;              // bytes of one dY tile
constexpr int DR_SLOTS = 3;
constexpr int DR_MAX_WAVES = 10;

__device__ __forceinline__ void dr_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(DR_MAX_WAVES * 64) void dgrad1x1_rows_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;                // = Ccat / 32
    const int li = lane & 31;
    const int hi = lane >> 5;
    // LDS: [ring 3 x 16 KB][T tiles: waves x 32 x 36 floats][group table][sc sh mu is][fp64 sums 2 x Ccat]
    char* ring = smem;
    float* tileT = reinterpret_cast<float*>(smem + DR_SLOTS * DR_SLOT);
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(tileT + (size_t)nwaves * 32 * 36);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat);      // [Ccat][2]
    const unsigned ring0 = (unsigned)(size_t)ring;

    setup_concat<true, 0>(p, grp, sc, sh, mu, is);
    for (int i = tid; i < 2 * p.Ccat; i += blockDim.x) redbuf[i] = 0.0;

    // this wave's 128 x 32 slice of the backward operand [k / 4][Npad][4]: b[q] = (k = 8 q + 4 hi + 0..3, column 32 wave + li)
    float4 b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = ldg4(p.wB + ((size_t)(2 * q + hi) * p.Npad + wave * 32 + li) * 4);

    const int ntiles = p.M >> 5;
    const int gstride = gridDim.x;
    int tile = blockIdx.x;
    // LDS-DMA requests of a tile: q = wave, wave + nwaves, ... < 16
    auto issue = [&](int t, int slot) {
        const float* src = p.a + (size_t)(t * 32 + li) * p.lda + 4 * hi;
        if (CUNET_DBG(p, 2048)) return;                // tuning builds: no dY requests (stale LDS contents: timing only)
        for (int q = wave; q < 16; q += nwaves)
            dr_dma16(src + 8 * q, __builtin_amdgcn_readfirstlane(ring0 + (unsigned)(slot * DR_SLOT + q * 1024)));
    };
    if (tile < ntiles) issue(tile, 0);
    if (tile + gstride < ntiles) issue(tile + gstride, 1);
    __syncthreads();                                   // tables visible

    const int col = wave * 32 + li;
    const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
    const int pc4 = lane & 7, pr0 = lane >> 3;        // this lane's 16-byte pieces of a 32 x 32 tile: column piece, first row (+ 8 j)
    const GrpEnt pg = grp[(wave * 32 + 4 * pc4) >> 2];
    const int HW = p.H * p.W;
    float4 xp[4], xn[4];
    auto request_x = [&](int t, float4 (&o)[4]) {
        if (CUNET_DBG(p, 8)) { o[0] = o[1] = o[2] = o[3] = make_float4(1.f, 1.f, 1.f, 1.f); return; }      // tuning builds: no x loads
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = t * 32 + pr0 + 8 * j;
            int row = mm;
            if (p.any_ups && pg.ups) {                 // (a branch around arithmetic only)
                int ni, yy, xx;
                if (p.wshift >= 0) { ni = mm >> p.hwshift; const int rm = mm & (HW - 1); yy = rm >> p.wshift; xx = rm & (p.W - 1); }
                else { ni = mm / HW; const int rm = mm - ni * HW; yy = rm / p.W; xx = rm - yy * p.W; }
                row = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
            }
            o[j] = ldg4(pg.ptr + (size_t)row * pg.ld);
        }
    };
    if (tile < ntiles) request_x(tile, xp);
    float* T = tileT + (size_t)wave * 32 * 36;
    int slot = 0;
    for (; tile < ntiles; tile += gstride) {
        // everything this wave has requested (its pieces of this tile's dY, issued a whole tile ago; this tile's x) has landed ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // ... and so have the other waves' pieces; the slot of tile - 1 is free
        const int nslot = slot == 0 ? 2 : slot - 1;     // (slot + 2) % 3
        if (tile + 2 * gstride < ntiles) issue(tile + 2 * gstride, nslot);
        const bool more = tile + gstride < ntiles;
        request_x(more ? tile + gstride : tile, xn);    // next tile's x: in flight across this tile (no branch around a request)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float4* A = reinterpret_cast<const float4*>(ring + slot * DR_SLOT) + lane;
        float4 a_cur = A[0];
        if (!CUNET_DBG(p, 4))                          // tuning builds: no contraction
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 a_nxt = A[64 * (q + 1 < 16 ? q + 1 : q)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b[q].w, acc, 0, 0, 0);
            a_cur = a_nxt;
        }
        // BatchNorm / ReLU backward, first half (as conv_body's LDS-tile epilogue)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * 36 + 4 * pc4) = xp[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float s1 = 0.f, s2 = 0.f;
        float* tcol = T + li;
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = fmaf(xv[r], csc, csh);
            const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[r] : 0.f;      // ReLU mask (+ QuanInput's straight-through mask)
            s1 += dz;
            s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
            xv[r] = dz;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36] = xv[r];
        atomicAdd(&redbuf[col * 2 + 0], (double)s1);
        atomicAdd(&redbuf[col * 2 + 1], (double)s2);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!CUNET_DBG(p, 2))                          // tuning builds: no dz stores
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = pr0 + 8 * j;
            *reinterpret_cast<float4*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + wave * 32 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * 36 + 4 * pc4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = xn[j];
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.ystats != nullptr && tid < p.Ccat) {
        atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
        atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
    }
}

// The row-tile data gradient with the contraction on the bf16 matrix pipe (planner options f32_split + dgrad_rows).  In the column-sliced
// kernel every slice cuts the same dY chunk into its bf16 pieces again (5 - 10 times per node: that kernel is bound by this VALU work once
// the MFMAs are cheap); here the 32 x 128 dY tile is cut ONCE per workgroup: it arrives in the LDS ring by LDS-DMA as above, all threads
// together turn it into three planes of MFMA operands (8 k-steps x 3 planes x 64 lanes x 16 bytes = 24 KB, double buffered), and a wave's
// contraction is 24 ds_read_b128 + 48 MFMAs with its weights -- cut once per launch -- in 96 registers.  The cut of tile t + 1 runs in
// front of the MFMAs of tile t, so one barrier per tile still covers both the DMA and the planes.  Ccat / 32 waves above 8 do not fit the
// register budget: the columns are dealt to column groups of at most 8 waves (blockIdx.x % cgroups), each staging dY for itself.
constexpr int DRS_PLANES = 8 * 3 * 64 * 16;       // bytes of one tile's operand planes
constexpr int DRS_MAX_WAVES = 8;

__global__ __launch_bounds__(DRS_MAX_WAVES * 64, 2) void dgrad1x1_rows_split_kernel(const ConvArgs p, int cgroups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    // LDS: [ring 3 x 16 KB][planes 2 x 24 KB][T tiles: waves x 32 x 36 floats][group table][sc sh mu is][fp64 sums 2 x Ccat]
    char* ring = smem;
    u32x4* planes = reinterpret_cast<u32x4*>(smem + DR_SLOTS * DR_SLOT);
    float* tileT = reinterpret_cast<float*>(smem + DR_SLOTS * DR_SLOT + 2 * DRS_PLANES);
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(tileT + (size_t)nwaves * 32 * 36);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat);      // [Ccat][2]
    const unsigned ring0 = (unsigned)(size_t)ring;

    setup_concat<true, 0>(p, grp, sc, sh, mu, is);
    for (int i = tid; i < 2 * p.Ccat; i += blockDim.x) redbuf[i] = 0.0;

    const int group = blockIdx.x % cgroups;
    const int col0 = (group * nwaves + wave) * 32;                // this wave's 32 columns
    const bool active = col0 < p.Ccat;                            // (the last group may carry idle waves: they stage and cut with the others)
    const int colw = active ? col0 : 0;
    // this wave's 128 x 32 slice of the backward operand [k / 4][Npad][4], cut: step s, lane (column li, half hi) = k 16 s + 4 hi + 0..3 and
    // 16 s + 8 + 4 hi + 0..3 (float4 rows 4 s + hi and + 2) -- the channels the dY pieces of requests 2 s and 2 s + 1 carry for that lane
    u32x4 bh[8], bm[8], bl[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        const float4 w0 = ldg4(p.wB + ((size_t)(4 * s8 + hi) * p.Npad + colw + li) * 4);
        const float4 w1 = ldg4(p.wB + ((size_t)(4 * s8 + 2 + hi) * p.Npad + colw + li) * 4);
        const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        split_bf16x3(f, bh[s8], bm[s8], bl[s8]);
    }

    const int ntiles = p.M >> 5;
    const int gstride = gridDim.x / cgroups;
    int tile = blockIdx.x / cgroups;
    auto issue = [&](int t, int slot) {                           // LDS-DMA requests of a tile: q = wave, wave + nwaves, ... < 16
        const float* src = p.a + (size_t)(t * 32 + li) * p.lda + 4 * hi;
        for (int q = wave; q < 16; q += nwaves)
            dr_dma16(src + 8 * q, __builtin_amdgcn_readfirstlane(ring0 + (unsigned)(slot * DR_SLOT + q * 1024)));
    };
    auto cut = [&](int slot, int pb) {                            // ring slot -> operand planes pb: item (step s, lane l) = requests 2 s, 2 s + 1
        const float4* R = reinterpret_cast<const float4*>(ring + slot * DR_SLOT);
        u32x4* P = planes + (size_t)pb * (DRS_PLANES / 16);
        for (int i = tid; i < 512; i += blockDim.x) {
            const int s8 = i >> 6, l = i & 63;
            const float4 a0 = R[(2 * s8) * 64 + l], a1 = R[(2 * s8 + 1) * 64 + l];
            const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            u32x4 h, m, lo;
            split_bf16x3(f, h, m, lo);
            P[(s8 * 3 + 0) * 64 + l] = h;
            P[(s8 * 3 + 1) * 64 + l] = m;
            P[(s8 * 3 + 2) * 64 + l] = lo;
        }
    };
    for (int k = 0; k < 3; ++k)
        if (tile + k * gstride < ntiles) issue(tile + k * gstride, k);

    const int pc4 = lane & 7, pr0 = lane >> 3;        // this lane's 16-byte pieces of a 32 x 32 tile: column piece, first row (+ 8 j)
    const int HW = p.H * p.W;
    __syncthreads();                                   // tables visible
    const int col = colw + li;
    const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
    const GrpEnt pg = grp[(colw + 4 * pc4) >> 2];
    float4 xp[4], xn[4];
    auto request_x = [&](int t, float4 (&o)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = t * 32 + pr0 + 8 * j;
            int row = mm;
            if (p.any_ups && pg.ups) {                 // (a branch around arithmetic only)
                int ni, yy, xx;
                if (p.wshift >= 0) { ni = mm >> p.hwshift; const int rm = mm & (HW - 1); yy = rm >> p.wshift; xx = rm & (p.W - 1); }
                else { ni = mm / HW; const int rm = mm - ni * HW; yy = rm / p.W; xx = rm - yy * p.W; }
                row = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
            }
            o[j] = ldg4(pg.ptr + (size_t)row * pg.ld);
        }
    };
    if (tile < ntiles) request_x(tile, xp);
    // the first tile's planes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tile < ntiles) cut(0, 0);
    float* T = tileT + (size_t)wave * 32 * 36;
    int k = 0;                                         // tiles done by this workgroup: ring slot k % 3, planes k & 1
    for (; tile < ntiles; tile += gstride, ++k) {
        // this wave's DMA pieces of the next tiles and this tile's x have landed ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // ... everyone's have; planes k & 1 are complete; ring slot k % 3 (cut one tile ago) is free
        if (tile + 3 * gstride < ntiles) issue(tile + 3 * gstride, k % 3);
        const bool more = tile + gstride < ntiles;
        request_x(more ? tile + gstride : tile, xn);   // next tile's x: in flight across this tile (no branch around a request)
        if (more) cut((k + 1) % 3, (k + 1) & 1);       // the next tile's operands, in front of this tile's MFMAs
        if (active) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const u32x4* P = planes + (size_t)(k & 1) * (DRS_PLANES / 16) + lane;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8)
                acc = mfma_split6(P[(s8 * 3 + 0) * 64], P[(s8 * 3 + 1) * 64], P[(s8 * 3 + 2) * 64], bh[s8], bm[s8], bl[s8], acc);
            // BatchNorm / ReLU backward, first half (as conv_body's LDS-tile epilogue)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * 36 + 4 * pc4) = xp[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float s1 = 0.f, s2 = 0.f;
            float* tcol = T + li;
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[r] : 0.f;      // ReLU mask (+ QuanInput's straight-through mask)
                s1 += dz;
                s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
                xv[r] = dz;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36] = xv[r];
            atomicAdd(&redbuf[col * 2 + 0], (double)s1);
            atomicAdd(&redbuf[col * 2 + 1], (double)s2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = pr0 + 8 * j;
                *reinterpret_cast<float4*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + col0 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * 36 + 4 * pc4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = xn[j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.ystats != nullptr) {
        for (int c = tid; c < nwaves * 32; c += blockDim.x) {      // this column group's sums
            const int cc = group * nwaves * 32 + c;
            if (cc < p.Ccat) {
                atomic_add_f64(p.ystats + cc, redbuf[cc * 2 + 0]);
                atomic_add_f64(p.ystats + p.Nout + cc, redbuf[cc * 2 + 1]);
            }
        }
    }
}

// Round 5: the same kernel with the vector-memory queue under the kernel's own control (planner option dgrad_rows_v = 2, the default).
// What the round-4 kernel's ISA showed (tools/isa_scan.py and a reading of the tile loop): (1) every iteration began with
// `s_waitcnt vmcnt(0)` and ended -- compiler-inserted, in front of the register moves that rotate the x look-ahead -- with another
// vmcnt(0): the dz stores a wave had issued a few instructions earlier had to be ACKNOWLEDGED before the wave could even reach the
// barrier, a full store round trip exposed per 32-row tile, twice; (2) the 24 ds_read_b128 of a tile's operand planes were issued three
// at a time and waited for right in front of their MFMAs; (3) 128 lane-wise fp64 LDS atomics per wave and tile for the BatchNorm sums.
// Here (1) the x pieces are requested by inline asm (invisible to the compiler's counter bookkeeping, like the LDS-DMA requests), go to
// the wave's LDS tile at the TOP of the next iteration and their registers are re-used for the next request at once -- no look-ahead
// copy, no compiler wait -- and the one wait per iteration is `vmcnt(4)`: everything this wave has asked for has landed EXCEPT the four
// dz stores of the previous tile, which complete under this tile's work; (2) the planes of step s + 1 are requested in front of the six
// MFMAs of step s (pinned with sched_barrier); (3) the sums stay in two fp64 registers per lane until the end of the launch.
typedef float f32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void drs_req16(f32x4a& v, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
}
template <int N> __device__ __forceinline__ void drs_wait(f32x4a& a, f32x4a& b, f32x4a& c, f32x4a& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

__global__ __launch_bounds__(DRS_MAX_WAVES * 64, 2) void dgrad1x1_rows_split2_kernel(const ConvArgs p, int cgroups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef CUNET_TUNING
    const unsigned long long tk_entry = CUNET_DBG(p, 16384) ? __builtin_amdgcn_s_memtime() : 0ull;      // (the wave's first instruction: set-up = [7] - the phases)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    // LDS: [ring 3 x 16 KB][planes 2 x 24 KB][T tiles: waves x 32 x 36 floats][group table][sc sh mu is][fp64 sums 2 x Ccat]
    char* ring = smem;
    u32x4* planes = reinterpret_cast<u32x4*>(smem + DR_SLOTS * DR_SLOT);
    float* tileT = reinterpret_cast<float*>(smem + DR_SLOTS * DR_SLOT + 2 * DRS_PLANES);
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(tileT + (size_t)nwaves * 32 * 36);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat);      // [Ccat][2]
    const unsigned ring0 = (unsigned)(size_t)ring;

    setup_concat<true, 0>(p, grp, sc, sh, mu, is);
    for (int i = tid; i < 2 * p.Ccat; i += blockDim.x) redbuf[i] = 0.0;

    const int group = blockIdx.x % cgroups;
    const int col0 = (group * nwaves + wave) * 32;                // this wave's 32 columns
    const bool active = col0 < p.Ccat;                            // (the last group may carry idle waves: they stage and cut with the others)
    const int colw = active ? col0 : 0;
    u32x4 bh[8], bm[8], bl[8];                                    // this wave's 128 x 32 slice of the backward operand, cut once per launch
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        const float4 w0 = ldg4(p.wB + ((size_t)(4 * s8 + hi) * p.Npad + colw + li) * 4);
        const float4 w1 = ldg4(p.wB + ((size_t)(4 * s8 + 2 + hi) * p.Npad + colw + li) * 4);
        const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        split_bf16x3(f, bh[s8], bm[s8], bl[s8]);
    }

    const int ntiles = p.M >> 5;
    const int gstride = gridDim.x / cgroups;
    int tile = blockIdx.x / cgroups;
    auto issue = [&](int t, int slot) {                           // LDS-DMA requests of a tile: q = wave, wave + nwaves, ... < 16
        const float* src = p.a + (size_t)(t * 32 + li) * p.lda + 4 * hi;
        for (int q = wave; q < 16; q += nwaves)
            dr_dma16(src + 8 * q, __builtin_amdgcn_readfirstlane(ring0 + (unsigned)(slot * DR_SLOT + q * 1024)));
    };
    auto cut = [&](int slot, int pb) {                            // ring slot -> operand planes pb: item (step s, lane l) = requests 2 s, 2 s + 1
        const float4* R = reinterpret_cast<const float4*>(ring + slot * DR_SLOT);
        u32x4* P = planes + (size_t)pb * (DRS_PLANES / 16);
        for (int i = tid; i < 512; i += blockDim.x) {
            const int s8 = i >> 6, l = i & 63;
            const float4 a0 = R[(2 * s8) * 64 + l], a1 = R[(2 * s8 + 1) * 64 + l];
            const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            u32x4 h, m, lo;
            split_bf16x3(f, h, m, lo);
            P[(s8 * 3 + 0) * 64 + l] = h;
            P[(s8 * 3 + 1) * 64 + l] = m;
            P[(s8 * 3 + 2) * 64 + l] = lo;
        }
    };
    for (int k = 0; k < 3; ++k)
        if (tile + k * gstride < ntiles) issue(tile + k * gstride, k);

    const int pc4 = lane & 7, pr0 = lane >> 3;        // this lane's 16-byte pieces of a 32 x 32 tile: column piece, first row (+ 8 j)
    const int HW = p.H * p.W;
    __syncthreads();                                   // tables visible
    const int col = colw + li;
    const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
    const GrpEnt pg = grp[(colw + 4 * pc4) >> 2];
    f32x4a xq[4];                                      // the x pieces on the way (asm requests: the compiler does not count them)
    auto request_x = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = t * 32 + pr0 + 8 * j;
            int row = mm;
            if (p.any_ups && pg.ups) {                 // (a branch around arithmetic only)
                int ni, yy, xx;
                if (p.wshift >= 0) { ni = mm >> p.hwshift; const int rm = mm & (HW - 1); yy = rm >> p.wshift; xx = rm & (p.W - 1); }
                else { ni = mm / HW; const int rm = mm - ni * HW; yy = rm / p.W; xx = rm - yy * p.W; }
                row = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
            }
            drs_req16(xq[j], pg.ptr + (size_t)row * pg.ld);
        }
    };
    request_x(tile < ntiles ? tile : 0);
    drs_wait<0>(xq[0], xq[1], xq[2], xq[3]);           // the first tiles' DMA pieces and x
    __syncthreads();
    if (tile < ntiles) cut(0, 0);
    float* T = tileT + (size_t)wave * 32 * 36;
    double sum1 = 0.0, sum2 = 0.0;                     // this lane's share of sum(dz), sum(dz * xhat) of column `col`
    // tuning builds, CUNET_CONV_DBG & 16384: shader cycles of a wave's phases summed over tiles into g_conv_phase (tools/ring_phase_clocks.py --rows):
    // [1] the counted wait, [2] barrier, [3] x -> LDS tile + the next requests (LDS-DMA, x), [4] cut of the next tile's planes, [5] MFMAs
    // (plane reads + chain), [6] epilogue + dz stores; [0] wave-tiles, [7] whole kernel
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const bool stamp = CUNET_DBG(p, 16384) != 0;
    auto now = [&]() -> unsigned long long { return stamp ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long tk0 = now();
    int k = 0;                                         // tiles done by this workgroup: ring slot k % 3, planes k & 1
    for (; tile < ntiles; tile += gstride, ++k) {
        // Everything this wave has requested has landed -- its DMA pieces of the next tiles, this tile's x -- except the previous tile's
        // dz stores (the youngest four entries of the in-order queue), which finish under this tile.
        const unsigned long long t0 = now();
        // (round 6: ONE register-tied wait.  With a second arm -- `else drs_wait<0>(xq ...)` -- hipcc merged the two arms' register
        // assignments with v_mov copies of xq placed IN FRONT of that arm's s_waitcnt: harmless only because the arm was taken when nothing
        // was in flight (k == 0) or by idle waves; tests/test_kernel_resources.py reads the ISA for exactly this.  k == 0: the requests
        // were waited for in front of the loop.  Idle waves never read xq: an untied wait for their DMA pieces.)
        if (active) { if (k > 0) drs_wait<4>(xq[0], xq[1], xq[2], xq[3]); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = now();
        __syncthreads();                               // everyone's pieces have; planes k & 1 are complete; ring slot k % 3 (cut one tile ago) is free
        const unsigned long long t2 = now();
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4a*>(T + (pr0 + 8 * j) * 36 + 4 * pc4) = xq[j];
        }
        if (tile + 3 * gstride < ntiles) issue(tile + 3 * gstride, k % 3);
        const bool more = tile + gstride < ntiles;
        request_x(more ? tile + gstride : tile);       // next tile's x, into the registers just written out (no branch around a request)
        const unsigned long long t3 = now();
        if (more) cut((k + 1) % 3, (k + 1) & 1);       // the next tile's operands, in front of this tile's MFMAs
        unsigned long long t4 = now(), t5 = t4;
        if (active) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const u32x4* P = planes + (size_t)(k & 1) * (DRS_PLANES / 16) + lane;
            u32x4 ah = P[0], am = P[64], al = P[128];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                u32x4 nh = ah, nm = am, nl = al;
                if (s8 < 7) {
                    nh = P[((s8 + 1) * 3 + 0) * 64];
                    nm = P[((s8 + 1) * 3 + 1) * 64];
                    nl = P[((s8 + 1) * 3 + 2) * 64];
                    __builtin_amdgcn_sched_barrier(0);         // (the requests stay in front of this step's MFMAs)
                }
                acc = mfma_split6(ah, am, al, bh[s8], bm[s8], bl[s8], acc);
                ah = nh; am = nm; al = nl;
            }
            if (stamp) { asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[15])); t5 = now(); }      // (the chain has landed before the stamp)
            // BatchNorm / ReLU backward, first half (as conv_body's LDS-tile epilogue); x is in T since the top of the iteration
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float s1 = 0.f, s2 = 0.f;
            float* tcol = T + li;
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[r] : 0.f;      // ReLU mask (+ QuanInput's straight-through mask)
                s1 += dz;
                s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
                xv[r] = dz;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36] = xv[r];
            sum1 += (double)s1;
            sum2 += (double)s2;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = pr0 + 8 * j;
                *reinterpret_cast<float4*>(p.y + (size_t)(tile * 32 + rr) * p.ldy + col0 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * 36 + 4 * pc4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's x pieces overwrite T)
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long t6 = now();
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5;
    }
#ifdef CUNET_TUNING
    if (stamp && lane == 0 && active) {
        const unsigned long long tk1 = now();      // (stamped BEFORE this wave queues its atomics: ~10 k same-address atomics per launch back up the memory path, which round 5's "whole kernel" column included)
        atomicAdd(&g_conv_phase[0], (unsigned long long)k);
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_phase[1 + i], ph[i]);
        atomicAdd(&g_conv_phase[7], tk1 - tk_entry);
    }
#endif
    (void)tk0; (void)ph;
    drs_wait<0>(xq[0], xq[1], xq[2], xq[3]);
    if (active) {
        atomicAdd(&redbuf[col * 2 + 0], sum1);
        atomicAdd(&redbuf[col * 2 + 1], sum2);
    }
    __syncthreads();
    if (p.ystats != nullptr) {
        for (int c = tid; c < nwaves * 32; c += blockDim.x) {      // this column group's sums
            const int cc = group * nwaves * 32 + c;
            if (cc < p.Ccat) {
                atomic_add_f64(p.ystats + cc, redbuf[cc * 2 + 0]);
                atomic_add_f64(p.ystats + p.Nout + cc, redbuf[cc * 2 + 1]);
            }
        }
    }
}

static bool dgrad1x1_rows_supported(const ConvArgs& a) {
    if (a.taps != 1 || a.K != 128 || a.Kpad != 128 || a.xbf16 || a.M % 32 || a.Nout % 32 || a.Nout != a.Ccat || a.ldy != a.Nout || a.lda % 4 ||
        a.Nout / 32 < 4 || a.Nout / 32 > DR_MAX_WAVES || a.wg_part != nullptr || a.mse_tgt != nullptr)
        return false;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 32 || a.seg[i].ld % 4) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Stem forward (conv0 7x7 / 2, pad 3, 3 -> 128 channels, models/cu_net.py:300) on image rows in LDS, split contraction (round 4).
// conv_kernel<LD_STEM> gathers its A operand element by element through an im2col address computation per load (206 us, the longest
// launch of the step; bound by that VALU work).  Here a 512-thread workgroup walks OUTPUT rows: the seven input rows an output row
// reads live in an LDS ring of eight row slots, each element cut into its three bf16 pieces ONCE on the way in (a slot = 3 channels x
// 3 planes x (IW + 8) bf16, four zero elements in front: element e = ix + 4).  The contraction's k is re-ordered so that a lane's eight
// k-values of a step are eight CONSECUTIVE input pixels of one (channel, kernel row): for output pixel ox they start at e = 2 ox -- a
// dword-aligned LDS address with no arithmetic beyond a per-row offset -- and cover kx = -1 .. 6, the first with a zero weight.  Step
// s = (channel, kernel row) pairs 2 s and 2 s + 1 (one per lane half), 11 steps for the 21 pairs.  Wave w owns output-channel tile w & 3
// and two of the row's four 32-pixel tiles, its weights cut once per launch in 132 registers.  The two new input rows of the next
// output row are requested before the MFMAs of the current one.
constexpr int STF_STEPS = 11;

__global__ __launch_bounds__(512, 2) void stem_fwd_split_kernel(const ConvArgs p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int IW = p.IW, IH = p.IH, OW = p.W, OH = p.H;
    const int RP = 2 * IW + 16;                                   // bytes of one (channel, plane) row
    const int SLOT = 9 * RP;
    char* ring = smem;                                            // 8 slots
    double* redbuf = reinterpret_cast<double*>(smem + 8 * SLOT);  // [128][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int nt = wave & 3;
    const int th = wave >> 2;                                     // tiles 2 th, 2 th + 1 of the row (and th + 2 ... for wider rows)
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
    unsigned yoff[4];                                             // byte offset of this lane's element in pixel row 8 g + 4 hi of a 32-pixel output tile
#pragma unroll
    for (int g = 0; g < 4; ++g) yoff[g] = 4u * (unsigned)((8 * g + 4 * hi) * p.ldy + nt * 32 + li);

    for (int i = tid; i < 8 * SLOT / 16; i += 512) reinterpret_cast<float4*>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 256; i += 512) redbuf[i] = 0.0;

    // weights: step s, lane (column 32 nt + li, half hi) = pair q = 2 s + hi = (channel q / 7, kernel row q % 7), slot j = kx + 1 (j = 0
    // and the 22nd pair: zero); packed operand [k / 4][Npad][4] with k = 49 c + 7 ky + kx
    u32x4 bh[STF_STEPS], bm[STF_STEPS], bl[STF_STEPS];
    const int q_lane0 = hi;                                       // q = 2 s + hi
#pragma unroll
    for (int s8 = 0; s8 < STF_STEPS; ++s8) {
        const int q = 2 * s8 + q_lane0;
        float f[8];
        f[0] = 0.f;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            const int k = q * 7 + (j - 1);                        // = 49 c + 7 ky + kx
            f[j] = (q < 21) ? ldg1(p.wB + ((size_t)(k >> 2) * p.Npad + nt * 32 + li) * 4 + (k & 3)) : 0.f;
        }
        split_bf16x3(f, bh[s8], bm[s8], bl[s8]);
    }

    // staging plan: an input row is 3 channels x IW / 4 float4; two rows per output row: thread t takes float4 t of the pair
    const int per_row = 3 * (IW >> 2);
    const int NR = p.M / OW;                                     // output rows in the batch
    const int r_begin = blockIdx.x * rows_per_wg;
    int r_end = r_begin + rows_per_wg;
    if (r_end > NR) r_end = NR;
    auto stage_item = [&](int n, int iy, int it, f32x4n& v) {     // request float4 `it` of input row iy of image n (zeros outside the image)
        const int c = it / (IW >> 2), x4 = (it - c * (IW >> 2)) << 2;
        const bool ok = iy >= 0 && iy < IH;
        const f32x4n z = {0.f, 0.f, 0.f, 0.f};
        v = ok ? *reinterpret_cast<const f32x4n*>(p.img + ((size_t)(n * 3 + c) * IH + iy) * IW + x4) : z;
    };
    auto commit_item = [&](int iy, int it, const f32x4n& v) {     // cut and write into slot (iy + 3) & 7
        const int c = it / (IW >> 2), x4 = (it - c * (IW >> 2)) << 2;
        char* d = ring + (size_t)((iy + 3) & 7) * SLOT + (size_t)(c * 3) * RP + (x4 + 4) * 2;
        u32x2n ph, pm, pl;
        unsigned a0, a1, a2;
        split_bf16x3_pair(f32x2_op{v[0], v[1]}, a0, a1, a2);
        ph[0] = a0; pm[0] = a1; pl[0] = a2;
        split_bf16x3_pair(f32x2_op{v[2], v[3]}, a0, a1, a2);
        ph[1] = a0; pm[1] = a1; pl[1] = a2;
        *reinterpret_cast<u32x2n*>(d) = ph;
        *reinterpret_cast<u32x2n*>(d + RP) = pm;
        *reinterpret_cast<u32x2n*>(d + 2 * RP) = pl;
    };
    auto stage_rows_blocking = [&](int n, int oy, int ky0, int ky1) {      // input rows 2 oy - 3 + ky, ky0 <= ky < ky1
        for (int ky = ky0; ky < ky1; ++ky)
            for (int it = tid; it < per_row; it += 512) {
                f32x4n v;
                stage_item(n, 2 * oy - 3 + ky, it, v);
                commit_item(2 * oy - 3 + ky, it, v);
            }
    };
    __syncthreads();                                              // zeros in place
    if (r_begin < r_end) stage_rows_blocking(r_begin / OH, r_begin % OH, 0, 7);
    __syncthreads();

    double dsum = 0.0, dsq = 0.0;                                 // this lane's column, summed over its rows
    for (int R = r_begin; R < r_end; ++R) {
        const int n = R / OH, oy = R - n * OH;
        // the next output row's two new input rows (2 oy + 4, 2 oy + 5): requested now, cut into the ring after this row's MFMAs
        const bool more = R + 1 < r_end;
        const bool next_same = more && oy + 1 < OH;
        f32x4n pv[2];
        int pit[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pit[u] = tid + 512 * u;
            const int row = pit[u] / per_row, it = pit[u] - row * per_row;
            const f32x4n z = {0.f, 0.f, 0.f, 0.f};
            pv[u] = z;
            if (next_same && pit[u] < 2 * per_row) stage_item(n, 2 * oy + 4 + row, it, pv[u]);
        }
        const int ntx = OW >> 5;
        for (int tx = 2 * th; tx < ntx; tx += 4) {                // (two consecutive tiles per wave and pass: OW = 128 -> one pass)
#pragma unroll 1
            for (int t2 = 0; t2 < 2; ++t2) {
                const int txx = tx + t2;
                if (txx >= ntx) break;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const char* abase = ring + 4 * (txx * 32 + li);
                auto fetch_a = [&](int s8, u32x4& h, u32x4& m, u32x4& l) {
                    // LDS row of this lane's (channel, kernel row) pair of the step (computed here: eleven offsets held across the tile
                    // loop cost eleven registers of a budget the weights fill)
                    int q = 2 * s8 + hi;
                    q = q < 21 ? q : 20;                          // (the 22nd pair: any valid address, its weights are zero)
                    const int c = (q * 37) >> 8, ky = q - c * 7;  // q / 7 for q <= 20
                    const int ro = ((2 * oy + ky) & 7) * SLOT + c * 3 * RP;
                    const unsigned* a0 = reinterpret_cast<const unsigned*>(abase + ro);
                    const unsigned* a1 = reinterpret_cast<const unsigned*>(abase + ro + RP);
                    const unsigned* a2 = reinterpret_cast<const unsigned*>(abase + ro + 2 * RP);
                    h = u32x4{a0[0], a0[1], a0[2], a0[3]};
                    m = u32x4{a1[0], a1[1], a1[2], a1[3]};
                    l = u32x4{a2[0], a2[1], a2[2], a2[3]};
                };
                // one step ahead, pinned: left alone hipcc requests all eleven steps' fragments of both tiles up front (231 spilled registers)
                u32x4 ah, am, al;
                fetch_a(0, ah, am, al);
#pragma unroll
                for (int s8 = 0; s8 < STF_STEPS; ++s8) {
                    u32x4 nh = ah, nm = am, nl = al;
                    if (s8 + 1 < STF_STEPS) fetch_a(s8 + 1, nh, nm, nl);
                    acc = mfma_split6(ah, am, al, bh[s8], bm[s8], bl[s8], acc);
                    __builtin_amdgcn_sched_barrier(0);
                    ah = nh; am = nm; al = nl;
                }
                // C layout: col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 hi (pixel)
                float s1 = 0.f, s2 = 0.f;
                const size_t m0 = ((size_t)R * OW + txx * 32);
                // (round 6) a uniform base per pixel row + four kernel-constant 32-bit lane offsets: a store is one instruction (the 64-bit address of
                // every element took ~7 VALU instructions in front of its store: 103.5 -> 99.6 us, 256 registers + spills -> 236)
                const char* yb = reinterpret_cast<const char*>(p.y + m0 * p.ldy);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[r];
                    *reinterpret_cast<float*>(const_cast<char*>(yb) + (size_t)(r & 3) * p.ldy * 4 + yoff[r >> 2]) = v;
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
                dsum += (double)s1;
                dsq += (double)s2;
            }
        }
        __syncthreads();                                          // every wave is done reading this row's slots
        if (more) {
            if (next_same) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (pit[u] < 2 * per_row) {
                        const int row = pit[u] / per_row, it = pit[u] - row * per_row;
                        commit_item(2 * oy + 4 + row, it, pv[u]);
                    }
                }
            } else {
                stage_rows_blocking((R + 1) / OH, 0, 0, 7);       // a new image: all seven rows (three of them zero)
            }
        }
        __syncthreads();
    }
    if (p.ystats != nullptr) {
        const double a = dsum + shfl_xor_d(dsum, 32), b = dsq + shfl_xor_d(dsq, 32);
        if (hi == 0) {
            atomicAdd(&redbuf[(nt * 32 + li) * 2 + 0], a);
            atomicAdd(&redbuf[(nt * 32 + li) * 2 + 1], b);
        }
        __syncthreads();
        if (tid < 128) {
            atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
            atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
        }
    }
}

static bool stem_fwd_split_supported(const ConvArgs& a) {
    return a.split && a.K == 147 && a.Nout == 128 && a.Npad >= 128 && a.Kpad >= 148 && a.IW % 64 == 0 && a.IH % 2 == 0 && a.H * 2 == a.IH &&
           a.W * 2 == a.IW && a.M % (a.H * a.W) == 0 && a.IW <= 512 && a.mse_tgt == nullptr && a.qin_bits == 0;
}

static hipError_t launch_stem_fwd_split(const ConvArgs& a, int num_cus, hipStream_t s) {
    const int NR = a.M / a.W;                                     // output rows in the batch
    int rows = (NR + num_cus - 1) / num_cus;
    if (rows < 1) rows = 1;
    const int grid = (NR + rows - 1) / rows;
    const size_t smem = (size_t)8 * 9 * (2 * a.IW + 16) + 256 * 8;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(stem_fwd_split_kernel, dim3(grid), dim3(512), smem, s, a, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 3x3 data gradient on a ring of dY rows, split contraction (round 4; planner options f32_split + dgrad3_ring):
//     dz[m][c] = mask . sum_{tap, n} dY[m (+) tap][n] W_b[tap][n][c]      (32 -> 128 channels, autograd dgrad of models/cu_net.py:47)
// The column-sliced kernel gathers the nine shifted taps of dY through per-lane global loads in every one of its four column slices
// (47 us per launch alone for 113 MB at 64 x 64).  Here a 512-thread workgroup walks image rows of a 32-pixel strip as the forward's
// conv3x3_ring_split_kernel does: a dY row enters the LDS ring ONCE, cut into its three bf16 planes on the way in (pixel = 3 x 64 bytes
// + 16: conflict-free 16-byte fragment reads), a tap is an LDS offset.  Wave w owns output-channel tile (w & 3) and taps 0..4 (w < 4) or
// 5..8, its weights cut once per launch in 120 registers; the second tap group hands its partial tile to the first through LDS, which
// runs the BatchNorm / ReLU-backward epilogue of the other data gradients (x pieces -> wave-private LDS tile -> column pass -> dz pieces,
// fp64 sums in LDS).  Row g + 2 of dY and the tile's x pieces are requested before the MFMAs of row g.
constexpr int D3R_PIX = 208;            // bytes per ring pixel
constexpr int D3R_SLOT = 34 * D3R_PIX;  // bytes per ring row

__global__ __launch_bounds__(512, 2) void dgrad3x3_ring_split_kernel(const ConvArgs p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = p.W, H = p.H;
    // LDS: [ring 3 rows][part 4 x 4 KB][T tiles 4 x 32 x 36 floats][group table][sc sh mu is][fp64 sums 2 x 128]
    char* ring = smem;
    float* part = reinterpret_cast<float*>(smem + 3 * D3R_SLOT + 64);
    float* tileT = part + 4 * 1024;
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(tileT + 4 * 32 * 36);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat);      // [128][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int nt = wave & 3;
    const int grp1 = wave >> 2;                                   // tap group: 0 = taps 0..4, 1 = taps 5..8
    const int nstrip = W >> 5;
    const int strip = blockIdx.x % nstrip;
    const int x0 = strip * 32;

    setup_concat<true, 0>(p, grp, sc, sh, mu, is);
    for (int i = tid; i < 2 * 128; i += 512) redbuf[i] = 0.0;
    for (int i = tid; i < (3 * D3R_SLOT + 64) / 16; i += 512)     // pixels outside the image stay zero for the whole launch
        reinterpret_cast<float4*>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // this wave's weights, cut: tap t (5 grp1 + u), step s2 = k 16 s2 + 8 hi .. + 7, column 32 nt + li of the backward operand
    // [tap][K / 4 = 8][Npad][4] (float4 rows 8 t + 4 s2 + 2 hi, + 1)
    u32x4 bh[10], bm[10], bl[10];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        int t = grp1 * 5 + u;
        if (t > 8) t = 8;                                         // (the second group's fifth slot: not contracted)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int row = t * 8 + 4 * s2 + 2 * hi;
            const float4 w0 = ldg4(p.wB + ((size_t)row * p.Npad + nt * 32 + li) * 4);
            const float4 w1 = ldg4(p.wB + ((size_t)(row + 1) * p.Npad + nt * 32 + li) * 4);
            const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            split_bf16x3(f, bh[2 * u + s2], bm[2 * u + s2], bl[2 * u + s2]);
        }
    }

    const int NH = p.M / W;                                       // image rows in the batch
    const int g_begin = (blockIdx.x / nstrip) * rows_per_wg;
    int g_end = g_begin + rows_per_wg;
    if (g_end > NH) g_end = NH;

    // staging plan: a ring row is 34 pixels x 8 float4 of dY; thread t < 272 takes float4 t
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
    const int spix = tid >> 3, sc4 = (tid & 7) << 2;              // ring pixel 0 .. 33 (threads 0 .. 271), first channel
    const int sx = x0 - 1 + spix;
    const bool sactive = tid < 272 && sx >= 0 && sx < W;
    // (a row is requested a whole iteration before it is cut into the ring: two register sets)
    f32x4n dv, dvn;
    bool dok = false, dokn = false;
    auto issue_d = [&](int g, f32x4n& v, bool& ok) {
        ok = g >= 0 && g < NH;
        const size_t base = (size_t)(ok ? g : 0) * W + (sactive ? sx : 0);
        v = *reinterpret_cast<const f32x4n*>(p.a + base * p.lda + sc4);
    };
    auto commit_d = [&](int g) {
        if (!dok || !sactive) return;
        char* slot = ring + (size_t)(((g % 3) + 3) % 3) * D3R_SLOT;
        u32x2n ph, pm, pl;
        unsigned a0, a1, a2;
        split_bf16x3_pair(f32x2_op{dv[0], dv[1]}, a0, a1, a2);
        ph[0] = a0; pm[0] = a1; pl[0] = a2;
        split_bf16x3_pair(f32x2_op{dv[2], dv[3]}, a0, a1, a2);
        ph[1] = a0; pm[1] = a1; pl[1] = a2;
        char* d = slot + spix * D3R_PIX + sc4 * 2;
        *reinterpret_cast<u32x2n*>(d) = ph;
        *reinterpret_cast<u32x2n*>(d + 64) = pm;
        *reinterpret_cast<u32x2n*>(d + 128) = pl;
    };
    __syncthreads();                                              // zeros and tables visible
    for (int g = g_begin - 1; g <= g_begin + 1; ++g) { issue_d(g, dv, dok); commit_d(g); }
    issue_d(g_begin + 2, dvn, dokn);

    // epilogue state of the first tap group's waves: column, BatchNorm constants, the x pieces of a 32 x 32 tile
    const int col = nt * 32 + li;
    const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
    const int pc4 = lane & 7, pr0 = lane >> 3;
    const float* xbase = p.seg[0].x + nt * 32 + 4 * pc4;
    const int xld = p.seg[0].ld;
    float* T = tileT + (size_t)nt * 32 * 36;
    float* mypart = part + nt * 1024;
    float4 xp[4], xn[4];
    auto request_x = [&](int g, float4 (&o)[4]) {                 // the x pieces of row g's tile (first tap group's waves)
        const size_t mm = (size_t)g * W + x0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ldg4(xbase + (mm + pr0 + 8 * j) * xld);
    };
    if (grp1 == 0 && g_begin < g_end) request_x(g_begin, xp);
    __syncthreads();

    // tuning builds, CUNET_CONV_DBG & 4096: shader cycles of a wave's phases summed over tiles into g_conv_phase (tools/ring_phase_clocks.py):
    // [1] requests, [2] MFMAs (LDS fragment reads + chain), [3] partial hand-over + first barrier, [4] ring commit (+ partial add), [5] epilogue
    // (first tap group), [6] second barrier; [0] wave-tiles, [7] whole kernel
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const bool stamp = CUNET_DBG(p, 4096) != 0;
    auto now = [&]() -> unsigned long long { return stamp ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long tk0 = now();
    for (int g = g_begin; g < g_end; ++g) {
        const int y = g % H;
        const unsigned long long t0 = now();
        dv = dvn; dok = dokn;                                     // row g + 2, requested one iteration ago
        issue_d(g + 3, dvn, dokn);                                // in flight across this whole iteration
        const size_t m0 = (size_t)g * W + x0;                     // first output row of this tile
        if (grp1 == 0) request_x(g + 1 < g_end ? g + 1 : g, xn);  // the next tile's x pieces likewise
        const unsigned long long t1 = now();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = grp1 * 5 + u;
            if (t > 8) break;
            const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
            if (y + dy < 0 || y + dy >= H) continue;              // (block-uniform: the row outside the image contributes zeros)
            const char* ap = ring + (size_t)((g + dy + 3) % 3) * D3R_SLOT + (li + dx + 1) * D3R_PIX + 16 * hi;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ap + 32 * s2);
                const u32x4 am = *reinterpret_cast<const u32x4*>(ap + 32 * s2 + 64);
                const u32x4 al = *reinterpret_cast<const u32x4*>(ap + 32 * s2 + 128);
                acc = mfma_split6(ah, am, al, bh[2 * u + s2], bm[2 * u + s2], bl[2 * u + s2], acc);
            }
        }
        if (stamp) asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[15]));      // (the chain has landed before the stamp)
        const unsigned long long t2 = now();
        if (grp1 == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mypart[r * 64 + lane] = acc[r];
        }
        __syncthreads();                                          // partial tiles written; every wave is done reading the ring rows of g
        const unsigned long long t3 = now();
        commit_d(g + 2);                                          // (into the ring row of g - 1)
        if (grp1 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += mypart[r * 64 + lane];
        }
        const unsigned long long t4 = now();
        if (grp1 == 0) {
            // BatchNorm / ReLU backward, first half (as conv_body's LDS-tile epilogue)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + (pr0 + 8 * j) * 36 + 4 * pc4) = xp[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float s1 = 0.f, s2v = 0.f;
            float* tcol = T + li;
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[r] : 0.f;      // ReLU mask (+ QuanInput's straight-through mask)
                s1 += dz;
                s2v = fmaf(dz, (xv[r] - cmu) * cis, s2v);
                xv[r] = dz;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tcol[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36] = xv[r];
            atomicAdd(&redbuf[col * 2 + 0], (double)s1);
            atomicAdd(&redbuf[col * 2 + 1], (double)s2v);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = pr0 + 8 * j;
                *reinterpret_cast<float4*>(p.y + (m0 + rr) * p.ldy + nt * 32 + 4 * pc4) = *reinterpret_cast<const float4*>(T + rr * 36 + 4 * pc4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = xn[j];
        }
        const unsigned long long t5 = now();
        __syncthreads();                                          // the new ring row is complete; the partial tiles have been read
        const unsigned long long t6 = now();
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5;
    }
#ifdef CUNET_TUNING
    if (stamp && lane == 0) {
        const unsigned long long tk1 = now();      // (stamped BEFORE this wave queues its atomics: ~10 k same-address atomics per launch back up the memory path, which round 5's "whole kernel" column included)
        atomicAdd(&g_conv_phase[0], (unsigned long long)(g_end > g_begin ? g_end - g_begin : 0));
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_phase[1 + i], ph[i]);
        atomicAdd(&g_conv_phase[7], tk1 - tk0);
    }
#endif
    (void)tk0; (void)ph;
    if (p.ystats != nullptr && tid < 128) {
        atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
        atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
    }
}

static bool dgrad3x3_ring_supported(const ConvArgs& a) {
    return a.split && a.taps == 9 && a.K == 32 && a.Kpad == 32 && a.Nout == 128 && a.Ccat == 128 && a.ldy == 128 && a.xbf16 == 0 && a.nseg == 1 &&
           !a.seg[0].ups && a.seg[0].C == 128 && a.seg[0].ld % 4 == 0 && a.lda % 4 == 0 && (a.W == 64 || a.W == 32) && a.M % a.W == 0 &&
           a.wg_part == nullptr && a.mse_tgt == nullptr && a.Npad >= 128;
}

static hipError_t launch_dgrad3x3_ring(const ConvArgs& a_in, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    static const int dbg = tune_int("CUNET_CONV_DBG", 0);      // tuning builds only: phase clocks (4096)
    a.dbg = dbg;
    const int NH = a.M / a.W;
    const int nstrip = a.W / 32;
    int rows = (NH * nstrip + num_cus - 1) / num_cus;
    if (rows < 2) rows = 2;
    const int grid = ((NH + rows - 1) / rows) * nstrip;
    const size_t smem = (size_t)3 * D3R_SLOT + 64 + (size_t)4 * 4096 + (size_t)4 * 32 * 36 * 4 + (size_t)(a.Ccat / 4) * sizeof(GrpEnt) +
                        (size_t)a.Ccat * 16 + (size_t)128 * 16;
    hipLaunchKernelGGL(dgrad3x3_ring_split_kernel, dim3(grid), dim3(512), smem, s, a, rows);
    return hipGetLastError();
}

static hipError_t launch_dgrad1x1_rows(const ConvArgs& a_in, int num_cus, hipStream_t s) {
    ConvArgs a = a_in;
    set_geometry_shifts(a);
    static const int dbg = tune_int("CUNET_CONV_DBG", 0);      // tuning builds only: work-skipping timing experiments
    a.dbg = dbg;
    const int nw = a.Nout / 32;
    if (a.split) {                      // the contraction on the bf16 matrix pipe: column groups of at most 8 waves, one workgroup per CU
        const int cgroups = (nw + DRS_MAX_WAVES - 1) / DRS_MAX_WAVES;
        const int wpg = (nw + cgroups - 1) / cgroups;
        const size_t smem = (size_t)DR_SLOTS * DR_SLOT + 2 * DRS_PLANES + (size_t)wpg * 32 * 36 * 4 + (size_t)(a.Ccat / 4) * sizeof(GrpEnt) +
                            (size_t)a.Ccat * 16 + (size_t)a.Ccat * 16;
        const int ntiles = a.M / 32;
        int streams = num_cus / cgroups;
        if (streams > ntiles) streams = ntiles;
        if (streams < 1) streams = 1;
        static bool attr_split = false;
        if (!attr_split) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad1x1_rows_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_split = true;
        }
        if (a.dgrad_rows_v >= 2) {
            static bool attr_split2 = false;
            if (!attr_split2) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad1x1_rows_split2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr_split2 = true;
            }
            hipLaunchKernelGGL(dgrad1x1_rows_split2_kernel, dim3(streams * cgroups), dim3(wpg * 64), smem, s, a, cgroups);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(dgrad1x1_rows_split_kernel, dim3(streams * cgroups), dim3(wpg * 64), smem, s, a, cgroups);
        return hipGetLastError();
    }
    const size_t smem = (size_t)DR_SLOTS * DR_SLOT + (size_t)nw * 32 * 36 * 4 + (size_t)(a.Ccat / 4) * sizeof(GrpEnt) + (size_t)a.Ccat * 16 + (size_t)a.Ccat * 16;
    const int ntiles = a.M / 32;
    const int bpc = smem <= 80 * 1024 && nw <= 6 ? 2 : 1;         // (12 waves per CU at most: three per SIMD at the kernel's register budget)
    int grid = bpc * num_cus;
    if (grid > ntiles) grid = ntiles;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad1x1_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(dgrad1x1_rows_kernel, dim3(grid), dim3(nw * 64), smem, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 3x3 forward, tap-split (models/cu_net.py:45-48,62: norm2 -> relu2 -> conv2, 128 -> 32 channels).
// The weight-stationary kernel above walks all 9 taps x K/32 chunks in ONE wave: 36 dependent
// load -> MFMA steps, ~40 us whatever the resolution (14 of the 18 3x3 launches of a CU-Net-2 step sit on
// that floor).  Here a block is 9 waves and wave t owns tap t: its slice of the weights (K x 32) lives in
// REGISTERS for the whole launch (no LDS operand at all), every lane issues the K/8 16-byte loads of its
// shifted row together, contracts them in K/2 MFMAs, and the nine partial 32x32 tiles meet in LDS, where
// eight waves add them, store the tile and keep the per-channel batch statistics.  The loads of the next
// tile are in flight across the reduction.
template <int NCK>       // K = 32 * NCK input channels
__global__ __launch_bounds__(576) void conv3x3_tapsplit_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 32 * NCK;
    float* part = reinterpret_cast<float*>(smem);                 // [9][1024] partial tiles
    float* sc = part + 9 * 1024;                                  // [K]
    float* sh = sc + K;                                           // [K]
    double* redbuf = reinterpret_cast<double*>(sh + K);           // [32][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tap = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg sg = p.seg[0];

    for (int c = tid; c < K; c += 576) {
        double mean, istd;
        if (p.training) {
            mean = sg.stats[c] / sg.count;
            double var = sg.stats[sg.C + c] / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            istd = 1.0 / sqrt(var + (double)BN_EPS);
        } else {
            mean = (double)p.rmean[c];
            istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
        }
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
    }
    if (tid < 64) redbuf[tid] = 0.0;

    // this wave's tap of the packed weights [tap][K/4][Npad][4]: element (k = 32c + 8q + 4hi + j, n = li).
    // The first half of K stays in registers, the second half in LDS (64 + 64 registers of operands would spill).
    constexpr int NR = NCK * 2;                                   // 16-byte pieces kept in registers
    float4 bw[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i)
        bw[i] = ldg4(p.wB + ((size_t)(tap * (K / 4) + 2 * i + hi) * p.Npad + li) * 4);
    float4* bl = reinterpret_cast<float4*>(redbuf + 64) + tap * (NR * 64);   // [9][NR][64] float4
#pragma unroll
    for (int i = 0; i < NR; ++i)
        bl[i * 64 + lane] = ldg4(p.wB + ((size_t)(tap * (K / 4) + 2 * (NR + i) + hi) * p.Npad + li) * 4);
    __syncthreads();
    const int HW = p.H * p.W;
    const int ntiles = p.M >> 5;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    float4 a[NCK * 4];
    bool valid = false;
    auto fetch = [&](int tile) {                      // raw loads of this wave's shifted rows (always a valid address)
        const int m = tile * 32 + li;
        const int nimg = m / HW;
        const int rem = m - nimg * HW;
        const int py = rem / p.W;
        const int px = rem - py * p.W;
        const int yy = py + dy, xx = px + dx;
        valid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        const float* src = sg.x + (size_t)(valid ? m + dy * p.W + dx : m) * sg.ld + 4 * hi;
#pragma unroll
        for (int i = 0; i < NCK * 4; ++i) a[i] = ldg4(src + i * 8);
    };
    double dsum = 0.0, dsq = 0.0;
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {        // block-uniform trip count
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool v = valid;
#pragma unroll
        for (int i = 0; i < NCK * 4; ++i) {
            float4 t = a[i];
            const float4 s4 = *reinterpret_cast<const float4*>(sc + i * 8 + 4 * hi);
            const float4 h4 = *reinterpret_cast<const float4*>(sh + i * 8 + 4 * hi);
            t.x = v ? fmaxf(fmaf(t.x, s4.x, h4.x), 0.f) : 0.f;            // zero padding is post-activation
            t.y = v ? fmaxf(fmaf(t.y, s4.y, h4.y), 0.f) : 0.f;
            t.z = v ? fmaxf(fmaf(t.z, s4.z, h4.z), 0.f) : 0.f;
            t.w = v ? fmaxf(fmaf(t.w, s4.w, h4.w), 0.f) : 0.f;
            const float4 b = i < NR ? bw[i < NR ? i : 0] : bl[(i - NR) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w, b.w, acc, 0, 0, 0);
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);               // in flight across the reduction below
#pragma unroll
        for (int r = 0; r < 16; ++r) part[tap * 1024 + r * 64 + lane] = acc[r];
        __syncthreads();
        if (tid < 512) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + 512 * u;
                float vsum = part[e];
#pragma unroll
                for (int w = 1; w < 9; ++w) vsum += part[w * 1024 + e];
                const int r = e >> 6;                 // C layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                p.y[(size_t)(tile * 32 + row) * p.ldy + li] = vsum;
                dsum += (double)vsum;
                dsq += (double)vsum * (double)vsum;
            }
        }
        __syncthreads();
    }
    if (p.ystats != nullptr) {
        const double a1 = dsum + shfl_xor_d(dsum, 32);
        const double b1 = dsq + shfl_xor_d(dsq, 32);
        for (int w = 0; w < 8; ++w) {
            if (tap == w && hi == 0) { redbuf[li * 2 + 0] += a1; redbuf[li * 2 + 1] += b1; }
            __syncthreads();
        }
        if (tid < 32 && tid < p.Nout) {
            atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
            atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
        }
    }
}

static hipError_t launch_conv3x3_tapsplit(const ConvArgs& a, int num_cus, hipStream_t s) {
    const int ntiles = a.M / 32;
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    const size_t smem = (size_t)9 * 1024 * 4 + (size_t)a.K * 8 + 64 * 8 + (size_t)9 * (a.K / 16) * 64 * 16;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_tapsplit_kernel<4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3x3_tapsplit_kernel<4>, dim3(grid), dim3(576), smem, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 3x3 forward on a ring of activated image rows (same nodes as above at 64x64 / 32x32: 128 -> 32 channels).
// The weight-stationary kernel reads every input row through nine shifted taps from L2; at 64x64 the rows a CU touches
// between two taps of the same line (6 rows x 32 KB per block, 32 blocks per XCD) do not stay in the 4 MB L2: the counters
// show 4.7x the algorithmic fetch (79.6 MB per launch, profiles/r02_traffic.txt).  Here an 8-wave workgroup walks image ROWS:
// BatchNorm + ReLU is applied ONCE per element on the way into an LDS ring of three rows [W+2 pixels][132 floats] (zero border
// columns; pixel pitch 132 makes the 16-byte fragment reads conflict-free), wave t owns tap t (and an eighth of tap 8) with its
// weights in registers, its A fragments are ds_read_b128 of the row (y + dy) shifted by dx, the eight partial tiles meet in LDS,
// where all threads add them, store the tile and keep the output statistics.  Row g+2 is requested from HBM before the MFMAs of row g.
constexpr int R3_PITCH = 132;           // floats per pixel in the ring
constexpr int R3_THREADS = 512;         // 8 waves: wave w owns tap w and the k-slice [16w, 16w + 16) of tap 8 -- two waves per SIMD with
                                        // 72 MFMAs each per tile (nine one-tap waves would put 3 + 2 + 2 + 2 on the four SIMDs)

__global__ __launch_bounds__(R3_THREADS) void conv3x3_ring_kernel(const ConvArgs p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 128;
    const int W = p.W, H = p.H;
    const int SLOT = (W + 2) * R3_PITCH;                          // floats per ring slot
    float* sc = reinterpret_cast<float*>(smem);                   // [K]
    float* sh = sc + K;                                           // [K]
    double* redbuf = reinterpret_cast<double*>(sh + K);           // [32][2]
    float* part = reinterpret_cast<float*>(redbuf + 64);          // [8][1024] partial tiles
    float* ring = part + 8 * 1024;                                // 3 slots
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tap = tid >> 6;                                     // 0 .. 7
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg sg = p.seg[0];

    for (int c = tid; c < K; c += R3_THREADS) {
        double mean, istd;
        if (p.training) {
            mean = sg.stats[c] / sg.count;
            double var = sg.stats[sg.C + c] / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            istd = 1.0 / sqrt(var + (double)BN_EPS);
        } else {
            mean = (double)p.rmean[c];
            istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
        }
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
    }
    if (tid < 64) redbuf[tid] = 0.0;
    for (int i = tid; i < 3 * SLOT / 4; i += R3_THREADS)          // border columns stay zero for the whole launch
        reinterpret_cast<float4*>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // packed weights [tap][K/4][Npad][4]: bw[q] = (tap, k = 8q + 4hi + 0..3, n = li); bx[q] = (tap 8, k = 16 tap + 8q + 4hi + 0..3)
    float4 bw[16], bx[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) bw[q] = ldg4(p.wB + ((size_t)(tap * (K / 4) + 2 * q + hi) * p.Npad + li) * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) bx[q] = ldg4(p.wB + ((size_t)(8 * (K / 4) + 4 * tap + 2 * q + hi) * p.Npad + li) * 4);
    __syncthreads();

    const int NH = p.M / W;                                       // image rows in the batch
    const int g_begin = blockIdx.x * rows_per_wg;
    int g_end = g_begin + rows_per_wg;
    if (g_end > NH) g_end = NH;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;

    // staging plan: a row is W * 32 float4; thread t takes float4 t, t + 512, ... : its channel group never changes
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    const int nx4 = W * 32;
    const int c4 = (tid & 31) << 2;
    const float4 s4 = *reinterpret_cast<const float4*>(sc + c4);
    const float4 h4 = *reinterpret_cast<const float4*>(sh + c4);
    f32x4n xv[4];
    bool xok = false;
    auto issue_x = [&](int g) {
        xok = g >= 0 && g < NH;
        const size_t base = (size_t)(xok ? g : 0) * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int idx = tid + R3_THREADS * j;
            if (idx >= nx4) idx = nx4 - 1;
            xv[j] = *reinterpret_cast<const f32x4n*>(sg.x + (base + (idx >> 5)) * sg.ld + c4);
        }
    };
    auto commit_x = [&](int g) {
        if (!xok) return;
        float* slot = ring + (size_t)(((g % 3) + 3) % 3) * SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + R3_THREADS * j;
            if (idx >= nx4) break;
            f32x4n v;
            v[0] = fmaxf(fmaf(xv[j][0], s4.x, h4.x), 0.f);
            v[1] = fmaxf(fmaf(xv[j][1], s4.y, h4.y), 0.f);
            v[2] = fmaxf(fmaf(xv[j][2], s4.z, h4.z), 0.f);
            v[3] = fmaxf(fmaf(xv[j][3], s4.w, h4.w), 0.f);
            if (p.qin_bits) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quan_input_act(v[e], p.qin_bits);
            }
            *reinterpret_cast<f32x4n*>(slot + ((idx >> 5) + 1) * R3_PITCH + c4) = v;
        }
    };

    for (int g = g_begin - 1; g <= g_begin + 1; ++g) { issue_x(g); commit_x(g); }
    __syncthreads();

    double dsum = 0.0, dsq = 0.0;
    const int ntile = W >> 5;
    for (int g = g_begin; g < g_end; ++g) {
        const int y = g % H;
        issue_x(g + 2);                                           // in flight across this row's MFMAs
        const bool rvalid = (y + dy >= 0) && (y + dy < H);
        const bool xvalid = y + 1 < H;                            // tap 8 = (dy, dx) = (+1, +1)
        const int sl = (g + dy + 3) % 3, slx = (g + 1) % 3;
        for (int t = 0; t < ntile; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (rvalid) {
                const float* ap = ring + (size_t)sl * SLOT + (t * 32 + li + dx + 1) * R3_PITCH + 4 * hi;
                f32x4n a_cur = *reinterpret_cast<const f32x4n*>(ap);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4n a_nxt = *reinterpret_cast<const f32x4n*>(ap + 8 * (q + 1 < 16 ? q + 1 : q));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], bw[q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], bw[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], bw[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], bw[q].w, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    a_cur = a_nxt;
                }
            }
            if (xvalid) {
                const float* ap = ring + (size_t)slx * SLOT + (t * 32 + li + 2) * R3_PITCH + 16 * tap + 4 * hi;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4n a4 = *reinterpret_cast<const f32x4n*>(ap + 8 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], bx[q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], bx[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], bx[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], bx[q].w, acc, 0, 0, 0);
                }
            }
            if (t > 0) __syncthreads();                           // the previous tile's sums have been read
#pragma unroll
            for (int r = 0; r < 16; ++r) part[tap * 1024 + r * 64 + lane] = acc[r];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + R3_THREADS * u;
                float vsum = part[e];
#pragma unroll
                for (int w = 1; w < 8; ++w) vsum += part[w * 1024 + e];
                const int r = e >> 6, l = e & 63;                 // C layout: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                p.y[((size_t)g * W + t * 32 + row) * p.ldy + (l & 31)] = vsum;
                dsum += (double)vsum;
                dsq += (double)vsum * (double)vsum;
            }
        }
        __syncthreads();                                          // everyone is done with row g-1's slot and with `part`
        commit_x(g + 2);
        __syncthreads();
    }
    if (p.ystats != nullptr) {                                    // a thread's column is tid & 31 in both passes (512 = 16 * 32)
        atomicAdd(&redbuf[(tid & 31) * 2 + 0], dsum);
        atomicAdd(&redbuf[(tid & 31) * 2 + 1], dsq);
        __syncthreads();
        if (tid < 32 && tid < p.Nout) {
            atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
            atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
        }
    }
}

// The same walk with the contraction on the bf16 matrix pipe (planner option f32_split, see conv_body's XBG = 6): an element is cut
// into its three bf16 pieces ONCE, on the way into the ring (the nine taps and 32 output channels that read it share the cut), a
// ring pixel is three 256-byte planes (+ 16 bytes: conflict-free 16-byte fragment reads), a step of the contraction is 16 channels:
// lane (pixel, half) reads channels 16 s + 8 half .. + 7 of each plane, six MFMAs.  The planes make a pixel 784 bytes instead of
// 528, so a workgroup walks a 32-pixel-wide strip of the image (34 ring pixels per row; at W = 64 two strips per row block).
constexpr int R3S_PIX = 784;            // bytes per ring pixel
constexpr int R3S_SLOT = 34 * R3S_PIX;  // bytes per ring row

__global__ __launch_bounds__(R3_THREADS) void conv3x3_ring_split_kernel(const ConvArgs p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 128;
    const int W = p.W, H = p.H;
    float* sc = reinterpret_cast<float*>(smem);                   // [K]
    float* sh = sc + K;                                           // [K]
    double* redbuf = reinterpret_cast<double*>(sh + K);           // [32][2]
    float* part = reinterpret_cast<float*>(redbuf + 64);          // [8][1024] partial tiles
    char* ring = reinterpret_cast<char*>(part + 8 * 1024);        // 3 rows
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tap = tid >> 6;                                     // 0 .. 7
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg sg = p.seg[0];
    const int nstrip = W >> 5;
    const int strip = blockIdx.x % nstrip;
    const int x0 = strip * 32;

    for (int c = tid; c < K; c += R3_THREADS) {
        double mean, istd;
        if (p.training) {
            mean = sg.stats[c] / sg.count;
            double var = sg.stats[sg.C + c] / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            istd = 1.0 / sqrt(var + (double)BN_EPS);
        } else {
            mean = (double)p.rmean[c];
            istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
        }
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
    }
    if (tid < 64) redbuf[tid] = 0.0;
    for (int i = tid; i < 3 * R3S_SLOT / 16; i += R3_THREADS)     // pixels outside the image stay zero for the whole launch
        reinterpret_cast<float4*>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // this wave's weights, cut: step s of tap `tap` = k 16 s + 8 hi .. + 7, column li (packed [tap][K/4][Npad][4]: float4 rows 4 s + 2 hi, + 1);
    // the ninth tap is shared: wave w contracts its k-slice [16 w, 16 w + 16)
    u32x4 bh[9], bm[9], bl[9];
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
        const int row = (s9 < 8 ? tap * (K / 4) + 4 * s9 : 8 * (K / 4) + 4 * tap) + 2 * hi;
        const float4 w0 = ldg4(p.wB + ((size_t)row * p.Npad + li) * 4);
        const float4 w1 = ldg4(p.wB + ((size_t)(row + 1) * p.Npad + li) * 4);
        const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        split_bf16x3(f, bh[s9], bm[s9], bl[s9]);
    }
    __syncthreads();

    const int NH = p.M / W;                                       // image rows in the batch
    const int g_begin = (blockIdx.x / nstrip) * rows_per_wg;
    int g_end = g_begin + rows_per_wg;
    if (g_end > NH) g_end = NH;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;

    // staging plan: a ring row is 34 pixels x 32 float4; thread t takes float4 t, t + 512, t + 1024: its channel group never changes
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
    constexpr int NX4 = 34 * 32;
    const int c4 = (tid & 31) << 2;
    const float4 s4 = *reinterpret_cast<const float4*>(sc + c4);
    const float4 h4 = *reinterpret_cast<const float4*>(sh + c4);
    // (a row is requested a whole iteration before it is cut into the ring: two register sets)
    f32x4n xv[3], xvn[3];
    bool xok = false, xokn = false;
    auto issue_x = [&](int g, f32x4n (&v)[3], bool& ok) {
        ok = g >= 0 && g < NH;
        const size_t base = (size_t)(ok ? g : 0) * W;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int idx = tid + R3_THREADS * j;
            if (idx >= NX4) idx = NX4 - 1;
            int x = x0 - 1 + (idx >> 5);
            x = x < 0 ? 0 : (x >= W ? W - 1 : x);                 // (clamped: the load stays unconditional, commit_x skips the pixel)
            v[j] = *reinterpret_cast<const f32x4n*>(sg.x + (base + x) * sg.ld + c4);
        }
    };
    auto commit_x = [&](int g) {
        if (!xok) return;
        char* slot = ring + (size_t)(((g % 3) + 3) % 3) * R3S_SLOT;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + R3_THREADS * j;
            if (idx >= NX4) break;
            const int x = x0 - 1 + (idx >> 5);
            if (x < 0 || x >= W) continue;
            f32x4n v;
            v[0] = fmaxf(fmaf(xv[j][0], s4.x, h4.x), 0.f);
            v[1] = fmaxf(fmaf(xv[j][1], s4.y, h4.y), 0.f);
            v[2] = fmaxf(fmaf(xv[j][2], s4.z, h4.z), 0.f);
            v[3] = fmaxf(fmaf(xv[j][3], s4.w, h4.w), 0.f);
            if (p.qin_bits) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quan_input_act(v[e], p.qin_bits);
            }
            u32x2n ph, pm, pl;
            unsigned a0, a1, a2;
            split_bf16x3_pair(f32x2_op{v[0], v[1]}, a0, a1, a2);
            ph[0] = a0; pm[0] = a1; pl[0] = a2;
            split_bf16x3_pair(f32x2_op{v[2], v[3]}, a0, a1, a2);
            ph[1] = a0; pm[1] = a1; pl[1] = a2;
            char* d = slot + (idx >> 5) * R3S_PIX + c4 * 2;
            *reinterpret_cast<u32x2n*>(d) = ph;
            *reinterpret_cast<u32x2n*>(d + 256) = pm;
            *reinterpret_cast<u32x2n*>(d + 512) = pl;
        }
    };

    for (int g = g_begin - 1; g <= g_begin + 1; ++g) { issue_x(g, xv, xok); commit_x(g); }
    issue_x(g_begin + 2, xvn, xokn);
    __syncthreads();

    double dsum = 0.0, dsq = 0.0;
    // tuning builds, CUNET_CONV_DBG & 8192: phase clocks as in dgrad3x3_ring_split_kernel -- [1] requests, [2] MFMAs, [3] partial tiles to
    // LDS + barrier, [4] sum of the eight partial tiles + store + statistics, [5] barrier, [6] ring commit + barrier
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const bool stamp = CUNET_DBG(p, 8192) != 0;
    auto now = [&]() -> unsigned long long { return stamp ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long tk0 = now();
    for (int g = g_begin; g < g_end; ++g) {
        const int y = g % H;
        const unsigned long long t0 = now();
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[j] = xvn[j];               // row g + 2, requested one iteration ago
        xok = xokn;
        issue_x(g + 3, xvn, xokn);                                // in flight across this whole iteration
        const bool rvalid = (y + dy >= 0) && (y + dy < H);
        const bool xvalid = y + 1 < H;                            // tap 8 = (dy, dx) = (+1, +1)
        const int sl = (g + dy + 3) % 3, slx = (g + 1) % 3;
        const unsigned long long t1 = now();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (rvalid) {
            const char* ap = ring + (size_t)sl * R3S_SLOT + (li + dx + 1) * R3S_PIX + 16 * hi;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ap + 32 * s8);
                const u32x4 am = *reinterpret_cast<const u32x4*>(ap + 32 * s8 + 256);
                const u32x4 al = *reinterpret_cast<const u32x4*>(ap + 32 * s8 + 512);
                acc = mfma_split6(ah, am, al, bh[s8], bm[s8], bl[s8], acc);
            }
        }
        if (xvalid) {
            const char* ap = ring + (size_t)slx * R3S_SLOT + (li + 2) * R3S_PIX + 32 * tap + 16 * hi;
            const u32x4 ah = *reinterpret_cast<const u32x4*>(ap);
            const u32x4 am = *reinterpret_cast<const u32x4*>(ap + 256);
            const u32x4 al = *reinterpret_cast<const u32x4*>(ap + 512);
            acc = mfma_split6(ah, am, al, bh[8], bm[8], bl[8], acc);
        }
        if (stamp) asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[15]));      // (the chain has landed before the stamp)
        const unsigned long long t2 = now();
#pragma unroll
        for (int r = 0; r < 16; ++r) part[tap * 1024 + r * 64 + lane] = acc[r];
        __syncthreads();
        const unsigned long long t3 = now();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + R3_THREADS * u;
            float vsum = part[e];
#pragma unroll
            for (int w = 1; w < 8; ++w) vsum += part[w * 1024 + e];
            const int r = e >> 6, l = e & 63;                     // C layout: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            p.y[((size_t)g * W + x0 + row) * p.ldy + (l & 31)] = vsum;
            dsum += (double)vsum;
            dsq += (double)vsum * (double)vsum;
        }
        const unsigned long long t4 = now();
        __syncthreads();                                          // everyone is done with row g-1's ring row and with `part`
        const unsigned long long t5 = now();
        commit_x(g + 2);
        __syncthreads();
        const unsigned long long t6 = now();
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += t6 - t5;
    }
#ifdef CUNET_TUNING
    if (stamp && lane == 0) {
        const unsigned long long tk1 = now();      // (stamped BEFORE this wave queues its atomics: ~10 k same-address atomics per launch back up the memory path, which round 5's "whole kernel" column included)
        atomicAdd(&g_conv_phase[0], (unsigned long long)(g_end > g_begin ? g_end - g_begin : 0));
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_phase[1 + i], ph[i]);
        atomicAdd(&g_conv_phase[7], tk1 - tk0);
    }
#endif
    (void)tk0; (void)ph;
    if (p.ystats != nullptr) {                                    // a thread's column is tid & 31 in both passes (512 = 16 * 32)
        atomicAdd(&redbuf[(tid & 31) * 2 + 0], dsum);
        atomicAdd(&redbuf[(tid & 31) * 2 + 1], dsq);
        __syncthreads();
        if (tid < 32 && tid < p.Nout) {
            atomic_add_f64(p.ystats + tid, redbuf[tid * 2 + 0]);
            atomic_add_f64(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1]);
        }
    }
}

static bool conv3x3_ring_supported(const ConvArgs& a) {
    return a.nseg == 1 && a.K == 128 && a.Kpad == 128 && a.Nout == 32 && a.Npad == 32 && !a.seg[0].ups && a.seg[0].ld % 4 == 0 &&
           a.seg[0].C == 128 && (a.W == 64 || a.W == 32) && a.M % a.W == 0 && a.ldy >= 32;
}

static hipError_t launch_conv3x3_ring(const ConvArgs& a, int num_cus, hipStream_t s) {
    const int NH = a.M / a.W;
    if (a.split) {                      // 32-pixel strips: (W / 32) strips x row blocks
        ConvArgs ad = a;
        static const int dbg = tune_int("CUNET_CONV_DBG", 0);      // tuning builds only: phase clocks (8192)
        ad.dbg = dbg;
        const int nstrip = a.W / 32;
        int rows = (NH * nstrip + num_cus - 1) / num_cus;
        if (rows < 2) rows = 2;
        const int grid = ((NH + rows - 1) / rows) * nstrip;
        const size_t smem = (size_t)2 * 128 * 4 + 64 * 8 + (size_t)8 * 1024 * 4 + (size_t)3 * R3S_SLOT;
        static bool attr_split = false;
        if (!attr_split) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_split = true;
        }
        hipLaunchKernelGGL(conv3x3_ring_split_kernel, dim3(grid), dim3(R3_THREADS), smem, s, ad, rows);
        return hipGetLastError();
    }
    int rows = (NH + num_cus - 1) / num_cus;
    if (rows < 2) rows = 2;
    const int grid = (NH + rows - 1) / rows;
    const size_t smem = (size_t)2 * 128 * 4 + 64 * 8 + (size_t)8 * 1024 * 4 + (size_t)3 * (a.W + 2) * R3_PITCH * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(conv3x3_ring_kernel, dim3(grid), dim3(R3_THREADS), smem, s, a, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 1x1 forward at the low levels (16x16 and below at batch 24: at most ~3 blocks per CU), split-K.  The weight-stationary
// kernel gives each wave a whole tile: 10 dependent load -> MFMA chunks for a 320-channel concat behind a 40 KB operand copy,
// 12-13 us per launch whatever the size (these launches are on the forward's critical path: nothing runs beside them).  Here a
// 4-wave block owns ONE 32 x 32 output tile: wave w takes chunks w, w+4, w+8 of K, requests all of its A pieces AND its B
// fragments (straight from L2, full lines: no operand copy) before the block builds the BatchNorm tables, contracts them in
// <= 48 MFMAs, and the four partial tiles meet in LDS, where wave 0 adds them, stores the tile and the output statistics.
constexpr int SK_MAXCH = 3;             // chunks per wave: K <= 384

__device__ __forceinline__ void conv1x1_splitk_body(const ConvArgs& p, const int bidx, const int bidy) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);                 // [3][1024]
    float* sc = part + 3 * 1024;                                  // [Ccat]
    float* sh = sc + p.Ccat;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    int bx = bidx, by = bidy;
    if (p.xcd_gx > 0) {                                           // column slices of a row tile on one XCD (see conv_kernel)
        const int L = bidx, xcd = L & 7, slot = L >> 3;
        by = slot % p.xcd_gy;
        bx = (slot / p.xcd_gy) * 8 + xcd;
        if (bx >= p.xcd_gx) return;
    }
    const int n0 = by * 32;
    const int HW = p.H * p.W;
    const int nchunks = p.K >> 5;
    const int m = bx * 32 + li;                                   // this lane's A row (M % 32 == 0)
    int rowU;
    {
        int nimg, py, px;
        if (p.wshift >= 0) { nimg = m >> p.hwshift; const int rem = m & (HW - 1); py = rem >> p.wshift; px = rem & (p.W - 1); }
        else { nimg = m / HW; const int rem = m - nimg * HW; py = rem / p.W; px = rem - py * p.W; }
        rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
    }
    // ---- all requests of this wave first
    float4 a[SK_MAXCH][4], b[SK_MAXCH][4];
#pragma unroll
    for (int u = 0; u < SK_MAXCH; ++u) {
        const int ch = wave + 4 * u;
        if (ch < nchunks) {                                       // wave-uniform
            int c = ch, s = 0;
            while (c >= (p.seg[s].C >> 5)) { c -= p.seg[s].C >> 5; ++s; }
            const Seg sg = p.seg[s];
            const float* src = sg.x + (size_t)(sg.ups ? rowU : m) * sg.ld + c * 32 + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) a[u][q] = ldg4(src + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) b[u][q] = ldg4(p.wB + ((size_t)(ch * 8 + 2 * q + hi) * p.Npad + n0 + li) * 4);
        }
    }
    // ---- BatchNorm scale / shift of the concat (every wave needs only its chunks, the block builds all: K <= 384 channels)
    for (int s = 0; s < p.nseg; ++s) {
        const Seg sg = p.seg[s];
        for (int lc = tid; lc < sg.C; lc += 256) {
            const int c = sg.choff + lc;
            double mean, istd;
            if (p.training) {
                mean = sg.stats[lc] / sg.count;
                double var = sg.stats[sg.C + lc] / sg.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                istd = 1.0 / sqrt(var + (double)BN_EPS);
            } else {
                mean = (double)p.rmean[c];
                istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
            }
            const double scale = (double)p.gamma[c] * istd;
            sc[c] = (float)scale;
            sh[c] = (float)((double)p.beta[c] - mean * scale);
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < SK_MAXCH; ++u) {
        const int ch = wave + 4 * u;
        if (ch < nchunks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s4 = *reinterpret_cast<const float4*>(sc + ch * 32 + 8 * q + 4 * hi);
                const float4 h4 = *reinterpret_cast<const float4*>(sh + ch * 32 + 8 * q + 4 * hi);
                float4 t = a[u][q];
                t.x = fmaxf(fmaf(t.x, s4.x, h4.x), 0.f);
                t.y = fmaxf(fmaf(t.y, s4.y, h4.y), 0.f);
                t.z = fmaxf(fmaf(t.z, s4.z, h4.z), 0.f);
                t.w = fmaxf(fmaf(t.w, s4.w, h4.w), 0.f);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.x, b[u][q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.y, b[u][q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.z, b[u][q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.w, b[u][q].w, acc, 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wave - 1) * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = n0 + li;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = ((acc[r] + part[r * 64 + lane]) + part[1024 + r * 64 + lane]) + part[2048 + r * 64 + lane];
            const int mm = bx * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;   // C layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            if (col < p.Nout) {
                p.y[(size_t)mm * p.ldy + col] = v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        if (p.ystats != nullptr) {
            double d1 = (double)s1, d2 = (double)s2;
            d1 += shfl_xor_d(d1, 32);
            d2 += shfl_xor_d(d2, 32);
            if (hi == 0 && col < p.Nout) {
                atomic_add_f64(p.ystats + col, d1);
                atomic_add_f64(p.ystats + p.Nout + col, d2);
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv1x1_splitk_kernel(const ConvArgs p) { conv1x1_splitk_body(p, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void conv1x1_splitk_pair_kernel(const ConvPair q) { conv1x1_splitk_body(q.a[blockIdx.z], blockIdx.x, blockIdx.y); }

static bool conv1x1_splitk_supported(const ConvArgs& a, int num_cus) {
    if (a.taps != 1 || a.K % 32 || a.K != a.Kpad || a.K < 128 || a.K > 32 * 4 * SK_MAXCH || a.M % 32 || a.qin_bits || a.xbf16 || a.mse_tgt) return false;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 32 || a.seg[i].ld % 4) return false;
    const long blocks = (long)(a.M / 32) * ((a.Nout + 31) / 32);
    return blocks <= 3L * num_cus;                                // everything resident at once: the launch is one round of blocks
}

static hipError_t launch_conv1x1_splitk(const ConvArgs& a_in, hipStream_t s, const ConvArgs* b_in = nullptr) {
    ConvArgs a = a_in;
    set_geometry_shifts(a);
    const int gx = a.M / 32, gy = (a.Nout + 31) / 32;
    dim3 grid(gx, gy);
    a.xcd_gx = a.xcd_gy = 0;
    if (gy > 1) { a.xcd_gx = gx; a.xcd_gy = gy; grid = dim3(8 * ((gx + 7) / 8) * gy, 1); }
    const size_t smem = (size_t)3 * 1024 * 4 + (size_t)a.Ccat * 8;
    if (b_in) {
        ConvPair q;
        q.a[0] = a;
        q.a[1] = *b_in;
        copy_launch_geometry(q.a[1], a);
        grid.z = 2;
        hipLaunchKernelGGL(conv1x1_splitk_pair_kernel, grid, dim3(256), smem, s, q);
    } else {
        hipLaunchKernelGGL(conv1x1_splitk_kernel, grid, dim3(256), smem, s, a);
    }
    return hipGetLastError();
}

static size_t conv_smem_bytes(int NT, int taps, int Kpad, int Ccat, bool split = false) {
    size_t b = (size_t)taps * (Kpad / 4) * NT * 32 * 16;   // resident B operand
    if (split) b += b / 2;                                 // (three bf16 planes)
    b += (size_t)(Ccat / 4) * sizeof(GrpEnt);              // group table
    b += (size_t)Ccat * 4 * 4;                             // sc, sh, mu, is
    b += (size_t)NT * 32 * 2 * 8;                          // reduction scratch
    return b;
}
constexpr size_t CONV_TEPI_TILE = 32 * 36 * 4;             // per wave: the data gradient's epilogue tile (TEPI instantiations)

constexpr size_t CONV_LDS_BUDGET = 160 * 1024;

// pairs exist for the shapes the adapters take: the nothing-ragged 1x1 forward and its fp32 data gradient
template <int LD, int EP, int NT, bool FAST, int XB>
static hipError_t launch_pair_inst(const ConvArgs& a, const ConvArgs& b, dim3 grid, int threads, size_t smem, hipStream_t s) {
    if constexpr (FAST && ((LD == LD_SEG && EP == EP_FWD) || (LD == LD_PLAIN && EP == EP_BWD && (XB == 0 || XB == 4 || XB == 5 || XB == 6)))) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<LD, EP, NT, FAST, XB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_LDS_BUDGET);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        ConvPair q;
        q.a[0] = a;
        q.a[1] = b;
        copy_launch_geometry(q.a[1], a);
        grid.z = 2;
        hipLaunchKernelGGL((conv_pair_kernel<LD, EP, NT, FAST, XB>), grid, dim3(threads), smem, s, q);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int LD, int EP, int NT, bool FAST, int XB = 0>      // XB: 0 / 1 / 2 as ConvArgs::xbf16
static hipError_t launch_inst(const ConvArgs& a, dim3 grid, int threads, size_t smem, hipStream_t s, const ConvArgs* b = nullptr) {
    if constexpr (EP == EP_FWD && (XB == 0 || XB == 6) && LD == LD_SEG) {
        if (a.mse_tgt != nullptr) {        // a head with the loss fused in: the instantiation that carries the MSE epilogue
            if (b) return hipErrorInvalidValue;
            // (never four channel tiles: 16 target values per tile next to 64 accumulators spilled 10 - 13 registers; the launcher
            // caps a fused-loss head at three)
            if constexpr (NT <= 3) return launch_inst<LD, EP, NT, FAST, (XB == 6 ? 7 : 3)>(a, grid, threads, smem, s, nullptr);
            else return hipErrorInvalidValue;
        }
    } else if constexpr (XB != 3 && XB != 7) {
        if (a.mse_tgt != nullptr) return hipErrorInvalidValue;      // (the fused loss exists for the 1x1 forward only)
    }
    if constexpr (XB != 3 && XB != 7) {
        if (b) return launch_pair_inst<LD, EP, NT, FAST, XB>(a, *b, grid, threads, smem, s);
    }
    static bool attr_done = false;
    if (!attr_done) {       // dynamic LDS above 64 KB has to be opted into, once per instantiation
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<LD, EP, NT, FAST, XB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_LDS_BUDGET);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_kernel<LD, EP, NT, FAST, XB>), grid, dim3(threads), smem, s, a);
    return hipGetLastError();
}

template <int LD, int EP>
static hipError_t launch_nt(const ConvArgs& a, int NT, bool fast, dim3 grid, int threads, size_t smem, hipStream_t s, const ConvArgs* b = nullptr) {
    if constexpr (LD == LD_PLAIN && EP == EP_BWD) {
        if (a.wg_part != nullptr) {            // data gradient + weight gradient in one pass (conv_body's fused tile loop)
            if (NT != 1 || !fast || a.xbf16 || (b && b->wg_part == nullptr)) return hipErrorInvalidValue;
            return launch_inst<LD, EP, 1, true, 4>(a, grid, threads, smem, s, b);
        }
        // one channel tile per wave over K = 128 (every bottleneck / adapter): two chunks of dY in flight (planner option dgrad_prefetch)
        if (NT == 1 && fast && !a.xbf16 && a.dgrad_prefetch >= 2 && a.taps == 1 && a.K == 128 && a.Kpad == 128 && a.Nout % 32 == 0)
            return launch_inst<LD, EP, 1, true, 5>(a, grid, threads, smem, s, b);
    } else {
        if (a.wg_part != nullptr) return hipErrorInvalidValue;
    }
    if (EP == EP_BWD && a.xbf16) {            // bf16 activations (and gradients): the one-tile variants only
        if (NT != 1) return hipErrorInvalidValue;
        constexpr int B = (EP == EP_BWD) ? 1 : 0;
        if (a.xbf16 == 2) {
            if (fast) return launch_inst<LD, EP, 1, (LD != LD_STEM), 2 * B>(a, grid, threads, smem, s, b);
            return launch_inst<LD, EP, 1, false, 2 * B>(a, grid, threads, smem, s, b);
        }
        if (fast) return launch_inst<LD, EP, 1, (LD != LD_STEM), B>(a, grid, threads, smem, s, b);
        return launch_inst<LD, EP, 1, false, B>(a, grid, threads, smem, s, b);      // heads: K = class_num
    }
    if constexpr (LD != LD_STEM) {
        if (fast && a.split) {             // the fp32 contraction on the bf16 matrix pipe (planner option f32_split)
            switch (NT) {
                case 1: return launch_inst<LD, EP, 1, true, 6>(a, grid, threads, smem, s, b);
                case 2: return launch_inst<LD, EP, 2, true, 6>(a, grid, threads, smem, s, b);
                case 3: return launch_inst<LD, EP, 3, true, 6>(a, grid, threads, smem, s, b);
                default: return launch_inst<LD, EP, 4, true, 6>(a, grid, threads, smem, s, b);
            }
        }
    }
    if (fast && LD != LD_STEM) {
        switch (NT) {
            case 1: return launch_inst<LD, EP, 1, (LD != LD_STEM)>(a, grid, threads, smem, s, b);
            case 2: return launch_inst<LD, EP, 2, (LD != LD_STEM)>(a, grid, threads, smem, s, b);
            case 3: return launch_inst<LD, EP, 3, (LD != LD_STEM)>(a, grid, threads, smem, s, b);
            default: return launch_inst<LD, EP, 4, (LD != LD_STEM)>(a, grid, threads, smem, s, b);
        }
    }
    if constexpr (LD == LD_PLAIN && EP == EP_BWD) {
        if (a.split && NT == 1) return launch_inst<LD, EP, 1, false, 6>(a, grid, threads, smem, s, b);      // heads: K = class_num
    }
    switch (NT) {
        case 1: return launch_inst<LD, EP, 1, false>(a, grid, threads, smem, s, b);
        case 2: return launch_inst<LD, EP, 2, false>(a, grid, threads, smem, s, b);
        case 3: return launch_inst<LD, EP, 3, false>(a, grid, threads, smem, s, b);
        default: return launch_inst<LD, EP, 4, false>(a, grid, threads, smem, s, b);
    }
}

// Launch geometry of the fp32 data gradient with the LDS-tile epilogue (TEPI instantiations) when a wave owns up to `c` channel tiles.
// `est`: modelled matrix-pipe time of the slowest SIMD in tile-times (row tiles per wave x channel tiles per wave x waves per SIMD):
// with only 32-row tiles to deal, a choice that leaves the waves 2.5 tiles each runs as long as one that leaves them 3.
struct TepiGeom {
    bool ok = false;
    int NT = 1, gy = 1, bpc = 1, waves = 4, maxw = CUNET_TEPI_WAVES, rem = 0, gbg = 0, gsm = 0, gx = 1;
    size_t smem = 0;
    long est = 0;
};
static TepiGeom tepi_geometry(const ConvArgs& a, int c, int ntiles, int ncol32, long target, int num_cus) {
    TepiGeom g;
    const bool fused = a.wg_part != nullptr;                          // the weight gradient fused in: one channel tile per wave (XBG = 4)
    if (fused) c = 1;
    const int slices = (ncol32 + c - 1) / c;
    g.NT = (ncol32 + slices - 1) / slices;
    if (g.NT > 1 && (long)ntiles * slices < target) return g;        // too few wave-tiles to fill the chip
    g.gy = (ncol32 + g.NT - 1) / g.NT;
    g.maxw = fused ? 8 : (g.NT <= 2 ? CUNET_TEPI_WAVES : 8);          // (conv_max_waves of those instantiations)
    const size_t base = conv_smem_bytes(g.NT, a.taps, a.Kpad, a.Ccat, a.split != 0);
    for (g.bpc = 3; g.bpc >= 1; --g.bpc) {
        const int wmax = g.maxw / g.bpc < 4 ? 4 : g.maxw / g.bpc;
        g.smem = base + (size_t)wmax * CONV_TEPI_TILE * (fused ? 2 : 1);      // + one epilogue tile per wave (fused: + one transpose tile)
        if (g.smem <= (g.bpc == 3 ? 52 * 1024 : (g.bpc == 2 ? 80 * 1024 : CONV_LDS_BUDGET))) break;
    }
    if (g.bpc < 1) {
        // one block per CU with fewer waves (a 3x3 operand on the split contraction with two channel tiles: 110 KB of planes leave room for
        // eight epilogue tiles, not twelve)
        for (int w : {8, 6, 4}) {
            if (w >= g.maxw) continue;
            g.smem = base + (size_t)w * CONV_TEPI_TILE * (fused ? 2 : 1);
            if (g.smem <= CONV_LDS_BUDGET) { g.bpc = 1; g.maxw = w; break; }
        }
        if (g.bpc < 1) return g;
    }
    const int max_blocks_x = (g.bpc * num_cus + g.gy - 1) / g.gy;
    g.waves = (ntiles + max_blocks_x - 1) / max_blocks_x;
    if (g.waves > g.maxw / g.bpc) g.waves = g.maxw / g.bpc;
    if (g.waves < 1) g.waves = 1;
    g.gx = (ntiles + g.waves - 1) / g.waves;
    if (g.gx > max_blocks_x) g.gx = max_blocks_x;
    if (g.gx < 1) g.gx = 1;
    const int wpb = g.waves < 4 ? 4 : g.waves;                        // waves a block is launched with
    const int per_simd = (g.bpc * wpb + 3) / 4;
    auto tiles_per_wave = [&](int blocks) { const long w = (long)blocks * g.waves; return (int)((ntiles + w - 1) / w); };
    if (g.NT == 1) {
        g.est = (long)tiles_per_wave(g.gx) * per_simd;
    } else {
        // slices of NT and (when the tile count does not divide) NT - 1 tiles; row blocks in proportion to a slice's tiles, out of
        // what the chip holds at once, so that every block does about the same work: 5 tiles = 3 + 2 on 512 blocks -> 307 + 204
        const int base_t = ncol32 / g.gy;
        g.rem = ncol32 - base_t * g.gy;
        const long cap_blocks = (long)g.bpc * num_cus;
        const int need = (ntiles + g.waves - 1) / g.waves;            // row blocks that give every wave at least one tile
        g.gsm = (int)(cap_blocks * base_t / ncol32);
        g.gbg = g.rem ? (int)(cap_blocks * (base_t + 1) / ncol32) : 0;
        if (g.gsm > need) g.gsm = need;
        if (g.gbg > need) g.gbg = need;
        if (g.gsm < 1) g.gsm = 1;
        if (g.rem > 0 && g.gbg < 1) g.gbg = 1;
        const long ts = (long)tiles_per_wave(g.gsm) * base_t, tb = g.rem ? (long)tiles_per_wave(g.gbg) * (base_t + 1) : 0;
        g.est = (ts > tb ? ts : tb) * per_simd;
    }
    g.ok = true;
    return g;
}

// Row blocks (= partial weight-gradient tiles) a fused data + weight gradient launch of `a` will use on `num_cus` CUs (`pair`: launched with
// its partner on half of the chip each); 0 when the shape has no fused kernel.  The runtime sizes the bucket's reduce by it.
int conv_fused_wgrad_splits(const ConvArgs& a_in, bool pair, int num_cus_all) {
    ConvArgs a = a_in;
    if (a.wg_part == nullptr) a.wg_part = reinterpret_cast<float*>(sizeof(float));      // (geometry only: any non-null value)
    const int num_cus = pair ? num_cus_all / 2 : num_cus_all;
    bool fast = (a.K % 32 == 0) && (a.M % 32 == 0) && (a.K == a.Kpad);
    if (!(fast && a.xbf16 == 0 && a.taps == 1 && a.K == 128 && a.Nout % 32 == 0 && a.ldy == a.Nout)) return 0;
    const TepiGeom g = tepi_geometry(a, 1, (a.M + 31) / 32, (a.Nout + 31) / 32, 2L * 4 * num_cus, num_cus);
    return g.ok ? g.gx : 0;
}

// Host launcher.  Picks the channel tile NT (all output channels per block when the node is big
// and its weights fit the LDS, fewer when there are too few 32-row tiles to fill the chip), the
// waves per block and the grid.
static hipError_t launch_conv_impl(const ConvArgs& a_in, const ConvArgs* b_in, int load, int epi, int num_cus_all, hipStream_t s);

hipError_t launch_conv(const ConvArgs& a_in, int load, int epi, int num_cus, hipStream_t s) { return launch_conv_impl(a_in, nullptr, load, epi, num_cus, s); }

// a_in and b_in in one launch (see conv_pair_kernel); hipErrorNotSupported (and nothing launched): the caller launches them one by one
hipError_t launch_conv_pair(const ConvArgs& a_in, const ConvArgs& b_in, int load, int epi, int num_cus, hipStream_t s) {
    if (!conv_pairable(a_in, b_in) || a_in.taps != 1) return hipErrorNotSupported;
    // (the row-tile data gradient gives every node the whole chip: no pair launch)
    if (load == LD_PLAIN && epi == EP_BWD && a_in.dgrad_rows > 0 && dgrad1x1_rows_supported(a_in) && a_in.M / 32 >= a_in.dgrad_rows) return hipErrorNotSupported;
    return launch_conv_impl(a_in, &b_in, load, epi, num_cus, s);
}

static hipError_t launch_conv_impl(const ConvArgs& a_in, const ConvArgs* b_in, int load, int epi, int num_cus_all, hipStream_t s) {
    const int num_cus = b_in ? num_cus_all / 2 : num_cus_all;      // a pair: each problem on half of the chip
    static const int ring_min_w = tune_int("CUNET_CONV_RING_MINW", 32);      // 3x3 forward on the LDS row ring at this width and above (64 and 32: +0.4 % over 64 only)
    if (!b_in && load == LD_3X3 && epi == EP_FWD && conv3x3_ring_supported(a_in) && a_in.W >= ring_min_w && a_in.M / a_in.W >= (a_in.ring_min_rows > 0 ? a_in.ring_min_rows : 512))
        return launch_conv3x3_ring(a_in, num_cus, s);
    // the stem on the split contraction: output rows over an LDS ring of cut input rows (planner option stem_split)
    if (!b_in && load == LD_STEM && epi == EP_FWD && a_in.split && stem_fwd_split_supported(a_in)) return launch_stem_fwd_split(a_in, num_cus, s);
    // fp32 3x3 data gradient at 64 x 64 / 32 x 32 on the split contraction: the dY row ring (planner option dgrad3_ring = least image rows)
    if (!b_in && load == LD_PLAIN3 && epi == EP_BWD && a_in.dgrad3_ring > 0 && dgrad3x3_ring_supported(a_in) && a_in.M / a_in.W >= a_in.dgrad3_ring)
        return launch_dgrad3x3_ring(a_in, num_cus, s);
    // fp32 1x1 data gradient of a 128-output-channel node with 128 ... 320 input channels: every column of a row tile in one workgroup, dY
    // staged once (dgrad1x1_rows_kernel); planner option dgrad_rows = the least number of 32-row tiles per launch that takes it
    if (!b_in && load == LD_PLAIN && epi == EP_BWD && a_in.dgrad_rows > 0 && dgrad1x1_rows_supported(a_in) && a_in.M / 32 >= a_in.dgrad_rows)
        return launch_dgrad1x1_rows(a_in, num_cus, s);
    static const int use_sk = tune_int("CUNET_CONV_SPLITK", 1);
    // (a pair: one round of one-tile blocks for both problems together)
    if (use_sk && load == LD_SEG && epi == EP_FWD && conv1x1_splitk_supported(a_in, num_cus)) return launch_conv1x1_splitk(a_in, s, b_in);
    static const int use_ts = tune_int("CUNET_CONV_TS", 1);
    if (!b_in && use_ts && load == LD_3X3 && epi == EP_FWD && a_in.nseg == 1 && a_in.K == 128 && a_in.Kpad == 128 && a_in.Nout == 32 &&
        a_in.Npad == 32 && a_in.M % 32 == 0 && a_in.seg[0].ld % 4 == 0 && !a_in.seg[0].ups &&
        a_in.qin_bits == 0 && a_in.M / 32 <= use_ts * 4 * num_cus)      // beyond ~4 tiles per CU the barrier-free kernel is ahead (93 vs 106 us at 64x64, bs 24)
        return launch_conv3x3_tapsplit(a_in, num_cus, s);
    static const int dbg = tune_int("CUNET_CONV_DBG", 0);      // tuning builds only: work-skipping timing experiments
    ConvArgs a = a_in;
    a.dbg = dbg;
    set_geometry_shifts(a);
    const int ntiles = (a.M + 31) / 32;
    const int ncol32 = (a.Nout + 31) / 32;
    static const int target_x10 = tune_int("CUNET_CONV_TARGET_X10", 20);      // (swept 7 ... 20: no change)
    const long target = (long)target_x10 * 4 * num_cus / 10;      // wave-tiles wanted: 2 per SIMD
    // channel tiles per block.  Every slice of NT tiles re-reads (and re-activates) the A operand and the
    // last slice is padded with zero tiles, so the cost of a choice is slices * (NT + overhead) tile-times:
    // a 160-channel dgrad (5 tiles) runs 2 x 3 instead of 2 x 4, a 288-channel one 3 x 3 instead of 3 x 4.
    static const float nt_ovh = tune_float("CUNET_CONV_NT_OVH", 0.3f);
    // fast path: nothing ragged (see the kernel)
    bool fast = (a.K % 32 == 0) && (a.M % 32 == 0) && (a.K == a.Kpad) && !tune_int("CUNET_CONV_GENERIC", 0);
    if (load == LD_SEG) {
        for (int i = 0; i < a.nseg; ++i) fast = fast && (a.seg[i].C % 32 == 0) && (a.seg[i].ld % 4 == 0);
    }
    // the fp32 data gradient with the LDS-tile epilogue (TEPI instantiations of the kernel)
    const bool tepi = epi == EP_BWD && fast && a.xbf16 == 0;
    // the split contraction: fp32 operands; of the ragged shapes only the heads' data gradient (the stem's im2col loader is bound by its
    // own address arithmetic: 200 -> 211 us with the split's VALU work on top); not next to the fused weight gradient / the two-chunk loop
    a.split = (a.split && a.xbf16 == 0 && a.wg_part == nullptr && load != LD_STEM && (fast || (load == LD_PLAIN && epi == EP_BWD))) ? 1 : 0;
    if (a.split) a.dgrad_prefetch = 1;
    if (a.wg_part != nullptr && !(tepi && load == LD_PLAIN && a.taps == 1 && a.K == 128 && a.Nout % 32 == 0 && a.ldy == a.Nout))
        return hipErrorInvalidValue;       // (the fused weight gradient: fast fp32 1x1 data gradient of a 128-output-channel node only)
    int NT = 1;
    TepiGeom tg;
    const bool split_in = a.split != 0;
    if (tepi) {
        // Channel tiles per wave (planner option dgrad_nt = the most allowed): more tiles per wave read dY fewer times and give every A
        // fragment more independent accumulator chains, but they are dealt in coarser units -- the candidate with the shortest modelled
        // matrix-pipe time wins, the larger NT on a tie.  dgrad_nt = 10 + k: as many as fit up to k, whatever the model says (sweeps).
        const int opt = a.dgrad_nt > 0 ? a.dgrad_nt : 4;
        const bool greedy = opt > 10;
        int cap = greedy ? opt - 10 : opt;
        cap = load == LD_PLAIN ? (cap > 4 ? 4 : (cap < 1 ? 1 : cap)) : 1;
        // (split contraction: the matrix pipe is no longer what a second tile per wave relieves, and the two- and four-tile instantiations
        // spill 10 - 27 registers: one tile per wave measured best, 3701 vs 3612 (model) / 3530 / 3573 img/s (forced 2 / 4))
        if (a.split && !greedy) cap = 1;
        // the 3x3 data gradient (planner option dgrad3_nt, forced): every column slice gathers -- and, on the split contraction, cuts -- the
        // nine shifted taps of dY again; 2 or 4 tiles per wave halve / remove that
        bool greedy3 = false;
        if (load == LD_PLAIN3 && a.dgrad3_nt > 1) { cap = a.dgrad3_nt > 4 ? 4 : a.dgrad3_nt; greedy3 = true; }
        for (int pass = 0; pass < 2 && !tg.ok; ++pass) {
            if (pass == 1) {
                if (!a.split) break;
                a.split = 0;                          // (an operand whose three planes exceed the LDS: the fp32 pipe)
            }
            for (int c = cap; c >= 1; --c) {
                const TepiGeom g = tepi_geometry(a, c, ntiles, ncol32, target, num_cus);
                if (!g.ok) continue;
                if (!tg.ok || g.est < tg.est) tg = g;
                if (greedy || greedy3) break;
            }
        }
        if (!tg.ok) return hipErrorInvalidValue;
        NT = tg.NT;
    } else if (nt_ovh < 0.f) {                                  // powers of two only (first version of this launcher)
        NT = ncol32 >= 4 ? 4 : (ncol32 >= 2 ? 2 : 1);
        while (NT > 1 && ((long)ntiles * ((ncol32 + NT - 1) / NT) < target ||
                          conv_smem_bytes(NT, a.taps, a.Kpad, a.Ccat, split_in) > CONV_LDS_BUDGET))
            NT >>= 1;
    } else {
        float best = 1e30f;
        static const int nt_max = tune_int("CUNET_CONV_NT_MAX", 4);
        static const int nt_max_bwd = tune_int("CUNET_CONV_NT_MAX_BWD", 1);
        for (int c = (a.xbf16 ? 1 : (epi == EP_BWD ? nt_max_bwd : (a.mse_tgt ? (nt_max < 3 ? nt_max : 3) : nt_max))); c >= 1; --c) {
            if (conv_smem_bytes(c, a.taps, a.Kpad, a.Ccat, split_in) > CONV_LDS_BUDGET) continue;
            const int slices = (ncol32 + c - 1) / c;
            if (c > 1 && (long)ntiles * slices < target) continue;
            const float cost = slices * ((float)c + nt_ovh);
            if (cost < best) { best = cost; NT = c; }
        }
    }
    size_t smem = conv_smem_bytes(NT, a.taps, a.Kpad, a.Ccat, a.split != 0);
    if (smem > CONV_LDS_BUDGET && a.split) {                     // (a 3x3 operand that fits as fp32 only: the fp32 pipe)
        a.split = 0;
        smem = conv_smem_bytes(NT, a.taps, a.Kpad, a.Ccat, false);
    }
    if (smem > CONV_LDS_BUDGET) return hipErrorInvalidValue;   // weights of one 32-channel slice exceed the LDS
    const int gy = (ncol32 + NT - 1) / NT;
    a.col_slices = gy;
    int blocks_per_cu = smem > 80 * 1024 ? 1 : (smem > 52 * 1024 ? 2 : 3);
    int maxw = CONV_MAX_WAVES;            // 16 waves for one-tile blocks (VGPR budget 128) measured 2-3 % slower
    if (tepi) { maxw = tg.maxw; blocks_per_cu = tg.bpc; smem = tg.smem; }
    const int max_blocks_x = (blocks_per_cu * num_cus + gy - 1) / gy;
    int waves = (ntiles + max_blocks_x - 1) / max_blocks_x;
    if (waves > maxw / blocks_per_cu) waves = maxw / blocks_per_cu;
    if (waves < 1) waves = 1;
    int gx = (ntiles + waves - 1) / waves;
    if (gx > max_blocks_x) gx = max_blocks_x;
    if (gx < 1) gx = 1;
    dim3 grid(gx, gy);
    static const int xcd_remap = tune_int("CUNET_CONV_XCD", 1);
    a.xcd_gx = a.xcd_gy = 0;
    static const int xcd_fwd = tune_int("CUNET_CONV_XCD_FWD", 1);
    if (xcd_remap && gy > 1 && (epi == EP_BWD || xcd_fwd)) {        // column slices of a row block re-read the same A rows: keep them on one XCD
        a.xcd_gx = gx; a.xcd_gy = gy;
        grid = dim3(8 * ((gx + 7) / 8) * gy, 1);
    }
    a.sl_rem = a.sl_gx_big = a.sl_gx_small = 0;
    if (tepi && NT > 1) {                 // row blocks in proportion to a slice's tiles (tepi_geometry), decoded from a 1-D grid
        a.sl_rem = tg.rem; a.sl_gx_big = tg.gbg; a.sl_gx_small = tg.gsm;
        a.xcd_gx = a.xcd_gy = 0;
        grid = dim3(8 * (tg.rem * ((tg.gbg + 7) / 8) + (gy - tg.rem) * ((tg.gsm + 7) / 8)), 1);
    }
    // never fewer than 4 waves: idle waves still help copying B into LDS and building the BN tables
    const int threads = (waves < 4 ? 4 : waves) * 64;
#define CUNET_CASE(L, E) \
    if (load == L && epi == E) return launch_nt<L, E>(a, NT, fast, grid, threads, smem, s, b_in);
    CUNET_CASE(LD_SEG, EP_FWD)
    CUNET_CASE(LD_3X3, EP_FWD)
    CUNET_CASE(LD_STEM, EP_FWD)
    CUNET_CASE(LD_PLAIN, EP_BWD)
    CUNET_CASE(LD_PLAIN3, EP_BWD)
#undef CUNET_CASE
    return hipErrorInvalidValue;
}

#ifdef CUNET_TUNING
// tuning builds: read (and clear) the phase clocks
extern "C" int cunet_tuning_conv_phase(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_phase), sizeof(g_conv_phase)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

}  // namespace cunet
