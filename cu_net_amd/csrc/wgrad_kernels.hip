// Weight gradients of the fused [concat -> BN -> ReLU] -> conv nodes:
//     dW[n][c][tap] = sum_m dY[m][n] * relu(bn(X))[m (+) tap][c]
// (the weight-gradient half of autograd for models/cu_net.py:24,43,47,197,300).
//
// GEMM view: the contraction runs over PIXELS, so both MFMA operands are read straight from
// HBM with fully coalesced 128-byte rows and no LDS staging: for v_mfma_f32_32x32x2_f32 the
// A fragment is dY[m0+half][n0 + (l&31)] and the B fragment is act(X[m0+half][c0 + (l&31)]).
// BN+ReLU of X is recomputed in registers (the reference re-runs cat->BN->ReLU under
// torch.utils.checkpoint for the same reason: nothing normalised is ever stored).
// A block = 4 waves working on the SAME output tile (one 32-wide n tile x `ctw` 32-wide c tiles,
// or x 9 taps for 3x3) over interleaved row pairs of its row chunk; partial tiles are summed
// through LDS and committed with one fp32 atomic per element per block.
#include <cstdlib>

#include "common.h"

namespace cunet {

constexpr int WG_UNROLL = 4;     // row pairs in flight per wave
#ifdef CUNET_TUNING
#define WG_COMMIT(p) ((p).ctw >= 0)      // CUNET_WG_NOCOMMIT flips the sign of ctw
#else
#define WG_COMMIT(p) true
#endif

enum WgLoad { WG_SEG = 0, WG_3X3 = 1, WG_STEM = 2 };

template <int LD, int NACC, int XBG>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);     // [4 waves][1024]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;

    // ---- job decode: blockIdx.y -> (n tile, c tile group)
    const int nct = (p.Ccat + 31) >> 5;              // c tiles in total
    const int ctw = (LD == WG_3X3) ? 1 : (p.ctw < 0 ? -p.ctw : p.ctw);
    const int ngroups = (nct + ctw - 1) / ctw;
    const int ntile = blockIdx.y / ngroups;
    const int cgrp = blockIdx.y - ntile * ngroups;
    const int n0 = ntile * 32;
    const int c0 = cgrp * ctw * 32;
    const int HW = p.H * p.W;

    // ---- per-lane column state of the B operand (channel c0 + a*32 + li)
    const float* xptr[NACC];                          // segment base (fp32 or bf16 storage, see ldx1) ...
    int xlc[NACC];                                    // ... and this lane's element offset in a row
    int xld[NACC], xups[NACC];
    float xsc[NACC], xsh[NACC];
    int skoff[NACC];                                  // stem: k -> (ci,ky,kx) packed
    bool cok[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int c = (LD == WG_3X3) ? (c0 + li) : (c0 + a * 32 + li);
        cok[a] = c < p.Ccat;
        xptr[a] = nullptr; xlc[a] = 0; xld[a] = 0; xups[a] = 0; xsc[a] = 0.f; xsh[a] = 0.f; skoff[a] = 0;
        if (LD == WG_STEM) {
            if (cok[a]) {
                const int ci = c / 49, r = c - ci * 49, ky = r / 7, kx = r - ky * 7;
                skoff[a] = (ci << 16) | (ky << 8) | kx;
            }
        } else if (cok[a] && (LD == WG_SEG || a == 0)) {
            int s = 0;
            for (int t = 1; t < p.nseg; ++t)
                if (c >= p.seg[t].choff) s = t;
            const Seg sg = p.seg[s];
            const int lc = c - sg.choff;
            const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
            const double mean = sum / sg.count;
            double var = sq / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const double scale = (double)p.gamma[c] * istd;
            xsc[a] = (float)scale;
            xsh[a] = (float)((double)p.beta[c] - mean * scale);
            xptr[a] = sg.x;
            xlc[a] = lc;
            xld[a] = sg.ld;
            xups[a] = sg.ups;
        }
    }

    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int row_begin = blockIdx.x * p.rows_per_block;
    int row_end = row_begin + p.rows_per_block;
    if (row_end > p.M) row_end = p.M;
    const bool nok = (n0 + li) < p.Cout;

    // each wave takes row pairs  row_begin + 2*(wave + 4*j).  Loads are branch-free (clamped addresses; validity
    // and BN+ReLU are applied when the value is consumed): a branch around a load serialises it behind its wait.
    const float* xbase0 = (LD != WG_STEM && cok[0]) ? xptr[0] : p.dy;
    const int xlc0 = (LD != WG_STEM && cok[0]) ? xlc[0] : 0;
    const int xld0 = (LD != WG_STEM && cok[0]) ? xld[0] : 0;
    for (int base = row_begin + 2 * wave * WG_UNROLL; base < row_end; base += 2 * 4 * WG_UNROLL) {
        float av[WG_UNROLL];
        float bv[WG_UNROLL][NACC];
        unsigned vmask[WG_UNROLL];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            const int m = base + 2 * u + hi;
            const bool mok = m < row_end;
            const int mc = mok ? m : row_begin;
            av[u] = ldx1<GB>(p.dy, (size_t)mc * p.lddy + (nok ? n0 + li : 0));
            const int nimg = mc / HW;
            const int rem = mc - nimg * HW;
            const int py = rem / p.W;
            const int px = rem - py * p.W;
            const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
            unsigned vm = 0;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (LD == WG_SEG) {
                    bv[u][a] = ldx1<XB>(cok[a] ? xptr[a] : p.dy, (size_t)(xups[a] ? rowU : mc) * (cok[a] ? xld[a] : 0) + (cok[a] ? xlc[a] : 0));
                    vm |= (unsigned)(cok[a] && mok) << a;
                } else if (LD == WG_3X3) {
                    const int dy = a / 3 - 1, dx = a - (a / 3) * 3 - 1;
                    const int yy = py + dy, xx = px + dx;
                    const bool ok = cok[0] && mok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    const int row = ok ? mc + dy * p.W + dx : mc;
                    bv[u][a] = ldx1<XB>(xbase0, (size_t)row * xld0 + xlc0);
                    vm |= (unsigned)ok << a;
                } else {  // WG_STEM
                    const int ci = skoff[a] >> 16, ky = (skoff[a] >> 8) & 255, kx = skoff[a] & 255;
                    const int iy = 2 * py - 3 + ky, ix = 2 * px - 3 + kx;
                    const bool ok = cok[a] && mok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    const int iyc = ok ? iy : 0, ixc = ok ? ix : 0;
                    bv[u][a] = ldg1(p.img + ((size_t)(nimg * 3 + ci) * p.IH + iyc) * p.IW + ixc);
                    vm |= (unsigned)ok << a;
                }
            }
            vmask[u] = vm | ((unsigned)(mok && nok) << 31);
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            const float a_ = (vmask[u] >> 31) ? av[u] : 0.f;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                float v = bv[u][a];
                if (LD == WG_SEG) v = fmaxf(fmaf(v, xsc[a], xsh[a]), 0.f);
                else if (LD == WG_3X3) v = fmaxf(fmaf(v, xsc[0], xsh[0]), 0.f);
                if (LD != WG_STEM && p.qin_bits) v = quan_input_act(v, p.qin_bits);      // the conv saw QuanInput(relu(bn(x)))
                v = ((vmask[u] >> a) & 1) ? v : 0.f;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, v, acc[a], 0, 0, 0);
            }
        }
    }

    // ---- block reduction through LDS, then one atomic per element, issued in MEMORY order so that a
    // wave's 64 atomics hit consecutive addresses (for 3x3 the torch layout [n][c][tap] interleaves the
    // nine accumulators: committing per accumulator was a stride-9 scatter and cost half the kernel).
    float* sum = red + 4 * 1024;                     // [NACC][1024] reduced tiles (3x3 only)
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[a][r];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            if (LD == WG_3X3) {
                sum[a * 1024 + e] = v;
            } else {
                const int r = e >> 6, l = e & 63;
                const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);   // MFMA C row -> output channel
                const int c = c0 + a * 32 + (l & 31);                        // MFMA C col -> input channel
                if (n < p.Cout && c < p.Ccat && WG_COMMIT(p)) atomicAdd(p.dw + (size_t)n * p.Ccat + c, v);
            }
        }
        __syncthreads();
    }
    if (LD == WG_3X3 && WG_COMMIT(p)) {
        // tile = 32 output channels x (32 input channels x 9 taps): per n, 288 consecutive floats
        for (int idx = tid; idx < 32 * 288; idx += 256) {
            const int nn = idx / 288;
            const int rem = idx - nn * 288;
            const int cc = rem / 9;
            const int tap = rem - cc * 9;
            // element (row nn, col cc) of accumulator `tap`: r*64 + l with row = (r&3)+8*(r>>2)+4*(l>>5), col = l&31
            const int hi2 = (nn >> 2) & 1;
            const int r = (nn & 3) | ((nn >> 3) << 2);
            const float v = sum[tap * 1024 + r * 64 + hi2 * 32 + cc];
            const int n = n0 + nn, c = c0 + cc;
            if (n < p.Cout && c < p.Ccat) atomicAdd(p.dw + ((size_t)n * p.Ccat + c) * 9 + tap, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 weight gradient, second generation.  One wave owns ALL output-channel tiles (NTW <= 4) x CT
// input-channel tiles and feeds them from two vector loads per pixel pair:
//     dY[m][4i..4i+3]        -> A fragments of the 4 n tiles   (n tile t = channels {4i+t})
//     X [m][c0+CT*j..+CT-1]  -> B fragments of the CT c tiles  (c tile t = channels {c0+CT*j+t})
// (an MFMA tile may own any 32 channels as long as A, B and the final scatter agree), so X is read
// once per group instead of once per n tile and a 4x4 wave issues 2 loads per 16 MFMAs.  Loads run
// PD pixel pairs ahead in registers; one wave per SIMD keeps the f32 matrix pipe busy on its own.
// blockIdx.y = channel group; groups of different width (CT = 4, 2, 1) cover Cin without padding.
constexpr int WG2_PD = 4;

struct Wg2Group { int c0; int ct; int chunk0; int nchunks; };   // per channel group
struct Wg2Args {
    WgradArgs w;
    int dbg;                 // unused (timing experiments)
    int any_ups;             // some segment is read through the nearest-upsample map
    int stem;                // X is the 7x7/2 im2col view of the NCHW image (w.img), no BatchNorm in front
    int ngroups;
    Wg2Group grp[12];
    int rows_per_chunk[12];
};

template <int NTW, int CT, bool UPS, bool STEM = false, int XBG = 0>      // XBG: 1 = x is bf16, 2 = x and dY are bf16
__device__ __forceinline__ void wg2_body(const Wg2Args& q, const Wg2Group g, int chunk, int rpc, float* lds) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    const WgradArgs& p = q.w;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;

    // ---- per-lane constants of the B operand: channels cb .. cb+CT-1 (all inside one segment)
    const int cb = g.c0 + CT * li;
    const bool cok = cb < p.Ccat;
    const float* xptr = nullptr;
    int xld = 0, xups = 0;
    float xsc[CT], xsh[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) { xsc[t] = 0.f; xsh[t] = 0.f; }
    // STEM: channel c of the im2col view is (ci, ky, kx) = (c / 49, (c % 49) / 7, c % 7); per-lane element offsets
    int soff[CT], sky[CT], skx[CT];
    bool sokc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int c = cb + t;
        sokc[t] = STEM && c < p.Ccat;
        const int cc = sokc[t] ? c : 0;
        const int ci = cc / 49, r = cc - ci * 49;
        sky[t] = r / 7; skx[t] = r - sky[t] * 7;
        soff[t] = (ci * p.IH + sky[t]) * p.IW + skx[t];
    }
    if (cok && !STEM) {
        int s = 0;
        for (int t = 1; t < p.nseg; ++t)
            if (cb >= p.seg[t].choff) s = t;
        const Seg sg = p.seg[s];
        const int lc = cb - sg.choff;
        xptr = XB ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(sg.x) + lc) : sg.x + lc;
        xld = sg.ld;
        xups = sg.ups;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const double sum = sg.stats[lc + t], sq = sg.stats[sg.C + lc + t];
            const double mean = sum / sg.count;
            double var = sq / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const double scale = (double)p.gamma[cb + t] * istd;
            xsc[t] = (float)scale;
            xsh[t] = (float)((double)p.beta[cb + t] - mean * scale);
        }
    }
    const bool nok = NTW * li < p.lddy;      // pad columns of dY are zero by construction

    f32x16 acc[NTW][CT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int row_begin = chunk * rpc;
    int row_end = row_begin + rpc;
    if (row_end > p.M) row_end = p.M;

    float av[WG2_PD][NTW], xv[WG2_PD][CT];
    bool xok[WG2_PD];
    // Branch-free: every lane always loads from a valid (clamped) address and validity is applied when
    // the value is consumed -- a branch around a load makes hipcc wait vmcnt(0) at the join.
    const float* xbase = cok ? xptr : p.dy;
    const int xldc = cok ? xld : 0;
    const int acol = nok ? NTW * li : 0;
    auto issue = [&](int m0, float (&a)[NTW], float (&x)[CT], bool& ok) {     // RAW loads of pixel pair (m0, m0+1)
        const int m = m0 + hi;
        const bool mok = m < row_end;
        const int mc = mok ? m : row_begin;
        {
            const size_t ao = (size_t)mc * p.lddy + acol;
            if (NTW == 4) { const float4 v = ldx4<GB>(p.dy, ao); a[0] = v.x; a[1 % NTW] = v.y; a[2 % NTW] = v.z; a[3 % NTW] = v.w; }
            else {
#pragma unroll
                for (int t = 0; t < NTW; ++t) a[t] = ldx1<GB>(p.dy, ao + t);
            }
        }
        if constexpr (STEM) {      // gather of the image: output pixel (py, px) reads rows 2py-3+ky, columns 2px-3+kx
            const int nimg = mc / HW;
            const int rem = mc - nimg * HW;
            const int py = rem / p.W;
            const int px = rem - py * p.W;
            const int iy0 = 2 * py - 3, ix0 = 2 * px - 3;
            const int base = (nimg * 3 * p.IH + iy0) * p.IW + ix0;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const bool in = sokc[t] && (iy0 + sky[t] >= 0) && (iy0 + sky[t] < p.IH) && (ix0 + skx[t] >= 0) && (ix0 + skx[t] < p.IW);
                const float v = ldg1(p.img + (in ? base + soff[t] : 0));     // always a valid address; zero padding by select
                x[t] = in ? v : 0.f;
            }
        } else {
            int xrow = mc;
            if constexpr (UPS) {
                const int nimg = mc / HW;
                const int rem = mc - nimg * HW;
                const int py = rem / p.W;
                const int px = rem - py * p.W;
                const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
                xrow = xups ? rowU : mc;
            }
            const size_t xo = (size_t)xrow * xldc;
            if (CT == 4) { const float4 v = ldx4<XB>(xbase, xo); x[0] = v.x; x[1] = v.y; x[2 % CT] = v.z; x[3 % CT] = v.w; }
            else if (CT == 2) { const float2 v = ldx2<XB>(xbase, xo); x[0] = v.x; x[1 % CT] = v.y; }
            else {
#pragma unroll
                for (int t = 0; t < CT; ++t) x[t] = ldx1<XB>(xbase, xo + t);
            }
        }
        ok = mok;
    };

    if constexpr (NTW == 4 && (CT == 2 || CT == 1) && !STEM) {
        // ---- counted-vmcnt pipeline.  hipcc drains vmcnt(0) at a loop back-edge, so with compiler-visible loads the
        // prefetch depth collapses to "whatever was issued in this iteration".  Here the loads are inline asm (invisible
        // to hipcc's wait bookkeeping) and every slot is awaited with an explicit, COUNTED s_waitcnt: when slot u is
        // consumed, the 2*(PD-1) loads of the younger slots may stay in flight (loads return in order).  The wait
        // statement names the slot's registers as read-write operands, which pins every consumer behind it.
        constexpr int PD = 6;
        f32x4 A4[PD];
        f32x2 A2[GB ? PD : 1];                  // bf16 dY: four values in two registers
        f32x2 X2[PD];
        float X1[PD];
        bool OK[PD];
        auto issue_asm = [&](int mm0, f32x4& a4, f32x2& a2, f32x2& x2, float& x1, bool& ok) {
            const int m = mm0 + hi;
            const bool mok = m < row_end;
            const int mc = mok ? m : row_begin;
            const float* asrc = p.dy + (size_t)mc * p.lddy + acol;
            int xrow = mc;
            if constexpr (UPS) {       // ~60 VALU instructions of index math per pixel pair: only for nodes that need it
                const int nimg = mc / HW;
                const int rem = mc - nimg * HW;
                const int py = rem / p.W;
                const int px = rem - py * p.W;
                const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
                xrow = xups ? rowU : mc;
            }
            if constexpr (GB) {
                const unsigned short* asrc16 = reinterpret_cast<const unsigned short*>(p.dy) + (size_t)mc * p.lddy + acol;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(a2) : "v"(asrc16) : "memory");
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a4) : "v"(asrc) : "memory");
            }
            if constexpr (XB) {        // bf16 x: CT = 2 -> one dword (two bf16), CT = 1 -> one ushort; unpacked at consumption (x1 carries the bits)
                const unsigned short* xsrc = reinterpret_cast<const unsigned short*>(xbase) + (size_t)xrow * xldc;
                if constexpr (CT == 2) asm volatile("global_load_dword %0, %1, off" : "=v"(x1) : "v"(xsrc) : "memory");
                else asm volatile("global_load_ushort %0, %1, off" : "=v"(x1) : "v"(xsrc) : "memory");
            } else {
                const float* xsrc = xbase + (size_t)xrow * xldc;
                if constexpr (CT == 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(x2) : "v"(xsrc) : "memory");
                else asm volatile("global_load_dword %0, %1, off" : "=v"(x1) : "v"(xsrc) : "memory");
            }
            ok = mok;
        };
        int mm = row_begin + 2 * wave;
#pragma unroll
        for (int u = 0; u < PD; ++u) issue_asm(mm + u * 8, A4[u], A2[GB ? u : 0], X2[u], X1[u], OK[u]);
        for (; mm < row_end; mm += PD * 8) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                if constexpr (GB)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(A2[GB ? u : 0]), "+v"(X1[u]) : "n"(2 * (PD - 1)) : "memory");
                else if constexpr (CT == 2 && !XB)
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(A4[u]), "+v"(X2[u]) : "n"(2 * (PD - 1)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(A4[u]), "+v"(X1[u]) : "n"(2 * (PD - 1)) : "memory");
                float a[4], x[CT];
                const bool okk = OK[u];
                if constexpr (GB) {
                    const unsigned b0 = __float_as_uint(A2[GB ? u : 0].x), b1 = __float_as_uint(A2[GB ? u : 0].y);
                    a[0] = (okk && nok) ? bf16_bits_lo(b0) : 0.f; a[1] = (okk && nok) ? bf16_bits_hi(b0) : 0.f;
                    a[2] = (okk && nok) ? bf16_bits_lo(b1) : 0.f; a[3] = (okk && nok) ? bf16_bits_hi(b1) : 0.f;
                } else {
                    a[0] = (okk && nok) ? A4[u].x : 0.f; a[1] = (okk && nok) ? A4[u].y : 0.f;
                    a[2] = (okk && nok) ? A4[u].z : 0.f; a[3] = (okk && nok) ? A4[u].w : 0.f;
                }
                if constexpr (XB) {
                    const unsigned bits = __float_as_uint(X1[u]);
                    x[0] = (okk && cok) ? fmaxf(fmaf(bf16_bits_lo(bits), xsc[0], xsh[0]), 0.f) : 0.f;
                    if constexpr (CT == 2) x[1 % CT] = (okk && cok) ? fmaxf(fmaf(bf16_bits_hi(bits), xsc[1 % CT], xsh[1 % CT]), 0.f) : 0.f;
                } else if constexpr (CT == 2) {
                    x[0] = (okk && cok) ? fmaxf(fmaf(X2[u].x, xsc[0], xsh[0]), 0.f) : 0.f;
                    x[1] = (okk && cok) ? fmaxf(fmaf(X2[u].y, xsc[1], xsh[1]), 0.f) : 0.f;
                } else {
                    x[0] = (okk && cok) ? fmaxf(fmaf(X1[u], xsc[0], xsh[0]), 0.f) : 0.f;
                }
                if (p.qin_bits) {                       // heads behind a QuanInput2d (uniform branch)
#pragma unroll
                    for (int t = 0; t < CT; ++t) x[t] = quan_input_act(x[t], p.qin_bits);
                }
                issue_asm(mm + (u + PD) * 8, A4[u], A2[GB ? u : 0], X2[u], X1[u], OK[u]);        // refill this slot PD pairs ahead
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int tb = 0; tb < CT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], x[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        // drain: the tail refills are still in flight and own their registers until they land
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            if constexpr (GB) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A2[GB ? u : 0]), "+v"(X1[u]) : : "memory");
            else if constexpr (CT == 2 && !XB) asm volatile("s_waitcnt vmcnt(0)" : "+v"(A4[u]), "+v"(X2[u]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(A4[u]), "+v"(X1[u]) : : "memory");
        }
    } else {
    // each wave takes pixel pairs  row_begin + 2*(wave + 4*k); a slot is refilled right after it
    // has been consumed, WG2_PD pairs ahead.  hipcc drains vmcnt(0) at the loop back-edge, so part of
    // the latency is hidden by the second wave on the SIMD (two blocks per CU) rather than by depth.
    const int stride = 8;
    int m0 = row_begin + 2 * wave;
#pragma unroll
    for (int u = 0; u < WG2_PD; ++u) issue(m0 + u * stride, av[u], xv[u], xok[u]);
    constexpr int WG2_UNROLL = 1;
    for (; m0 < row_end; m0 += WG2_UNROLL * WG2_PD * stride) {
#pragma unroll
        for (int uu = 0; uu < WG2_UNROLL * WG2_PD; ++uu) {
            const int u = uu % WG2_PD;
            float a[NTW], x[CT];
#pragma unroll
            for (int t = 0; t < NTW; ++t) a[t] = (xok[u] && nok) ? av[u][t] : 0.f;
            // BN + ReLU at consumption time (rows outside the chunk contribute exactly 0)
#pragma unroll
            for (int t = 0; t < CT; ++t)
                x[t] = STEM ? (xok[u] ? xv[u][t] : 0.f)          // raw image, validity already applied per element
                            : ((xok[u] && cok) ? fmaxf(fmaf(xv[u][t], xsc[t], xsh[t]), 0.f) : 0.f);
            if (!STEM && p.qin_bits) {
#pragma unroll
                for (int t = 0; t < CT; ++t) x[t] = quan_input_act(x[t], p.qin_bits);
            }
            issue(m0 + (uu + WG2_PD) * stride, av[u], xv[u], xok[u]);       // refill this slot PD pairs ahead
#pragma unroll
            for (int ta = 0; ta < NTW; ++ta)
#pragma unroll
                for (int tb = 0; tb < CT; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], x[tb], acc[ta][tb], 0, 0, 0);
        }
    }

    }   // generic (compiler-scheduled) pipeline

    // ---- reduce the 4 waves through LDS, then commit in memory order (coalesced atomics)
    float* red = lds;                    // [4][1024]
    float* sum = lds + 4 * 1024;         // [NTW*CT][1024]
#pragma unroll
    for (int ta = 0; ta < NTW; ++ta)
#pragma unroll
        for (int tb = 0; tb < CT; ++tb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[ta][tb][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256)
                sum[(ta * CT + tb) * 1024 + e] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            __syncthreads();
        }
    if (!WG_COMMIT(p)) return;           // tuning builds only: timing without the commit
    const int width = 32 * CT;
    for (int idx = tid; idx < NTW * 32 * width; idx += 256) {
        const int n = idx / width;
        const int cc = idx - n * width;
        const int ta = n % NTW, i = n / NTW;          // n = NTW*i + ta
        const int tb = cc % CT, j = cc / CT;          // c = c0 + CT*j + tb
        const int r = (i & 3) | ((i >> 3) << 2);
        const int e = r * 64 + ((i >> 2) & 1) * 32 + j;
        const int c = g.c0 + cc;
        if (n < p.Cout && c < p.Ccat) atomicAdd(p.dw + (size_t)n * p.Ccat + c, sum[(ta * CT + tb) * 1024 + e]);
    }
}

template <int NTW, int XB>
__global__ __launch_bounds__(256, 2) void wgrad2_kernel(const Wg2Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    // blockIdx.x enumerates (group, chunk) pairs: group g owns chunks [chunk0, chunk0 + nchunks)
    int gi = 0;
    for (int t = 1; t < q.ngroups; ++t)
        if ((int)blockIdx.x >= q.grp[t].chunk0) gi = t;
    const Wg2Group g = q.grp[gi];
    const int chunk = blockIdx.x - g.chunk0;
    const int rpc = q.rows_per_chunk[gi];
    if (q.any_ups) {
        if (g.ct == 4 && NTW < 4) wg2_body<NTW, (NTW < 4 ? 4 : 2), true, false, XB>(q, g, chunk, rpc, lds);
        else if (g.ct == 2) wg2_body<NTW, 2, true, false, XB>(q, g, chunk, rpc, lds);
        else wg2_body<NTW, 1, true, false, XB>(q, g, chunk, rpc, lds);
    } else {
        if (g.ct == 4 && NTW < 4) wg2_body<NTW, (NTW < 4 ? 4 : 2), false, false, XB>(q, g, chunk, rpc, lds);
        else if (g.ct == 2) wg2_body<NTW, 2, false, false, XB>(q, g, chunk, rpc, lds);
        else wg2_body<NTW, 1, false, false, XB>(q, g, chunk, rpc, lds);
    }
}

// The stem's 7x7/2 weight gradient on the same design (its own symbol so that profiles separate it):
// X is the im2col view of the NCHW image, gathered per lane; no BatchNorm in front of it.
__global__ __launch_bounds__(256, 2) void wgrad2_stem_kernel(const Wg2Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    int gi = 0;
    for (int t = 1; t < q.ngroups; ++t)
        if ((int)blockIdx.x >= q.grp[t].chunk0) gi = t;
    const Wg2Group g = q.grp[gi];
    const int chunk = blockIdx.x - g.chunk0;
    const int rpc = q.rows_per_chunk[gi];
    if (g.ct == 3) wg2_body<4, 3, false, true>(q, g, chunk, rpc, lds);       // 147 input channels = 3 + 2 tiles: dY is read twice
    else if (g.ct == 2) wg2_body<4, 2, false, true>(q, g, chunk, rpc, lds);
    else wg2_body<4, 1, false, true>(q, g, chunk, rpc, lds);
}

static hipError_t launch_wgrad2(const WgradArgs& a, int num_cus, hipStream_t s) {
    Wg2Args q{};
    q.w = a;
    static const int nocommit = tune_int("CUNET_WG_NOCOMMIT", 0);      // tuning builds only
    q.w.ctw = nocommit ? -1 : 1;
    q.dbg = 0;
    q.any_ups = 0;
    for (int i = 0; i < a.nseg; ++i) q.any_ups |= a.seg[i].ups;
    q.stem = (a.img != nullptr && a.nseg == 0) ? 1 : 0;
    // channel groups: as many CT=4 (128-channel) groups as fit, then one CT=2 and/or CT=1 remainder
    int c = 0, ng = 0, weight = 0;
    const int C32 = (a.Ccat + 31) / 32;          // 32-channel tiles
    const int ntw = a.lddy > 64 ? 4 : (a.lddy > 32 ? 2 : 1);
    const int ctmax = ntw == 4 ? 2 : 4;          // <= 8 accumulators: two waves per SIMD
    int left = C32;
    while (left > 0 && ng < 12) {
        int ct = (left >= 4 && ctmax >= 4) ? 4 : (left >= 2 ? 2 : 1);
        if (q.stem && left >= 3 && ng == 0) ct = 3;          // scalar gathers: any tile count works (12 accumulators still fit)
        q.grp[ng].c0 = c; q.grp[ng].ct = ct;
        c += 32 * ct; left -= ct; weight += ct; ++ng;
    }
    if (left > 0) return hipErrorInvalidValue;
    q.ngroups = ng;
    // blocks: ~1 per CU in total, split over the groups in proportion to their MFMA work
    // two blocks per CU for the big nodes; small nodes get fewer blocks (>= WG2_MIN_ROWS rows each): every block
    // commits a whole output tile with atomics, which is a fixed ~8-16K atomics per block whatever M is
    static const int min_rows = tune_int("CUNET_WG_MIN_ROWS", 64);
    static const int blocks_per_cu = tune_int("CUNET_WG_BPC", 2);
    int total = blocks_per_cu * num_cus;
    {
        const long cap = ((long)a.M * ng + min_rows - 1) / min_rows;
        if (total > cap) total = (int)(cap < ng ? ng : cap);
    }
    int chunk0 = 0;
    for (int g = 0; g < ng; ++g) {
        int nch = (total * q.grp[g].ct + weight - 1) / weight;
        int rpc = (a.M + nch - 1) / nch;
        rpc = (rpc + 7) / 8 * 8;                 // whole 4-wave sweeps of pixel pairs
        nch = (a.M + rpc - 1) / rpc;
        q.grp[g].chunk0 = chunk0; q.grp[g].nchunks = nch;
        q.rows_per_chunk[g] = rpc;
        chunk0 += nch;
    }
    const size_t smem = (size_t)(4 + ntw * 4) * 1024 * 4;
    const dim3 grid(chunk0);
    auto set_attr = [&](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); };
    static bool done = false;
    if (!done) {
        hipError_t e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<4, 0>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<4, 1>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<4, 2>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<2, 0>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_stem_kernel));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<1, 0>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<2, 1>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<2, 2>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<1, 1>));
        if (e == hipSuccess) e = set_attr(reinterpret_cast<const void*>(&wgrad2_kernel<1, 2>));
        if (e != hipSuccess) return e;
        done = true;
    }
    if (q.stem) hipLaunchKernelGGL(wgrad2_stem_kernel, grid, dim3(256), smem, s, q);
    else if (a.xbf16) {          // (heads with <= 64 landmarks -- MPII's 16 joints -- take the narrow variants)
#define CUNET_WG2X(N) do { if (a.xbf16 == 2) hipLaunchKernelGGL((wgrad2_kernel<N, 2>), grid, dim3(256), smem, s, q); \
                           else hipLaunchKernelGGL((wgrad2_kernel<N, 1>), grid, dim3(256), smem, s, q); } while (0)
        if (ntw == 4) CUNET_WG2X(4); else if (ntw == 2) CUNET_WG2X(2); else CUNET_WG2X(1);
#undef CUNET_WG2X
    } else if (ntw == 4) hipLaunchKernelGGL((wgrad2_kernel<4, 0>), grid, dim3(256), smem, s, q);
    else if (ntw == 2) hipLaunchKernelGGL((wgrad2_kernel<2, 0>), grid, dim3(256), smem, s, q);
    else hipLaunchKernelGGL((wgrad2_kernel<1, 0>), grid, dim3(256), smem, s, q);
    return hipGetLastError();
}

template <int LD>
static hipError_t launch_acc(const WgradArgs& a, int nacc, dim3 grid, hipStream_t s) {
    const size_t smem = (size_t)4 * 1024 * 4 + (LD == WG_3X3 ? (size_t)9 * 1024 * 4 : 0);
#define CUNET_WG(N) case N: if (a.xbf16 == 2) hipLaunchKernelGGL((wgrad_kernel<LD, N, 2>), grid, dim3(256), smem, s, a); \
                    else if (a.xbf16) hipLaunchKernelGGL((wgrad_kernel<LD, N, 1>), grid, dim3(256), smem, s, a); \
                    else hipLaunchKernelGGL((wgrad_kernel<LD, N, 0>), grid, dim3(256), smem, s, a); break;
    switch (nacc) {
        CUNET_WG(1) CUNET_WG(2) CUNET_WG(3) CUNET_WG(4) CUNET_WG(5) CUNET_WG(9)
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG
    return hipGetLastError();
}

hipError_t launch_wgrad(WgradArgs a, int load, int num_cus, hipStream_t s) {
    if (load == WG_SEG && a.Cout <= 128 && a.lddy % 4 == 0 && !tune_int("CUNET_WG_OLD", 0)) return launch_wgrad2(a, num_cus, s);
    if (load == WG_STEM && a.Cout <= 128 && a.lddy > 64 && a.lddy % 4 == 0 && !tune_int("CUNET_WG_OLD_STEM", 0)) return launch_wgrad2(a, num_cus, s);
    const int nct = (a.Ccat + 31) / 32;
    const int ntiles = (a.Cout + 31) / 32;
    int nacc, jobs;
    if (load == WG_3X3) {
        nacc = 9; a.ctw = 1; jobs = ntiles * nct;
    } else {
        // c tiles per job: <= 5 accumulators; spread evenly
        int ctw = nct;
        if (ctw > 5) { const int g = (nct + 4) / 5; ctw = (nct + g - 1) / g; }
        a.ctw = ctw; nacc = ctw; jobs = ntiles * ((nct + ctw - 1) / ctw);
    }
    // row chunks: enough blocks to fill the chip, each chunk a multiple of 8*UNROLL rows
    const int quantum = 8 * WG_UNROLL;
    static const int mult = tune_int("CUNET_WG_CHUNK_MULT", 2);
    static const int nocommit = tune_int("CUNET_WG_NOCOMMIT", 0);                                   // tuning builds only
    if (nocommit) a.ctw = -a.ctw;
    int chunks = (mult * num_cus + jobs - 1) / jobs;
    if (chunks < 1) chunks = 1;
    int rpb = (a.M + chunks - 1) / chunks;
    rpb = (rpb + quantum - 1) / quantum * quantum;
    chunks = (a.M + rpb - 1) / rpb;
    a.rows_per_block = rpb;
    const dim3 grid(chunks, jobs);
    if (load == WG_SEG) return launch_acc<WG_SEG>(a, nacc, grid, s);
    if (load == WG_3X3) return launch_acc<WG_3X3>(a, nacc, grid, s);
    if (load == WG_STEM) return launch_acc<WG_STEM>(a, nacc, grid, s);
    return hipErrorInvalidValue;
}

}  // namespace cunet
