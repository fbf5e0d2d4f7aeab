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
#include "common.h"

namespace cunet {

constexpr int WG_MAXACC = 9;     // accumulators per wave: up to 9 taps (3x3) or ctw<=9 c-tiles
constexpr int WG_UNROLL = 4;     // row pairs in flight per wave

enum WgLoad { WG_SEG = 0, WG_3X3 = 1, WG_STEM = 2 };

template <int LD, int NACC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);     // [4 waves][1024]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;

    // ---- job decode: blockIdx.y -> (n tile, c tile group)
    const int nct = (p.Ccat + 31) >> 5;              // c tiles in total
    const int ctw = (LD == WG_3X3) ? 1 : p.ctw;
    const int ngroups = (nct + ctw - 1) / ctw;
    const int ntile = blockIdx.y / ngroups;
    const int cgrp = blockIdx.y - ntile * ngroups;
    const int n0 = ntile * 32;
    const int c0 = cgrp * ctw * 32;
    const int HW = p.H * p.W;

    // ---- per-lane column state of the B operand (channel c0 + a*32 + li)
    const float* xptr[NACC];
    int xld[NACC], xups[NACC];
    float xsc[NACC], xsh[NACC];
    int skoff[NACC];                                  // stem: k -> (ci,ky,kx) packed
    bool cok[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int c = (LD == WG_3X3) ? (c0 + li) : (c0 + a * 32 + li);
        cok[a] = c < p.Ccat;
        xptr[a] = nullptr; xld[a] = 0; xups[a] = 0; xsc[a] = 0.f; xsh[a] = 0.f; skoff[a] = 0;
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
            xptr[a] = sg.x + lc;
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

    // each wave takes row pairs  row_begin + 2*(wave + 4*j)
    for (int base = row_begin + 2 * wave * WG_UNROLL; base < row_end; base += 2 * 4 * WG_UNROLL) {
        float av[WG_UNROLL];
        float bv[WG_UNROLL][NACC];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            const int m = base + 2 * u + hi;
            const bool mok = m < row_end;
            const int mc = mok ? m : row_begin;
            av[u] = (mok && nok) ? p.dy[(size_t)mc * p.lddy + n0 + li] : 0.f;
            int nimg = 0, py = 0, px = 0;
            if (LD != WG_SEG || true) {
                nimg = mc / HW;
                const int rem = mc - nimg * HW;
                py = rem / p.W;
                px = rem - py * p.W;
            }
            const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                float v = 0.f;
                if (LD == WG_SEG) {
                    if (cok[a] && mok) {
                        const float xv = xptr[a][(size_t)(xups[a] ? rowU : mc) * xld[a]];
                        v = fmaxf(fmaf(xv, xsc[a], xsh[a]), 0.f);
                    }
                } else if (LD == WG_3X3) {
                    const int dy = a / 3 - 1, dx = a - (a / 3) * 3 - 1;
                    const int yy = py + dy, xx = px + dx;
                    const bool ok = cok[0] && mok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    if (ok) {
                        const float xv = xptr[0][(size_t)(mc + dy * p.W + dx) * xld[0]];
                        v = fmaxf(fmaf(xv, xsc[0], xsh[0]), 0.f);
                    }
                } else {  // WG_STEM
                    const int ci = skoff[a] >> 16, ky = (skoff[a] >> 8) & 255, kx = skoff[a] & 255;
                    const int iy = 2 * py - 3 + ky, ix = 2 * px - 3 + kx;
                    const bool ok = cok[a] && mok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    if (ok) v = p.img[((size_t)(nimg * 3 + ci) * p.IH + iy) * p.IW + ix];
                }
                bv[u][a] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][a], acc[a], 0, 0, 0);
    }

    // ---- block reduction through LDS (one accumulator at a time: 16 KB), then one atomic per element
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[a][r];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const int r = e >> 6, l = e & 63;
            const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);   // MFMA C row -> output channel
            const int cc = l & 31;                                         // MFMA C col -> input channel
            int c, tap;
            if (LD == WG_3X3) { c = c0 + cc; tap = a; } else { c = c0 + a * 32 + cc; tap = 0; }
            if (n < p.Cout && c < p.Ccat)
                atomicAdd(p.dw + ((size_t)n * p.Ccat + c) * p.taps + tap, v);
        }
        __syncthreads();
    }
}

template <int LD>
static hipError_t launch_acc(const WgradArgs& a, int nacc, dim3 grid, hipStream_t s) {
    const size_t smem = (size_t)4 * 1024 * 4;
#define CUNET_WG(N) case N: hipLaunchKernelGGL((wgrad_kernel<LD, N>), grid, dim3(256), smem, s, a); break;
    switch (nacc) {
        CUNET_WG(1) CUNET_WG(2) CUNET_WG(3) CUNET_WG(4) CUNET_WG(5) CUNET_WG(9)
        default: return hipErrorInvalidValue;
    }
#undef CUNET_WG
    return hipGetLastError();
}

hipError_t launch_wgrad(WgradArgs a, int load, int num_cus, hipStream_t s) {
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
    int chunks = (2 * num_cus + jobs - 1) / jobs;
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
