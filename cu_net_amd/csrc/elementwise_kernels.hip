// HBM-bound kernels of the CU-Net hot path: BatchNorm backward apply, max-pool / nearest-upsample
// index maps, the stem's BN-ReLU-pool, layout transposes at the NCHW boundary, pixelwise MSE,
// running-statistics update, weight repacking, fused RMSprop and the arg-max landmark decode.
// All are vectorised (16 B per lane) streaming kernels; per-channel reductions are carried in
// fp64 and committed with one atomic per channel per block.
#include "common.h"
#include <algorithm>

namespace cunet {

__device__ __forceinline__ void atomic_add_f64e(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward, second half (first half = EP_BWD epilogue of conv_kernel), gathered per TENSOR:
// every conv node c that reads tensor X through its own BatchNorm contributes
//   scale_c * (dz_c - mean(dz_c) - xhat * mean(dz_c * xhat))  =  A_c*dz_c + E_c - D_c*x
// with per-channel A, E, D.  The gradient of X is assembled here in one pass: x is read once, each
// consumer's dz slice once, dX written once (the first version applied consumer by consumer and
// re-read x and dX every time: 4.8 GB per CU-Net-2 step instead of 2.9).  For a consumer that reads X
// through the nearest-upsample map (models/cu_net.py:250,265) the four children of a source pixel
// share x: A*sum(dz) + 4E - 4D*x, which IS the backward of the index map (y>>1, x>>1).
// NSRC consumers, all plain (UPS = 0) or all through the upsample map (UPS = 1): compile-time source
// indices keep every load of an element in flight together (a runtime loop would chain them).
// V = channels per work item: 4 (one 16-byte fp32 piece) or, with bf16 gradient tensors, 8 (one 16-byte bf16 piece: an
// 8-byte access per lane reaches only 0.54-0.70x of the 16-byte rate, MI355X_MICROARCH.md)
template <int XB, int V> __device__ __forceinline__ void ldxv(const float* base, size_t off, float (&o)[V]) {
    if constexpr (V == 4) {
        const float4 v = ldx4<XB>(base, off);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        static_assert(XB == 1, "8-channel items are the bf16 path");
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + off);
        o[0] = bf16_bits_lo(v.x); o[1] = bf16_bits_hi(v.x); o[2] = bf16_bits_lo(v.y); o[3] = bf16_bits_hi(v.y);
        o[4] = bf16_bits_lo(v.z); o[5] = bf16_bits_hi(v.z); o[6] = bf16_bits_lo(v.w); o[7] = bf16_bits_hi(v.w);
    }
}
template <int XB, int V> __device__ __forceinline__ void stxv(float* base, size_t off, const float (&v)[V]) {
    if constexpr (V == 4) {
        stx4<XB>(base, off, make_float4(v[0], v[1], v[2], v[3]));
    } else {
        uint4 q;
        q.x = (unsigned)f32_to_bf16_rne(v[0]) | ((unsigned)f32_to_bf16_rne(v[1]) << 16);
        q.y = (unsigned)f32_to_bf16_rne(v[2]) | ((unsigned)f32_to_bf16_rne(v[3]) << 16);
        q.z = (unsigned)f32_to_bf16_rne(v[4]) | ((unsigned)f32_to_bf16_rne(v[5]) << 16);
        q.w = (unsigned)f32_to_bf16_rne(v[6]) | ((unsigned)f32_to_bf16_rne(v[7]) << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + off) = q;
    }
}

template <int NSRC, int UPS, int XBG, int V = 4>      // XBG: 0 fp32, 1 x bf16, 2 x / dz / gx bf16
__global__ __launch_bounds__(256) void grad_gather_kernel(const GradGatherArgs p) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cE = reinterpret_cast<float*>(smem);       // [C]   sum of E (x4 for upsampled consumers)
    float* cD = cE + p.C;                             // [C]   sum of D
    float* cA = cD + p.C;                             // [NSRC][C]
    const int tid = threadIdx.x;
    constexpr double mult = UPS ? 4.0 : 1.0;
    const double invM = 1.0 / ((double)p.rows * mult);
    for (int c = tid; c < p.C; c += 256) {
        const double mean = p.stats[c] / p.count;
        double var = p.stats[p.C + c] / p.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        double Es = 0.0, Ds = 0.0;
#pragma unroll
        for (int e = 0; e < NSRC; ++e) {
            const int cc = p.src[e].choff + c;
            const double scale = (double)p.src[e].gamma[cc] * istd;
            const double c1 = p.src[e].red[cc] * invM;
            const double c2 = p.src[e].red[p.src[e].lddz + cc] * invM;
            const double D = scale * c2 * istd;
            cA[e * p.C + c] = (float)scale;
            Es += mult * (D * mean - scale * c1);
            Ds += mult * D;
        }
        cE[c] = (float)Es;
        cD[c] = (float)Ds;
    }
    __syncthreads();

    const int gv = p.C / V;
    const int HW = p.H * p.W;
    const long total = (long)p.rows * gv;
    constexpr int U = UPS ? 4 : 1;
    for (long idx = (long)blockIdx.x * 256 + tid; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / gv;
        const int c = V * (int)(idx - row * gv);
        size_t drow = (size_t)row;                    // row of the consumers' dz
        if (UPS) {                                    // top-left child in a consumer at twice the resolution
            const int ni = (int)(row / HW);
            const int rm = (int)(row - (long)ni * HW);
            const int ys = rm / p.W, xs = rm - ys * p.W;
            drow = (size_t)ni * 4 * HW + (size_t)(2 * ys) * (2 * p.W) + 2 * xs;
        }
        float d[NSRC][U][V];
#pragma unroll
        for (int e = 0; e < NSRC; ++e) {
            const size_t b = drow * p.src[e].lddz + p.src[e].choff + c;
            ldxv<GB, V>(p.src[e].dz, b, d[e][0]);
            if (UPS) {
                ldxv<GB, V>(p.src[e].dz, b + p.src[e].lddz, d[e][1 % U]);
                ldxv<GB, V>(p.src[e].dz, b + (size_t)2 * p.W * p.src[e].lddz, d[e][2 % U]);
                ldxv<GB, V>(p.src[e].dz, b + (size_t)(2 * p.W + 1) * p.src[e].lddz, d[e][3 % U]);
            }
        }
        float x[V], o[V], r[V];
        ldxv<XB, V>(p.x, (size_t)row * p.ld + c, x);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = 0.f;
        if (p.accumulate) ldxv<GB, V>(p.gx, (size_t)row * p.ld + c, o);
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] = cE[c + k] - cD[c + k] * x[k];
#pragma unroll
        for (int e = 0; e < NSRC; ++e) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float v = d[e][0][k];
                if (UPS) v = (d[e][0][k] + d[e][1 % U][k]) + (d[e][2 % U][k] + d[e][3 % U][k]);
                r[k] = fmaf(cA[e * p.C + c + k], v, r[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) r[k] += o[k];
        stxv<GB, V>(p.gx, (size_t)row * p.ld + c, r);
    }
}

// The same gather with the CHANNEL PIECE FIXED PER THREAD (round 5): when the pieces of a row divide the block (C / 4 = 8 or 32 -- every
// tensor of the fp32 path), thread t owns piece t % gv of rows t / gv, + 256 / gv, ...: its E, D and A coefficients are loop-invariant
// registers (the flat-index kernel above re-reads (2 + NSRC) x 4 floats from LDS and does a 64-bit division per 16-byte piece), row
// arithmetic is 32-bit, two rows are in flight per thread (R = 2, plain consumers), and the launcher sizes the grid so that every
// block walks the same number of row groups (3072 groups on 2048 blocks was 2 rounds for half of them, 1 for the rest).  Element for
// element the same operations in the same order as the kernel above: bit-identical results.
// (round 6: the body is a device function of (arguments, block index, block count) so that two problems can share a launch --
// gather_pool_pair_kernel below.  POOL = 1: the gathered tensor is the output of a 2 x 2 max-pool; the same thread routes its piece of
// the gradient to the window's first arg-max of the pre-pool tensor `px` (pool_bwd_kernel's operation) and writes zeros to the other three.)
struct PoolRoute {
    const float* px;       // pre-pool activations [N * 2H * 2W][pC]   (H, W: the gathered = pooled tensor's)
    float* pgx;            // their gradient
    int pC;
};

template <int NSRC, int UPS, int XBG, int V, bool POOL>
__device__ __forceinline__ void gather_rows_body(const GradGatherArgs& p, int gshift, int bid, int nblocks, const PoolRoute& pr, char* smem) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    static_assert(!(POOL && UPS), "a pooled tensor is read at its own resolution");
    float* cE = reinterpret_cast<float*>(smem);
    float* cD = cE + p.C;
    float* cA = cD + p.C;
    const int tid = threadIdx.x;
    constexpr double mult = UPS ? 4.0 : 1.0;
    const double invM = 1.0 / ((double)p.rows * mult);
    for (int c = tid; c < p.C; c += 256) {
        const double mean = p.stats[c] / p.count;
        double var = p.stats[p.C + c] / p.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        double Es = 0.0, Ds = 0.0;
#pragma unroll
        for (int e = 0; e < NSRC; ++e) {
            const int cc = p.src[e].choff + c;
            const double scale = (double)p.src[e].gamma[cc] * istd;
            const double c1 = p.src[e].red[cc] * invM;
            const double c2 = p.src[e].red[p.src[e].lddz + cc] * invM;
            const double D = scale * c2 * istd;
            cA[e * p.C + c] = (float)scale;
            Es += mult * (D * mean - scale * c1);
            Ds += mult * D;
        }
        cE[c] = (float)Es;
        cD[c] = (float)Ds;
    }
    __syncthreads();

    const int c = V * (tid & ((1 << gshift) - 1));
    const int rpb = 256 >> gshift;                     // rows per block and round
    float rE[V], rD[V], rA[NSRC][V];
#pragma unroll
    for (int k = 0; k < V; ++k) { rE[k] = cE[c + k]; rD[k] = cD[c + k]; }
#pragma unroll
    for (int e = 0; e < NSRC; ++e)
#pragma unroll
        for (int k = 0; k < V; ++k) rA[e][k] = cA[e * p.C + c + k];
    const int HW = p.H * p.W;
    constexpr int U = UPS ? 4 : 1;
    constexpr int R = UPS ? 1 : 2;                     // rows in flight per thread
    const int stride = nblocks * rpb;
    for (int row0 = bid * rpb + (tid >> gshift); row0 < p.rows; row0 += R * stride) {
        float d[R][NSRC][U][V], x[R][V], o[R][V];
        float win[POOL ? R : 1][4][V];                 // POOL: the 2 x 2 window of the pre-pool tensor
        size_t woff[POOL ? R : 1][4];
        bool live[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = row0 + q * stride;
            live[q] = row < p.rows;
            const int rr = live[q] ? row : row0;       // (a dead slot re-reads row0: no branch around the requests)
            size_t drow = (size_t)rr;
            if (UPS) {
                const int ni = rr / HW;
                const int rm = rr - ni * HW;
                const int ys = rm / p.W, xs = rm - ys * p.W;
                drow = (size_t)ni * 4 * HW + (size_t)(2 * ys) * (2 * p.W) + 2 * xs;
            }
#pragma unroll
            for (int e = 0; e < NSRC; ++e) {
                const size_t b = drow * p.src[e].lddz + p.src[e].choff + c;
                ldxv<GB, V>(p.src[e].dz, b, d[q][e][0]);
                if (UPS) {
                    ldxv<GB, V>(p.src[e].dz, b + p.src[e].lddz, d[q][e][1 % U]);
                    ldxv<GB, V>(p.src[e].dz, b + (size_t)2 * p.W * p.src[e].lddz, d[q][e][2 % U]);
                    ldxv<GB, V>(p.src[e].dz, b + (size_t)(2 * p.W + 1) * p.src[e].lddz, d[q][e][3 % U]);
                }
            }
            ldxv<XB, V>(p.x, (size_t)rr * p.ld + c, x[q]);
#pragma unroll
            for (int k = 0; k < V; ++k) o[q][k] = 0.f;
            if (p.accumulate) ldxv<GB, V>(p.gx, (size_t)rr * p.ld + c, o[q]);
            if constexpr (POOL) {
                const int ni = rr / HW;
                const int rm = rr - ni * HW;
                const int yo = rm / p.W, xo = rm - yo * p.W;
                const size_t m00 = ((size_t)ni * (2 * p.H) + 2 * yo) * (2 * p.W) + 2 * xo;
                woff[q][0] = m00; woff[q][1] = m00 + 1; woff[q][2] = m00 + 2 * p.W; woff[q][3] = m00 + 2 * p.W + 1;
#pragma unroll
                for (int w = 0; w < 4; ++w) ldxv<XB, V>(pr.px, woff[q][w] * pr.pC + c, win[q][w]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            float r[V];
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = rE[k] - rD[k] * x[q][k];
#pragma unroll
            for (int e = 0; e < NSRC; ++e) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float v = d[q][e][0][k];
                    if (UPS) v = (d[q][e][0][k] + d[q][e][1 % U][k]) + (d[q][e][2 % U][k] + d[q][e][3 % U][k]);
                    r[k] = fmaf(rA[e][k], v, r[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] += o[q][k];
            if (live[q]) stxv<GB, V>(p.gx, (size_t)(row0 + q * stride) * p.ld + c, r);
            if constexpr (POOL) {
                if (live[q]) {
                    float outw[4][V];
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        // what pool_bwd_kernel would read back from the pooled gradient tensor: the stored (bf16: rounded) value
                        const float gv = GB ? bf16_bits_lo((unsigned)f32_to_bf16_rne(r[k])) : r[k];
                        int am = 0;
                        float best = win[q][0][k];
#pragma unroll
                        for (int w = 1; w < 4; ++w)
                            if (win[q][w][k] > best) { best = win[q][w][k]; am = w; }
#pragma unroll
                        for (int w = 0; w < 4; ++w) outw[w][k] = (w == am) ? gv : 0.f;
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) stxv<GB, V>(pr.pgx, woff[q][w] * pr.pC + c, outw[w]);
                }
            }
        }
    }
}

template <int NSRC, int UPS, int XBG, int V = 4>
__global__ __launch_bounds__(256) void grad_gather_rows_kernel(const GradGatherArgs p, int gshift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gather_rows_body<NSRC, UPS, XBG, V, false>(p, gshift, (int)blockIdx.x, (int)gridDim.x, PoolRoute{}, smem);
}

// Round 6: the three element-wise launches in front of a down block's adapter pair in backward -- gather(pool output) -> pool backward
// (dependent), gather(skip adapter output) (independent of both) -- as ONE launch: blockIdx.y = 0 gathers the pooled tensor's gradient and
// routes it through the arg-max map in the same thread (the pooled gradient is still written: tests and debuggers read it), blockIdx.y = 1
// gathers the skip adapter's.  At r <= 16 each of the three is a 4 - 7 us launch (8 of them per U-Net saved); at 64 x 64 the pooled
// gradient's re-read goes away.  Same operations on the same values as the three launches: bit-identical (tests/test_gpu_exact.py).
template <int NA, int NB, int XBG, int V>
__global__ __launch_bounds__(256) void gather_pool_pair_kernel(const GradGatherArgs a, const GradGatherArgs b, const PoolRoute pr, int gshift_a,
                                                               int gshift_b, int nblk_a, int nblk_b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < nblk_a) gather_rows_body<NA, 0, XBG, V, true>(a, gshift_a, (int)blockIdx.x, nblk_a, pr, smem);
    } else {
        if ((int)blockIdx.x < nblk_b) gather_rows_body<NB, 0, XBG, V, false>(b, gshift_b, (int)blockIdx.x, nblk_b, PoolRoute{}, smem);
    }
}

template <int UPS, int XB, int V = 4>
static hipError_t launch_gather_rows_n(const GradGatherArgs& a, dim3 grid, size_t smem, int gshift, hipStream_t s) {
    switch (a.nsrc) {
#define CUNET_G(N) case N: hipLaunchKernelGGL((grad_gather_rows_kernel<N, UPS, XB, V>), grid, dim3(256), smem, s, a, gshift); break;
        CUNET_G(1) CUNET_G(2) CUNET_G(3) CUNET_G(4) CUNET_G(5) CUNET_G(6) CUNET_G(7) CUNET_G(8)
#undef CUNET_G
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int UPS, int XB, int V>
static hipError_t launch_gather_nv(const GradGatherArgs& a, dim3 grid, size_t smem, hipStream_t s) {
    switch (a.nsrc) {
#define CUNET_G(N) case N: hipLaunchKernelGGL((grad_gather_kernel<N, UPS, XB, V>), grid, dim3(256), smem, s, a); break;
        CUNET_G(1) CUNET_G(2) CUNET_G(3) CUNET_G(4) CUNET_G(5) CUNET_G(6) CUNET_G(7) CUNET_G(8)
#undef CUNET_G
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int UPS, int XB>
static hipError_t launch_gather_n(const GradGatherArgs& a, dim3 grid, size_t smem, hipStream_t s) {
    switch (a.nsrc) {
#define CUNET_G(N) case N: hipLaunchKernelGGL((grad_gather_kernel<N, UPS, XB>), grid, dim3(256), smem, s, a); break;
        CUNET_G(1) CUNET_G(2) CUNET_G(3) CUNET_G(4) CUNET_G(5) CUNET_G(6) CUNET_G(7) CUNET_G(8)
#undef CUNET_G
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// All sources of one launch share the access map: a.src[*].ups must be equal (the caller groups them).
hipError_t launch_grad_gather(const GradGatherArgs& a, int num_cus, hipStream_t s) {
    const long total = (long)a.rows * (a.C / 4);
    long gx = (total + 255) / 256;
    if (gx > 8L * num_cus) gx = 8L * num_cus;
    if (gx < 1) gx = 1;
    const size_t smem = (size_t)a.C * 4 * (2 + a.nsrc);
#ifndef CUNET_GATHER_FLAT      // (probe builds: -DCUNET_GATHER_FLAT keeps every launch on the flat-index kernel of rounds 1-4)
    {   // fixed channel piece per thread: pieces per row a power of two <= 256 (every tensor of the network)
        bool v8 = a.xbf16 == 2 && a.C % 8 == 0 && a.ld % 8 == 0;             // bf16 gradient tensors: 16-byte pieces of 8 channels
        for (int e = 0; e < a.nsrc; ++e) v8 = v8 && a.src[e].lddz % 8 == 0 && a.src[e].choff % 8 == 0;
        const int V = v8 ? 8 : 4;
        const int gv = a.C / V;
        if ((a.xbf16 != 2 || v8) && a.C % V == 0 && gv >= 1 && gv <= 256 && (gv & (gv - 1)) == 0 && (long)a.rows * a.ld < (1L << 31)) {
            int gshift = 0;
            while ((1 << gshift) < gv) ++gshift;
            const int rpb = 256 >> gshift;
            const long groups = ((long)a.rows + rpb - 1) / rpb;                 // row groups of one block round
            const int R = a.src[0].ups ? 1 : 2;
            long blocks = std::min(groups, 8L * num_cus);
            const long rounds = (groups + blocks * R - 1) / (blocks * R);       // rounds of R row groups per block ...
            blocks = std::max(1L, (groups + rounds * R - 1) / (rounds * R));    // ... and the fewest blocks that need no more
            const dim3 grid((unsigned)blocks);
            if (v8) return a.src[0].ups ? launch_gather_rows_n<1, 2, 8>(a, grid, smem, gshift, s) : launch_gather_rows_n<0, 2, 8>(a, grid, smem, gshift, s);
            if (a.xbf16) return a.src[0].ups ? launch_gather_rows_n<1, 1>(a, grid, smem, gshift, s) : launch_gather_rows_n<0, 1>(a, grid, smem, gshift, s);
            return a.src[0].ups ? launch_gather_rows_n<1, 0>(a, grid, smem, gshift, s) : launch_gather_rows_n<0, 0>(a, grid, smem, gshift, s);
        }
    }
#endif
    if (a.xbf16 == 2) {
        bool v8 = a.C % 8 == 0 && a.ld % 8 == 0;                 // 16-byte bf16 pieces everywhere
        for (int e = 0; e < a.nsrc; ++e) v8 = v8 && a.src[e].lddz % 8 == 0 && a.src[e].choff % 8 == 0;
        if (v8) {
            const long total8 = (long)a.rows * (a.C / 8);
            long g8 = (total8 + 255) / 256;
            if (g8 > 8L * num_cus) g8 = 8L * num_cus;
            if (g8 < 1) g8 = 1;
            return a.src[0].ups ? launch_gather_nv<1, 2, 8>(a, dim3((unsigned)g8), smem, s) : launch_gather_nv<0, 2, 8>(a, dim3((unsigned)g8), smem, s);
        }
        return a.src[0].ups ? launch_gather_n<1, 2>(a, dim3((unsigned)gx), smem, s) : launch_gather_n<0, 2>(a, dim3((unsigned)gx), smem, s);
    }
    if (a.xbf16) return a.src[0].ups ? launch_gather_n<1, 1>(a, dim3((unsigned)gx), smem, s) : launch_gather_n<0, 1>(a, dim3((unsigned)gx), smem, s);
    return a.src[0].ups ? launch_gather_n<1, 0>(a, dim3((unsigned)gx), smem, s) : launch_gather_n<0, 0>(a, dim3((unsigned)gx), smem, s);
}

// geometry of a gather on the fixed-channel-piece kernel (the one launch_grad_gather would pick): false when the tensor's shape keeps the
// flat-index kernel
static bool gather_rows_geometry(const GradGatherArgs& a, int num_cus, int& V, int& gshift, long& blocks) {
    bool v8 = a.xbf16 == 2 && a.C % 8 == 0 && a.ld % 8 == 0;
    for (int e = 0; e < a.nsrc; ++e) v8 = v8 && a.src[e].lddz % 8 == 0 && a.src[e].choff % 8 == 0;
    V = v8 ? 8 : 4;
    const int gv = a.C / V;
    if (!((a.xbf16 != 2 || v8) && a.C % V == 0 && gv >= 1 && gv <= 256 && (gv & (gv - 1)) == 0 && (long)a.rows * a.ld < (1L << 31))) return false;
    gshift = 0;
    while ((1 << gshift) < gv) ++gshift;
    const int rpb = 256 >> gshift;
    const long groups = ((long)a.rows + rpb - 1) / rpb;
    const int R = a.src[0].ups ? 1 : 2;
    blocks = std::min(groups, 8L * num_cus);
    const long rounds = (groups + blocks * R - 1) / (blocks * R);
    blocks = std::max(1L, (groups + rounds * R - 1) / (rounds * R));
    return true;
}

template <int NA, int XBG, int V>
static hipError_t launch_gpp_b(const GradGatherArgs& a, const GradGatherArgs& b, const PoolRoute& pr, dim3 grid, size_t smem, int ga, int gb,
                               int na, int nb, hipStream_t s) {
    switch (b.nsrc) {
#define CUNET_G(N) case N: hipLaunchKernelGGL((gather_pool_pair_kernel<NA, N, XBG, V>), grid, dim3(256), smem, s, a, b, pr, ga, gb, na, nb); break;
        CUNET_G(1) CUNET_G(2) CUNET_G(3) CUNET_G(4)
#undef CUNET_G
        default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}
template <int XBG, int V>
static hipError_t launch_gpp_a(const GradGatherArgs& a, const GradGatherArgs& b, const PoolRoute& pr, dim3 grid, size_t smem, int ga, int gb,
                               int na, int nb, hipStream_t s) {
    switch (a.nsrc) {
        case 1: return launch_gpp_b<1, XBG, V>(a, b, pr, grid, smem, ga, gb, na, nb, s);
        case 2: return launch_gpp_b<2, XBG, V>(a, b, pr, grid, smem, ga, gb, na, nb, s);
        case 3: return launch_gpp_b<3, XBG, V>(a, b, pr, grid, smem, ga, gb, na, nb, s);
        case 4: return launch_gpp_b<4, XBG, V>(a, b, pr, grid, smem, ga, gb, na, nb, s);
        default: return hipErrorNotSupported;
    }
}

// `a`: the gather of a pooled tensor's gradient (its geometry = the pooled tensor's), routed on to `pool.gx` through the arg-max map of
// `pool.x` [N][2H][2W][C]; `b`: an independent plain gather.  hipErrorNotSupported, nothing launched, when the pair does not fit this kernel
// (up-sampled consumers, more than four consumers, an accumulating pass, mixed piece widths): the caller runs the three launches.
hipError_t launch_gather_pool_pair(const GradGatherArgs& a, const GradGatherArgs& b, const PoolArgs& pool, int num_cus, hipStream_t s) {
    if (a.nsrc < 1 || b.nsrc < 1 || a.accumulate || b.accumulate || a.xbf16 != b.xbf16 || a.xbf16 != pool.xbf16) return hipErrorNotSupported;
    for (int e = 0; e < a.nsrc; ++e) if (a.src[e].ups) return hipErrorNotSupported;
    for (int e = 0; e < b.nsrc; ++e) if (b.src[e].ups) return hipErrorNotSupported;
    if (pool.C != a.C || a.ld != a.C || pool.H != 2 * a.H || pool.W != 2 * a.W || (long)pool.N * a.H * a.W != (long)a.rows) return hipErrorNotSupported;
    int va, vb, ga, gb;
    long na, nb;
    if (!gather_rows_geometry(a, num_cus, va, ga, na) || !gather_rows_geometry(b, num_cus, vb, gb, nb) || va != vb) return hipErrorNotSupported;
    if ((long)pool.N * pool.H * pool.W * pool.C >= (1L << 31)) return hipErrorNotSupported;
    PoolRoute pr{pool.x, pool.gx, pool.C};
    const size_t smem = std::max((size_t)a.C * 4 * (2 + a.nsrc), (size_t)b.C * 4 * (2 + b.nsrc));
    const dim3 grid((unsigned)std::max(na, nb), 2);
    if (a.xbf16 == 2) return va == 8 ? launch_gpp_a<2, 8>(a, b, pr, grid, smem, ga, gb, (int)na, (int)nb, s) : launch_gpp_a<2, 4>(a, b, pr, grid, smem, ga, gb, (int)na, (int)nb, s);
    if (a.xbf16) return launch_gpp_a<1, 4>(a, b, pr, grid, smem, ga, gb, (int)na, (int)nb, s);
    return launch_gpp_a<0, 4>(a, b, pr, grid, smem, ga, gb, (int)na, (int)nb, s);
}

// dgamma / dbeta of a batch of BatchNorms from their backward reductions (one block per BatchNorm)
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const BnParamGradArgs p) {
    const auto& e = p.e[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += 256) {
        e.dbeta[c] = (float)e.red[c];
        e.dgamma[c] = (float)e.red[e.C + c];
    }
}

hipError_t launch_bn_param_grad(const BnParamGradArgs& a, hipStream_t s) {
    if (a.n < 1) return hipSuccess;
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(a.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 2x2/2 max-pool over NHWC (+ fp64 batch statistics of the pooled tensor).  MODE 0: plain
// (models/cu_net.py:249,260); MODE 1: the stem's BN -> ReLU -> pool (models/cu_net.py:301-303).

constexpr int POOL_U = 4;

template <int MODE>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const PoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int g4 = p.C >> 2;
    const int rpi = 256 / g4;                      // rows per block iteration
    const int g = tid % g4;
    const int ry = tid / g4;
    const bool active = ry < rpi;
    float* sc = reinterpret_cast<float*>(smem);
    float* sh = sc + p.C;
    double* red = reinterpret_cast<double*>(smem + (size_t)(MODE == 1 ? 2 * p.C * 4 : 0));   // [rpi][C][2]
    if (MODE == 1) {
        for (int c = tid; c < p.C; c += 256) {
            double mean, istd;
            if (p.training) {
                mean = p.xstats[c] / p.count;
                double var = p.xstats[p.C + c] / p.count - mean * mean;
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
        __syncthreads();
    }
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const long rows = (long)p.N * Ho * Wo;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (active) {
        float4 S = make_float4(1.f, 1.f, 1.f, 1.f), Hh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) {
            S = *reinterpret_cast<const float4*>(sc + 4 * g);
            Hh = *reinterpret_cast<const float4*>(sh + 4 * g);
        }
        // POOL_U output rows per thread and iteration, all 4 * POOL_U loads requested before the first is used: the launch is a few
        // hundred fat blocks (one round, <= 2 per CU) instead of a thousand thin ones -- a third of the fp64 statistics atomics, which
        // all blocks aim at the same 2 * C addresses, and no load -> store -> load chain across grid-stride iterations
        const long stride = (long)gridDim.x * rpi;
        for (long row0 = (long)blockIdx.x * rpi + ry; row0 < rows; row0 += stride * POOL_U) {
          float4 vv[POOL_U][4];
#pragma unroll
          for (int u = 0; u < POOL_U; ++u) {
            const long row = row0 + u * stride < rows ? row0 + u * stride : row0;
            const int ni = (int)(row / (Ho * Wo));
            const int rm = (int)(row - (long)ni * Ho * Wo);
            const int yo = rm / Wo, xo = rm - yo * Wo;
            const size_t m00 = ((size_t)ni * p.H + 2 * yo) * p.W + 2 * xo;
            vv[u][0] = ldg4(p.x + m00 * p.C + 4 * g);
            vv[u][1] = ldg4(p.x + (m00 + 1) * p.C + 4 * g);
            vv[u][2] = ldg4(p.x + (m00 + p.W) * p.C + 4 * g);
            vv[u][3] = ldg4(p.x + (m00 + p.W + 1) * p.C + 4 * g);
          }
#pragma unroll
          for (int u = 0; u < POOL_U; ++u) {
            const long row = row0 + u * stride;
            if (row >= rows) break;
            float4 v[4] = {vv[u][0], vv[u][1], vv[u][2], vv[u][3]};
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k].x = fmaxf(fmaf(v[k].x, S.x, Hh.x), 0.f);
                    v[k].y = fmaxf(fmaf(v[k].y, S.y, Hh.y), 0.f);
                    v[k].z = fmaxf(fmaf(v[k].z, S.z, Hh.z), 0.f);
                    v[k].w = fmaxf(fmaf(v[k].w, S.w, Hh.w), 0.f);
                }
            }
            float4 m;
            m.x = fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x));
            m.y = fmaxf(fmaxf(v[0].y, v[1].y), fmaxf(v[2].y, v[3].y));
            m.z = fmaxf(fmaxf(v[0].z, v[1].z), fmaxf(v[2].z, v[3].z));
            m.w = fmaxf(fmaxf(v[0].w, v[1].w), fmaxf(v[2].w, v[3].w));
            *reinterpret_cast<float4*>(p.y + (size_t)row * p.C + 4 * g) = m;
            s1[0] += m.x; s2[0] += (double)m.x * m.x;
            s1[1] += m.y; s2[1] += (double)m.y * m.y;
            s1[2] += m.z; s2[2] += (double)m.z * m.z;
            s1[3] += m.w; s2[3] += (double)m.w * m.w;
          }
        }
    }
    if (p.ystats == nullptr) return;
    if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[((size_t)ry * p.C + 4 * g + e) * 2 + 0] = s1[e];
            red[((size_t)ry * p.C + 4 * g + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        double a = 0.0, b = 0.0;
        for (int r = 0; r < rpi; ++r) {
            a += red[((size_t)r * p.C + c) * 2 + 0];
            b += red[((size_t)r * p.C + c) * 2 + 1];
        }
        atomic_add_f64e(p.ystats + c, a);
        atomic_add_f64e(p.ystats + p.C + c, b);
    }
}

static size_t pool_smem(int C, int mode) {
    const int g4 = C / 4;
    const int rpi = 256 / g4;
    return (size_t)(mode == 1 ? 2 * C * 4 : 0) + (size_t)rpi * C * 2 * 8;
}

hipError_t launch_pool_fwd(const PoolArgs& a, int mode, int num_cus, hipStream_t s) {
    const int rpi = 256 / (a.C / 4);
    const long rows = (long)a.N * (a.H / 2) * (a.W / 2);
    long gx = (rows + (long)rpi * POOL_U - 1) / ((long)rpi * POOL_U);
    if (gx > 2L * num_cus) gx = 2L * num_cus;
    if (gx < 1) gx = 1;
    if (mode == 0)
        hipLaunchKernelGGL(pool_fwd_kernel<0>, dim3((unsigned)gx), dim3(256), pool_smem(a.C, 0), s, a);
    else
        hipLaunchKernelGGL(pool_fwd_kernel<1>, dim3((unsigned)gx), dim3(256), pool_smem(a.C, 1), s, a);
    return hipGetLastError();
}

// max-pool backward: the gradient goes to the FIRST maximum in window order (0,0),(0,1),(1,0),(1,1)
// (torch CPU max_pool2d tie-break), every other input position gets 0.
template <int XBG>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const PoolArgs p) {
    constexpr int XB = XBG != 0, GB = XBG == 2;
    const int g4 = p.C >> 2;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const long total = (long)p.N * Ho * Wo * g4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / g4;
        const int g = (int)(idx - row * g4);
        const int ni = (int)(row / (Ho * Wo));
        const int rm = (int)(row - (long)ni * Ho * Wo);
        const int yo = rm / Wo, xo = rm - yo * Wo;
        const size_t m00 = ((size_t)ni * p.H + 2 * yo) * p.W + 2 * xo;
        const size_t off[4] = {m00, m00 + 1, m00 + p.W, m00 + p.W + 1};
        float v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 t = ldx4<XB>(p.x, off[k] * p.C + 4 * g);
            v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
        }
        const float4 gq = ldx4<GB>(p.gy, (size_t)row * p.C + 4 * g);
        const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
        float o[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int am = 0;
            float best = v[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][e] > best) { best = v[k][e]; am = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][e] = (k == am) ? gv[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            stx4<GB>(p.gx, off[k] * p.C + 4 * g, make_float4(o[k][0], o[k][1], o[k][2], o[k][3]));
    }
}

hipError_t launch_pool_bwd(const PoolArgs& a, int num_cus, hipStream_t s) {
    const long total = (long)a.N * (a.H / 2) * (a.W / 2) * (a.C / 4);
    long gx = (total + 255) / 256;
    if (gx > 8L * num_cus) gx = 8L * num_cus;
    if (gx < 1) gx = 1;
    if (a.xbf16 == 2) hipLaunchKernelGGL(pool_bwd_kernel<2>, dim3((unsigned)gx), dim3(256), 0, s, a);
    else if (a.xbf16) hipLaunchKernelGGL(pool_bwd_kernel<1>, dim3((unsigned)gx), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pool_bwd_kernel<0>, dim3((unsigned)gx), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Stem backward (pool -> ReLU -> BN), two streaming passes because BN backward needs full-batch
// reductions.  PASS 0: reductions sum(dz), sum(dz*xhat) -> p.red.  PASS 1: dC = A*dz + E - D*x.
// dz is non-zero only at the window's first arg-max, and only where the BN output was > 0.
template <int PASS, int GB>      // GB = 1: gy (gradient of the pooled features) is stored as bf16
__global__ __launch_bounds__(256) void stem_bwd_kernel(const PoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int g4 = p.C >> 2;
    const int rpi = 256 / g4;
    const int g = tid % g4;
    const int ry = tid / g4;
    const bool active = ry < rpi;
    float* sc = reinterpret_cast<float*>(smem);
    float* sh = sc + p.C;
    float* mu = sh + p.C;
    float* is = mu + p.C;
    float* cE = is + p.C;      // PASS 1
    float* cD = cE + p.C;
    double* red = reinterpret_cast<double*>(cD + p.C);   // PASS 0: [rpi][C][2]
    const double invM = 1.0 / p.count;
    for (int c = tid; c < p.C; c += 256) {
        const double mean = p.xstats[c] / p.count;
        double var = p.xstats[p.C + c] / p.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double istd = 1.0 / sqrt(var + (double)BN_EPS);
        const double scale = (double)p.gamma[c] * istd;
        sc[c] = (float)scale;
        sh[c] = (float)((double)p.beta[c] - mean * scale);
        mu[c] = (float)mean;
        is[c] = (float)istd;
        if (PASS == 1) {
            const double c1 = p.red[c] * invM, c2 = p.red[p.C + c] * invM;
            const double D = scale * c2 * istd;
            cD[c] = (float)D;
            cE[c] = (float)(D * mean - scale * c1);
        }
    }
    __syncthreads();
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const long rows = (long)p.N * Ho * Wo;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (active) {
        float S[4], Hh[4], MU[4], IS[4], E[4], D[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            S[e] = sc[4 * g + e]; Hh[e] = sh[4 * g + e]; MU[e] = mu[4 * g + e]; IS[e] = is[4 * g + e];
            E[e] = PASS == 1 ? cE[4 * g + e] : 0.f;
            D[e] = PASS == 1 ? cD[4 * g + e] : 0.f;
        }
        for (long row = (long)blockIdx.x * rpi + ry; row < rows; row += (long)gridDim.x * rpi) {
            const int ni = (int)(row / (Ho * Wo));
            const int rm = (int)(row - (long)ni * Ho * Wo);
            const int yo = rm / Wo, xo = rm - yo * Wo;
            const size_t m00 = ((size_t)ni * p.H + 2 * yo) * p.W + 2 * xo;
            const size_t off[4] = {m00, m00 + 1, m00 + p.W, m00 + p.W + 1};
            float v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(p.x + off[k] * p.C + 4 * g);
                v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
            }
            const float4 gq = ldx4<GB>(p.gy, (size_t)row * p.C + 4 * g);
            const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
            float o[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int am = 0;
                float best = fmaxf(fmaf(v[0][e], S[e], Hh[e]), 0.f);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float a = fmaxf(fmaf(v[k][e], S[e], Hh[e]), 0.f);
                    if (a > best) { best = a; am = k; }
                }
                const float dz = best > 0.f ? gv[e] : 0.f;
                if (PASS == 0) {
                    float xa = v[0][e];
#pragma unroll
                    for (int k = 1; k < 4; ++k) xa = (k == am) ? v[k][e] : xa;
                    s1[e] += dz;
                    s2[e] += (double)(dz * ((xa - MU[e]) * IS[e]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k][e] = fmaf(-D[e], v[k][e], fmaf(S[e], (k == am) ? dz : 0.f, E[e]));      // (explicit: wgrad3_stem_kernel<true> computes the same two operations)
                }
            }
            if (PASS == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4*>(p.gx + off[k] * p.C + 4 * g) =
                        make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
            }
        }
    }
    if (PASS == 0) {
        if (active) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[((size_t)ry * p.C + 4 * g + e) * 2 + 0] = s1[e];
                red[((size_t)ry * p.C + 4 * g + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int c = tid; c < p.C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < rpi; ++r) {
                a += red[((size_t)r * p.C + c) * 2 + 0];
                b += red[((size_t)r * p.C + c) * 2 + 1];
            }
            atomic_add_f64e(p.red + c, a);
            atomic_add_f64e(p.red + p.C + c, b);
        }
    }
}

// writes dgamma/dbeta of the stem BN from its reductions (tiny)
__global__ void stem_bn_param_grad_kernel(const double* red, float* dgamma, float* dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        dbeta[c] = (float)red[c];
        dgamma[c] = (float)red[C + c];
    }
}

hipError_t launch_stem_bwd(const PoolArgs& a, int pass, float* dgamma, float* dbeta, int num_cus, hipStream_t s) {
    const int rpi = 256 / (a.C / 4);
    const long rows = (long)a.N * (a.H / 2) * (a.W / 2);
    long gx = (rows + rpi - 1) / rpi;
    if (gx > 4L * num_cus) gx = 4L * num_cus;
    if (gx < 1) gx = 1;
    const size_t smem = (size_t)6 * a.C * 4 + (size_t)rpi * a.C * 2 * 8;
    if (pass == 0) {
        if (a.xbf16 == 2) hipLaunchKernelGGL((stem_bwd_kernel<0, 1>), dim3((unsigned)gx), dim3(256), smem, s, a);
        else hipLaunchKernelGGL((stem_bwd_kernel<0, 0>), dim3((unsigned)gx), dim3(256), smem, s, a);
    } else if (pass == 2) {      // the BatchNorm parameter gradients only: the dz pass is fused into the stem's weight gradient (stem_fuse_dz)
        hipLaunchKernelGGL(stem_bn_param_grad_kernel, dim3((a.C + 255) / 256), dim3(256), 0, s,
                           (const double*)a.red, dgamma, dbeta, a.C);
    } else {
        if (a.xbf16 == 2) hipLaunchKernelGGL((stem_bwd_kernel<1, 1>), dim3((unsigned)gx), dim3(256), smem, s, a);
        else hipLaunchKernelGGL((stem_bwd_kernel<1, 0>), dim3((unsigned)gx), dim3(256), smem, s, a);
        if (pass == 3) return hipGetLastError();      // (the dz pass alone: cunet_debug_materialise)
        hipLaunchKernelGGL(stem_bn_param_grad_kernel, dim3((a.C + 255) / 256), dim3(256), 0, s,
                           (const double*)a.red, dgamma, dbeta, a.C);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC at the public boundary (heat maps / their gradients / targets), 32x32 LDS tiles.
// nchw [N][C][HW]   nhwc [N*HW][ld]  (channels C..ld-1 are written as zeros)
template <int TO_NHWC>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int C, int HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (TO_NHWC) {
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, pp = p0 + tx;
            tile[j][tx] = (c < C && pp < HW) ? src[((size_t)n * C + c) * HW + pp] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int pp = p0 + j, c = c0 + tx;
            if (pp < HW && c < ld) dst[((size_t)n * HW + pp) * ld + c] = tile[tx][j];
        }
    } else {
        for (int j = ty; j < 32; j += 8) {
            const int pp = p0 + j, c = c0 + tx;
            tile[j][tx] = (pp < HW && c < C) ? src[((size_t)n * HW + pp) * ld + c] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, pp = p0 + tx;
            if (c < C && pp < HW) dst[((size_t)n * C + c) * HW + pp] = tile[tx][j];
        }
    }
}

hipError_t launch_transpose(const float* src, float* dst, int N, int C, int HW, int ld, int to_nhwc, hipStream_t s) {
    const dim3 grid((HW + 31) / 32, (ld + 31) / 32, N);
    if (to_nhwc)
        hipLaunchKernelGGL(transpose_kernel<1>, grid, dim3(256), 0, s, src, dst, C, HW, ld);
    else
        hipLaunchKernelGGL(transpose_kernel<0>, grid, dim3(256), 0, s, src, dst, C, HW, ld);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Pixelwise MSE of one head (cu-net.py:175-178): loss += sum((o-t)^2)/numel, dO = 2(o-t)/numel.
template <int GB>      // GB = 1: d(loss)/d(out) is stored as bf16
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                  float* __restrict__ dout, double* loss_acc,
                                                  long rows, int C, int ld, int ldd) {      // ldd >= ld: pitch of dout, pad columns zero
    __shared__ double wsum[4];
    const long total = rows * ldd;
    const double inv = 1.0 / ((double)rows * C);
    const float ginv = (float)(2.0 * inv);
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / ldd;
        const int c = (int)(i - r * ldd);
        float d = 0.f;
        if (c < C) d = out[r * ld + c] - tgt[r * ld + c];
        stx1<GB>(dout, (size_t)i, d * ginv);
        acc += (double)d * d;
    }
    for (int o = 32; o > 0; o >>= 1) {
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __shfl_xor(lo, o, 64); hi = __shfl_xor(hi, o, 64);
        acc += __hiloint2double(hi, lo);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f64e(loss_acc, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv);
}

__global__ void loss_finalize_kernel(const double* acc, float* loss) { *loss = (float)(*acc); }

hipError_t launch_mse(const float* out, const float* tgt, float* dout, double* loss_acc, long rows, int C, int ld, int ldd,
                      int grad_bf16, int num_cus, hipStream_t s) {
    if (ldd < ld) ldd = ld;
    long gx = (rows * ldd + 255) / 256;
    if (gx > 4L * num_cus) gx = 4L * num_cus;
    if (grad_bf16) hipLaunchKernelGGL(mse_kernel<1>, dim3((unsigned)gx), dim3(256), 0, s, out, tgt, dout, loss_acc, rows, C, ld, ldd);
    else hipLaunchKernelGGL(mse_kernel<0>, dim3((unsigned)gx), dim3(256), 0, s, out, tgt, dout, loss_acc, rows, C, ld, ldd);
    return hipGetLastError();
}

hipError_t launch_loss_finalize(const double* acc, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, acc, loss);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// running_mean / running_var / num_batches_tracked update for every BN of the net in ONE launch
// (nn.BatchNorm2d train-mode semantics: momentum 0.1, unbiased variance for the running estimate).
// mode 0: every entry once (forward); mode 1: only entries of BNs the reference re-runs under
// torch.utils.checkpoint (models/cu_net.py:30-31,58-59), applied from backward.
__global__ __launch_bounds__(256) void running_update_kernel(const RunStatEntry* tab, const double* stats_base,
                                                             float* buffers, int64_t* counters, int mode) {
    const RunStatEntry e = tab[blockIdx.x];
    if (mode == 1 && e.times < 2) return;
    const double mom = 0.1;
    for (int c = threadIdx.x; c < e.C; c += 256) {
        const double mean = stats_base[e.stats + c] / e.count;
        double var = stats_base[e.stats + e.C + c] / e.count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double unb = var * e.unbias;
        float* rm = buffers + e.rmean + c;
        float* rv = buffers + e.rvar + c;
        *rm = (float)((1.0 - mom) * (double)*rm + mom * mean);
        *rv = (float)((1.0 - mom) * (double)*rv + mom * unb);
    }
    if (threadIdx.x == 0 && e.counter >= 0) counters[e.counter] += 1;
}

hipError_t launch_running_update(const RunStatEntry* tab, int n, const double* stats_base, float* buffers,
                                 int64_t* counters, int mode, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(running_update_kernel, dim3(n), dim3(256), 0, s, tab, stats_base, buffers, counters, mode);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Weight repack: torch [Cout][Cin][taps] -> MFMA B-operand layouts [tap][K/4][Npad][4]
//   forward : K = Cin,  N = Cout
//   backward: K = Cout, N = Cin, taps flipped (transposed convolution for the data gradient)
__global__ __launch_bounds__(256) void repack_kernel(const RepackEntry* tab, const float* params, float* ws) {
    const RepackEntry e = tab[blockIdx.y];
    const float* w = params + e.src;
    {
        const long total = (long)e.taps * e.KpadF * e.NpadF;
        float* dst = ws + e.dstF;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int ee = (int)(i & 3);
            long r = i >> 2;
            const int n = (int)(r % e.NpadF); r /= e.NpadF;
            const int kq = (int)(r % (e.KpadF >> 2));
            const int t = (int)(r / (e.KpadF >> 2));
            const int k = 4 * kq + ee;
            dst[i] = (k < e.Cin && n < e.Cout) ? w[((size_t)n * e.Cin + k) * e.taps + t] : 0.f;
        }
    }
    if (e.dstB >= 0) {
        const long total = (long)e.taps * e.KpadB * e.NpadB;
        float* dst = ws + e.dstB;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int ee = (int)(i & 3);
            long r = i >> 2;
            const int n = (int)(r % e.NpadB); r /= e.NpadB;
            const int kq = (int)(r % (e.KpadB >> 2));
            const int t = (int)(r / (e.KpadB >> 2));
            const int k = 4 * kq + ee;          // output channel of the forward conv
            dst[i] = (k < e.Cout && n < e.Cin) ? w[((size_t)k * e.Cin + n) * e.taps + (e.taps - 1 - t)] : 0.f;
        }
    }
}

hipError_t launch_repack(const RepackEntry* tab, int n, const float* params, float* ws, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(repack_kernel, dim3(16, n), dim3(256), 0, s, tab, params, ws);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Fused RMSprop over the flat arena (torch.optim.RMSprop, momentum 0, centered False; cu-net.py:60-61):
//   v = alpha*v + (1-alpha)*g*g ;  p -= lr * g / (sqrt(v) + eps)
// torch's op order (torch/optim/rmsprop.py: v.mul_(alpha).addcmul_(g, g, value=1-alpha); p.addcdiv_(g, sqrt(v)+eps, value=-lr)),
// each operation rounded separately: `#pragma clang fp contract(off)` keeps hipcc (default -ffp-contract=fast) from fusing
// the multiplies into the adds (it did: `v_fma_f32 p, -lr, q, p`), plain `/` and sqrtf are the correctly rounded forms under
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt (v_sqrt_f32 + fix-up, v_div_scale / v_div_fmas / v_div_fixup;
// __fsqrt_rn is the 1-ulp v_sqrt_f32).  `oma` = (float)(1 - alpha) is formed in double on the host.
__device__ __forceinline__ void rmsprop_elem(float& p, float g, float& v, float lr, float alpha, float oma, float eps) {
#pragma clang fp contract(off)
    const float t0 = alpha * v;
    const float t1 = oma * g;
    const float t2 = t1 * g;
    v = t0 + t2;
    const float d = sqrtf(v) + eps;
    const float q = g / d;
    const float u = lr * q;
    p = p - u;
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ v, long n, float lr, float alpha, float oma,
                                                      float eps, float gscale) {
#pragma clang fp contract(off)
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        gg.x *= gscale; gg.y *= gscale; gg.z *= gscale; gg.w *= gscale;
        rmsprop_elem(pp.x, gg.x, vv.x, lr, alpha, oma, eps);
        rmsprop_elem(pp.y, gg.y, vv.y, lr, alpha, oma, eps);
        rmsprop_elem(pp.z, gg.z, vv.z, lr, alpha, oma, eps);
        rmsprop_elem(pp.w, gg.w, vv.w, lr, alpha, oma, eps);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        float pp = p[i], vv = v[i];
        rmsprop_elem(pp, g[i] * gscale, vv, lr, alpha, oma, eps);
        v[i] = vv;
        p[i] = pp;
    }
}

hipError_t launch_rmsprop(float* p, const float* g, float* v, long n, float lr, float alpha, float oma, float eps,
                          float gscale, hipStream_t s) {
    long gx = (n / 4 + 255) / 256;
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)gx), dim3(256), 0, s, p, g, v, n, lr, alpha, oma, eps, gscale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Landmark decode (pylib/Evaluation.py:6-23 get_preds): flat arg-max per (n,k) map with the
// LOWEST index winning ties (torch.max on CPU), x = idx % W + 1, y = floor(idx / H) + 1
// (the reference divides by size(2)), zeroed where max <= 0.  One wave per map; bit-exact.
__global__ __launch_bounds__(256) void get_preds_kernel(const float* __restrict__ heat, float* __restrict__ preds,
                                                        int maps, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int map = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= maps) return;
    const int HW = H * W;
    const float* h = heat + (size_t)map * HW;
    float best = -INFINITY;
    int bi = HW;                       // sentinel: larger than any index
    for (int i = lane; i < HW; i += 64) {
        const float v = h[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        if (bi >= HW) bi = 0;
        float x = (float)(bi % W + 1);
        float y = floorf((float)bi / (float)H) + 1.f;
        if (!(best > 0.f)) { x = 0.f; y = 0.f; }
        preds[(size_t)map * 2 + 0] = x;
        preds[(size_t)map * 2 + 1] = y;
    }
}

// final_preds (pylib/Evaluation.py:108-132) with rot == 0 (validation, data/mpii_for_mpii_22.py:120): arg-max,
// quarter-pixel shift toward the larger neighbour, +0.5, inverse crop transform (Evaluation.py:152-187 with
// size = 200): the reference builds the 3x3 matrix from float32 scalars, inverts it in float64 (LAPACK on an
// upper-triangular matrix: inv[0][2] = -(b * (1/a))) and truncates toward zero; the same sequence is used here.
// `inv` != nullptr (rot != 0, pylib/Evaluation.py:163-178): the caller hands over rows 0 and 1 of the inverted 3x3 transform per image
// ([N][6] float64, built on the host with the reference's own numpy operations -- cu_net_amd/trainer.py) and the kernel applies it as
// np.dot(t, [x - 1, y - 1, 1]) does: a 3-term float64 dot product per coordinate, accumulated in k order by fused multiply-adds from
// zero (what the BLAS dgemm kernels behind np.dot do on x86-64), then astype(int) + 1.
__global__ __launch_bounds__(256) void final_preds_kernel(const float* __restrict__ heat, const float* __restrict__ center,
                                                          const float* __restrict__ scale, const double* __restrict__ inv,
                                                          float* __restrict__ preds,
                                                          int maps, int K, int H, int W, int res0, int res1) {
    const int lane = threadIdx.x & 63;
    const int map = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= maps) return;
    const int HW = H * W;
    const float* h = heat + (size_t)map * HW;
    float best = -INFINITY;
    int bi = HW;
    for (int i = lane; i < HW; i += 64) {
        const float v = h[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane != 0) return;
    if (bi >= HW) bi = 0;
    float x = (float)(bi % W + 1);
    float y = floorf((float)bi / (float)H) + 1.f;
    if (!(best > 0.f)) { x = 0.f; y = 0.f; }
    const int px = (int)floorf(x), py = (int)floorf(y);
    if (px > 1 && px < res0 && py > 1 && py < res1) {
        const float dx = h[(py - 1) * W + px] - h[(py - 1) * W + px - 2];
        const float dy = h[py * W + px - 1] - h[(py - 2) * W + px - 1];
        x += (float)((dx > 0.f) - (dx < 0.f)) * 0.25f;
        y += (float)((dy > 0.f) - (dy < 0.f)) * 0.25f;
    }
    x += 0.5f;
    y += 0.5f;
    const int n = map / K;
    if (inv != nullptr) {
        const double* t = inv + (size_t)n * 6;
        const double xd = (double)(x - 1.f), yd = (double)(y - 1.f);      // (pts - 1 is float32 arithmetic, then concatenated to float64)
        const double nx = __fma_rn(t[2], 1.0, __fma_rn(t[1], yd, __dmul_rn(t[0], xd)));
        const double ny = __fma_rn(t[5], 1.0, __fma_rn(t[4], yd, __dmul_rn(t[3], xd)));
        preds[(size_t)map * 2 + 0] = (float)((long long)nx + 1);
        preds[(size_t)map * 2 + 1] = (float)((long long)ny + 1);
        return;
    }
    const float hh = 200.f * scale[n];                       // float32 arithmetic, as numpy does for float32 scalars
    const float a32 = (float)res0 / hh;
    const float b32 = (float)res0 * (-center[2 * n + 0] / hh + 0.5f);
    const float c32 = (float)res0 * (-center[2 * n + 1] / hh + 0.5f);
    const double ia = 1.0 / (double)a32;
    const double i02 = -((double)b32 * ia), i12 = -((double)c32 * ia);
    const double nx = __dadd_rn(__dmul_rn(ia, (double)(x - 1.f)), i02);
    const double ny = __dadd_rn(__dmul_rn(ia, (double)(y - 1.f)), i12);
    preds[(size_t)map * 2 + 0] = (float)((long long)nx + 1);   // astype(int): truncation toward zero
    preds[(size_t)map * 2 + 1] = (float)((long long)ny + 1);
}

hipError_t launch_final_preds(const float* heat, const float* center, const float* scale, const double* inv, float* preds, int N, int K,
                              int H, int W, int res0, int res1, hipStream_t s) {
    hipLaunchKernelGGL(final_preds_kernel, dim3((N * K + 3) / 4), dim3(256), 0, s, heat, center, scale, inv, preds, N * K, K, H, W, res0, res1);
    return hipGetLastError();
}

hipError_t launch_get_preds(const float* heat, float* preds, int maps, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(get_preds_kernel, dim3((maps + 3) / 4), dim3(256), 0, s, heat, preds, maps, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Flip test-time augmentation (cu-net.py:240-249): average the heat maps of an image with the
// un-flipped, channel-swapped heat maps of its mirror image.  One row of W floats per thread group.
__global__ __launch_bounds__(256) void flip_merge_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const int* __restrict__ perm, float* __restrict__ out,
                                                          int K, int H, int W, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const long row = idx / W;                  // (n*K + c)*H + y
        const int y = (int)(row % H);
        const long nc = row / H;
        const int c = (int)(nc % K);
        const long n = nc / K;
        const long src = ((n * K + perm[c]) * H + y) * (long)W + (W - 1 - x);
        out[idx] = (a[idx] + b[src]) / 2.0f;
    }
}

hipError_t launch_flip_merge(const float* a, const float* b, const int* perm, float* out, int N, int K, int H, int W, hipStream_t s) {
    const long total = (long)N * K * H * W;
    long gx = (total + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(flip_merge_kernel, dim3((unsigned)gx), dim3(256), 0, s, a, b, perm, out, K, H, W, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Gaussian target maps (pylib/HumanPts.py:35-76).  One block per map: zero fill, then the cropped patch whose
// centre is pixel (int(x), int(y)) -- int() truncates toward zero exactly as python's does on the float64 input.
__global__ __launch_bounds__(256) void render_targets_kernel(const double* __restrict__ pts, const float* __restrict__ patch,
                                                              int half, float* __restrict__ out, int H, int W) {
    const int m = blockIdx.x;
    float* o = out + (size_t)m * H * W;
    const double px = pts[2 * m], py = pts[2 * m + 1];
    const int ulx = (int)(px - (double)half), uly = (int)(py - (double)half);      // C conversion truncates toward zero
    const int brx = (int)(px + (double)half), bry = (int)(py + (double)half);
    const bool draw = px > 0.0 && py > 0.0 && !(ulx >= W || uly >= H || brx < 0 || bry < 0);
    const int size = 2 * half + 1;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        const int gx = x - ulx, gy = y - uly;
        float v = 0.f;
        if (draw && gx >= 0 && gx < size && gy >= 0 && gy < size && x <= brx && y <= bry) v = patch[gy * size + gx];
        o[i] = v;
    }
}

hipError_t launch_render_targets(const double* pts, const float* patch, int half, float* out, int NK, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(render_targets_kernel, dim3(NK), dim3(256), 0, s, pts, patch, half, out, H, W);
    return hipGetLastError();
}

}  // namespace cunet
