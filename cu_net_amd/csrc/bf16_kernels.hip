// bf16-storage inference forward (BASELINE config 3 direction: "bf16 storage + fp32 accumulate").
//
// Activations live as bf16 NHWC, weights as bf16 MFMA operands, BatchNorm (running statistics) + ReLU are applied
// in fp32 on load and re-rounded to bf16, the contraction is v_mfma_f32_32x32x16_bf16 with fp32 accumulators.
// Same structure as conv_kernels.hip (weight-stationary B operand in LDS, independent waves over 32-row tiles, the
// concat read in place segment by segment, nearest-upsample folded into the row index), at half the bytes per
// element.  Eval mode only: no batch statistics, no gradients.  The stem (7x7/2 on the fp32 image, BN-ReLU-pool)
// runs on the fp32 kernels and is converted once.
//
// Operand maps of v_mfma_f32_32x32x16_bf16: A lane l holds A[i = l&31][k = 8*(l>>5) .. +7], B lane l holds
// B[k = 8*(l>>5) .. +7][j = l&31], C/D as the f32 form (col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)).
#include <cstdlib>

#include "common.h"
#include "conv_common.h"

namespace cunet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {     // round to nearest even, (hi << 16) | lo
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

__device__ __forceinline__ double shfl_xor_d16(double v) {     // value of lane ^ 32
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, 32, 64);
    hi = __shfl_xor(hi, 32, 64);
    return __hiloint2double(hi, lo);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* gptr_u32x4;
__device__ __forceinline__ uint4 ldg16(const u16* p) {       // explicit global address space: never a flat load
    const u32x4 v = *(gptr_u32x4)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------
// fp32 NHWC [rows][C] -> bf16 (round to nearest even) + fp64 batch statistics of the ROUNDED values when
// ystats != null (they are what every consumer BatchNorm will normalise).  A thread owns 8 channels.
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, double* ystats,
                                                        long rows, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);           // [rpi][C][2]
    const int tid = threadIdx.x;
    const int g8 = C >> 3;
    const int rpi = 256 / g8;
    const int g = tid % g8;
    const int ry = tid / g8;
    const bool active = ry < rpi;
    double s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
    if (active) {
        const long stride = (long)gridDim.x * rpi;
        for (long row0 = (long)blockIdx.x * rpi + ry; row0 < rows; row0 += stride * 4) {      // four rows' loads in flight together
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long row = row0 + u * stride < rows ? row0 + u * stride : row0;
                va[u] = ldg4(src + (size_t)row * C + 8 * g);
                vb[u] = ldg4(src + (size_t)row * C + 8 * g + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long row = row0 + u * stride;
                if (row >= rows) break;
                const float4 a = va[u], b = vb[u];
                const uint4 q = make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w));
                *reinterpret_cast<uint4*>(dst + (size_t)row * C + 8 * g) = q;
                const float v[8] = {bf16_lo(q.x), bf16_hi(q.x), bf16_lo(q.y), bf16_hi(q.y), bf16_lo(q.z), bf16_hi(q.z), bf16_lo(q.w), bf16_hi(q.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += (double)v[e] * v[e]; }
            }
        }
    }
    if (ystats == nullptr) return;
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)ry * C + 8 * g + e) * 2 + 0] = s1[e];
            red[((size_t)ry * C + 8 * g + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0.0, b = 0.0;
        for (int r = 0; r < rpi; ++r) { a += red[((size_t)r * C + c) * 2 + 0]; b += red[((size_t)r * C + c) * 2 + 1]; }
        __hip_atomic_fetch_add(ystats + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ystats + C + c, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_cvt_bf16(const float* src, void* dst, double* ystats, long rows, int C, int num_cus, hipStream_t s) {
    const int g8 = C / 8;
    if (g8 < 1 || g8 > 256 || C % 8) return hipErrorInvalidValue;
    const int rpi = 256 / g8;
    long gx = (rows + rpi - 1) / rpi;
    if (gx > 2L * num_cus) gx = 2L * num_cus;      // (every block ends in 2 * C fp64 atomics on the same addresses)
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)gx), dim3(256), (size_t)rpi * C * 16, s, src, (u16*)dst, ystats, rows, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Weights torch [Cout][Cin][taps] fp32 -> bf16 B operand [tap][Kpad/8][Npad][8] (K = Cin, N = Cout), one launch
// for every conv (blockIdx.y); same element offsets (dstF) as the fp32 operand, in the bf16 arena.
__global__ __launch_bounds__(256) void repack_bf16_kernel(const RepackEntry* tab, const float* params, u16* arena, int with_backward) {
    const RepackEntry e = tab[blockIdx.y];
    const float* w = params + e.src;
    {
        const long total = (long)e.taps * e.KpadF * e.NpadF;
        u16* dst = arena + e.dstF;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total / 2; i += (long)gridDim.x * 256) {
            const long j = 2 * i;
            const int ee = (int)(j & 7);
            long r = j >> 3;
            const int n = (int)(r % e.NpadF); r /= e.NpadF;
            const int kq = (int)(r % (e.KpadF >> 3));
            const int t = (int)(r / (e.KpadF >> 3));
            const int k = 8 * kq + ee;
            const float a = (k < e.Cin && n < e.Cout) ? w[((size_t)n * e.Cin + k) * e.taps + t] : 0.f;
            const float b = (k + 1 < e.Cin && n < e.Cout) ? w[((size_t)n * e.Cin + k + 1) * e.taps + t] : 0.f;
            reinterpret_cast<unsigned*>(dst)[i] = pack_bf16(a, b);
        }
    }
    if (with_backward && e.dstB >= 0 && (e.KpadB & 7) == 0) {      // K = Cout, N = Cin, taps flipped (data gradient)
        const long total = (long)e.taps * e.KpadB * e.NpadB;
        u16* dst = arena + e.dstB;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total / 2; i += (long)gridDim.x * 256) {
            const long j = 2 * i;
            const int ee = (int)(j & 7);
            long r = j >> 3;
            const int n = (int)(r % e.NpadB); r /= e.NpadB;
            const int kq = (int)(r % (e.KpadB >> 3));
            const int t = (int)(r / (e.KpadB >> 3));
            const int k = 8 * kq + ee;                             // output channel of the forward conv
            const float a = (k < e.Cout && n < e.Cin) ? w[((size_t)k * e.Cin + n) * e.taps + (e.taps - 1 - t)] : 0.f;
            const float b = (k + 1 < e.Cout && n < e.Cin) ? w[((size_t)(k + 1) * e.Cin + n) * e.taps + (e.taps - 1 - t)] : 0.f;
            reinterpret_cast<unsigned*>(dst)[i] = pack_bf16(a, b);
        }
    }
}

hipError_t launch_repack_bf16(const RepackEntry* tab, int n, const float* params, void* arena, int with_backward, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(repack_bf16_kernel, dim3(16, n), dim3(256), 0, s, tab, params, (u16*)arena, with_backward);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 2x2/2 max-pool over bf16 NHWC (the max of bf16 values is exact) + fp64 batch statistics of the pooled tensor when
// ystats != null (training).  A thread owns 8 channels; 256 / (C/8) rows per block iteration.
constexpr int B16_POOL_U = 4;

__global__ __launch_bounds__(256) void pool_bf16_kernel(const u16* __restrict__ x, u16* __restrict__ y, double* ystats,
                                                         int N, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);           // [rpi][C][2]
    const int tid = threadIdx.x;
    const int g8 = C >> 3;
    const int rpi = 256 / g8;
    const int g = tid % g8;
    const int ry = tid / g8;
    const bool active = ry < rpi;
    const int Ho = H >> 1, Wo = W >> 1;
    const long rows = (long)N * Ho * Wo;
    double s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
    if (active) {
        // B16_POOL_U output rows per thread and iteration with all their loads requested up front; a few hundred fat blocks (see
        // pool_fwd_kernel): at 64 x 64 and batch 24 the thin version was 1536 blocks x 256 fp64 atomics on the same 256 addresses
        const long stride = (long)gridDim.x * rpi;
        for (long row0 = (long)blockIdx.x * rpi + ry; row0 < rows; row0 += stride * B16_POOL_U) {
          uint4 vv[B16_POOL_U][4];
#pragma unroll
          for (int u = 0; u < B16_POOL_U; ++u) {
            const long row = row0 + u * stride < rows ? row0 + u * stride : row0;
            const int ni = (int)(row / (Ho * Wo));
            const int rm = (int)(row - (long)ni * Ho * Wo);
            const int yo = rm / Wo, xo = rm - yo * Wo;
            const size_t m00 = ((size_t)ni * H + 2 * yo) * W + 2 * xo;
            vv[u][0] = ldg16(x + m00 * C + 8 * g);
            vv[u][1] = ldg16(x + (m00 + 1) * C + 8 * g);
            vv[u][2] = ldg16(x + (m00 + W) * C + 8 * g);
            vv[u][3] = ldg16(x + (m00 + W + 1) * C + 8 * g);
          }
#pragma unroll
          for (int u = 0; u < B16_POOL_U; ++u) {
            const long row = row0 + u * stride;
            if (row >= rows) break;
            float best[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 v = vv[u][k];
                const unsigned q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf16_lo(q[e]), hi = bf16_hi(q[e]);
                    best[2 * e] = k == 0 ? lo : fmaxf(best[2 * e], lo);
                    best[2 * e + 1] = k == 0 ? hi : fmaxf(best[2 * e + 1], hi);
                }
            }
            reinterpret_cast<uint4*>(y + (size_t)row * C + 8 * g)[0] =
                make_uint4(pack_bf16(best[0], best[1]), pack_bf16(best[2], best[3]), pack_bf16(best[4], best[5]), pack_bf16(best[6], best[7]));
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += best[e]; s2[e] += (double)best[e] * best[e]; }
          }
        }
    }
    if (ystats == nullptr) return;
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)ry * C + 8 * g + e) * 2 + 0] = s1[e];
            red[((size_t)ry * C + 8 * g + e) * 2 + 1] = s2[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double a = 0.0, b = 0.0;
        for (int r = 0; r < rpi; ++r) { a += red[((size_t)r * C + c) * 2 + 0]; b += red[((size_t)r * C + c) * 2 + 1]; }
        __hip_atomic_fetch_add(ystats + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ystats + C + c, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_pool_bf16(const void* x, void* y, double* ystats, int N, int H, int W, int C, int num_cus, hipStream_t s) {
    const int g8 = C / 8;
    if (g8 < 1 || g8 > 256 || C % 8) return hipErrorInvalidValue;
    const int rpi = 256 / g8;
    const long rows = (long)N * (H / 2) * (W / 2);
    long gx = (rows + (long)rpi * B16_POOL_U - 1) / ((long)rpi * B16_POOL_U);
    if (gx > 2L * num_cus) gx = 2L * num_cus;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pool_bf16_kernel, dim3((unsigned)gx), dim3(256), (size_t)rpi * C * 16, s, (const u16*)x, (u16*)y, ystats, N, H, W, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// [concat -> BatchNorm(eval) -> ReLU] -> 1x1 / 3x3 convolution, bf16 in, bf16 (OUTF32 = 0) or fp32 (heads) out.
// Seg::x / ConvArgs::y carry bf16 pointers here (typed float* in the shared argument struct); ld in elements.
// Requirements (checked by the launcher): every segment and K a multiple of 32 channels, M a multiple of 32 rows.
constexpr int B16_MAX_WAVES = 16;        // <= 128 VGPRs per lane (the widest variant uses 118)
// (four channel tiles: 64 accumulators -- at 128 registers the kernel spilled 5 (8 with nine taps), reloaded inside the tile loop; twelve
// waves = 168 registers.  At batch 24 the launcher gives those blocks twelve waves anyway: 3072 tiles on 256 one-block CUs.)
constexpr int b16_max_waves(int nt) { return nt == 4 ? 12 : B16_MAX_WAVES; }

template <int TAPS, int NT, int OUTF32>
__device__ __forceinline__ void conv_bf16_body(const ConvArgs& p, const int bidx, const int bidy, const int gdimx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = NT * 32;
    const int kq8 = p.Kpad >> 3;
    const int brows = TAPS * kq8;                     // 16-byte rows of B
    uint4* Bs = reinterpret_cast<uint4*>(smem);       // [TAPS * Kpad/8][NB]
    float* sc = reinterpret_cast<float*>(Bs + (size_t)brows * NB);
    float* sh = sc + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(sh + p.Ccat + (p.Ccat & 1));      // [NB][2] output statistics (training)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    int bx = bidx, by = bidy, gxd = gdimx;      // see conv_kernel: column slices of a row block on one XCD
    if (p.xcd_gx > 0) {
        const int L = bidx, xcd = L & 7, slot = L >> 3;
        by = slot % p.xcd_gy;
        bx = (slot / p.xcd_gy) * 8 + xcd;
        gxd = p.xcd_gx;
        if (bx >= gxd) return;
    }
    const int n0 = by * NB;
    const u16* wB = reinterpret_cast<const u16*>(p.wB);

    if (!CUNET_DBG(p, 32)) {   // B operand -> LDS (8 loads in flight per thread)
        const int total = brows * NB;
        for (int base = tid; base < total; base += blockDim.x * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                const int ic = idx < total ? idx : 0;
                const int row = ic / NB;
                const int n = ic - row * NB;
                const int nn = (n0 + n < p.Npad) ? n0 + n : 0;
                v[u] = ldg16(wB + ((size_t)row * p.Npad + nn) * 8);
                if (n0 + n >= p.Npad) v[u] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                if (idx < total) Bs[idx] = v[u];
            }
        }
    }
    if (CUNET_DBG(p, 64)) {    // (timing experiments: no BatchNorm table)
    } else if (p.training) {          // batch statistics of each segment (fp64 sums written by the producers' epilogues)
        for (int sgi = 0; sgi < p.nseg; ++sgi) {
            const Seg sg = p.seg[sgi];
            for (int lc = tid; lc < sg.C; lc += blockDim.x) {
                const int c = sg.choff + lc;
                const double mean = sg.stats[lc] / sg.count;
                double var = sg.stats[sg.C + lc] / sg.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const double scale = (double)p.gamma[c] / sqrt(var + (double)BN_EPS);
                sc[c] = (float)scale;
                sh[c] = (float)((double)p.beta[c] - mean * scale);
            }
        }
    } else {
        for (int c = tid; c < p.Ccat; c += blockDim.x) {   // running statistics
            const double scale = (double)p.gamma[c] / sqrt((double)p.rvar[c] + (double)BN_EPS);
            sc[c] = (float)scale;
            sh[c] = (float)((double)p.beta[c] - (double)p.rmean[c] * scale);
        }
    }
    for (int i = tid; i < NB * 2; i += blockDim.x) redbuf[i] = 0.0;
    __syncthreads();

    const int HW = p.H * p.W;
    const int nck = p.Kpad >> 5;                      // 32-channel chunks per tap
    const int ntiles = p.M >> 5;
    double dsum[NT], dsq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dsum[nt] = 0.0; dsq[nt] = 0.0; }

    for (int tile = bx * nwaves + wave; tile < ntiles; tile += gxd * nwaves) {
        const int m = tile * 32 + li;
        const int nimg = m / HW;
        const int rem = m - nimg * HW;
        const int py = rem / p.W;
        const int px = rem - py * p.W;
        const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        int sidx = 0, cl = 0, ncs = 0, tap = 0;
        const u16* rowptr = nullptr;
        bool tvalid = true;
        auto enter = [&]() {
            if (TAPS == 1) {
                const Seg sg = p.seg[sidx];
                rowptr = reinterpret_cast<const u16*>(sg.x) + (size_t)(sg.ups ? rowU : m) * sg.ld + 8 * hi;
                ncs = sg.C >> 5;
            } else {
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int yy = py + dy, xx = px + dx;
                tvalid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
                rowptr = reinterpret_cast<const u16*>(p.seg[0].x) + (size_t)(tvalid ? m + dy * p.W + dx : m) * p.seg[0].ld + 8 * hi;
                ncs = nck;
            }
            cl = 0;
        };
        const int nchunks = TAPS * nck;
        enter();
        uint4 anext[2];
        bool vcur = tvalid;
        anext[0] = ldg16(rowptr);
        anext[1] = ldg16(rowptr + 16);
        for (int ch = 0; ch < nchunks; ++ch) {
            uint4 acur[2] = {anext[0], anext[1]};
            const bool vthis = vcur;
            if (ch + 1 < nchunks) {
                if (++cl == ncs) { ++sidx; ++tap; enter(); }
                vcur = tvalid;
                anext[0] = ldg16(rowptr + cl * 32);
                anext[1] = ldg16(rowptr + cl * 32 + 16);
            }
            const int cc = (TAPS == 9) ? (ch % nck) : ch;                 // channel chunk inside the concat
            const uint4* bb = Bs + (size_t)ch * 4 * NB;                      // rows (ch*4 + 2s + hi)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // BN + ReLU on the 8 channels cc*32 + 16s + 8hi .. +7, back to bf16
                const float* scp = sc + cc * 32 + 16 * s + 8 * hi;
                const float* shp = sh + cc * 32 + 16 * s + 8 * hi;
                const float4 s0 = *reinterpret_cast<const float4*>(scp), s1 = *reinterpret_cast<const float4*>(scp + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(shp), h1 = *reinterpret_cast<const float4*>(shp + 4);
                const uint4 v = acur[s];
                uint4 a;
                a.x = pack_bf16(fmaxf(fmaf(bf16_lo(v.x), s0.x, h0.x), 0.f), fmaxf(fmaf(bf16_hi(v.x), s0.y, h0.y), 0.f));
                a.y = pack_bf16(fmaxf(fmaf(bf16_lo(v.y), s0.z, h0.z), 0.f), fmaxf(fmaf(bf16_hi(v.y), s0.w, h0.w), 0.f));
                a.z = pack_bf16(fmaxf(fmaf(bf16_lo(v.z), s1.x, h1.x), 0.f), fmaxf(fmaf(bf16_hi(v.z), s1.y, h1.y), 0.f));
                a.w = pack_bf16(fmaxf(fmaf(bf16_lo(v.w), s1.z, h1.z), 0.f), fmaxf(fmaf(bf16_hi(v.w), s1.w, h1.w), 0.f));
                if (TAPS == 9 && !vthis) a = make_uint4(0, 0, 0, 0);        // zero padding is post-activation
                const bf16x8 av = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, bb[(2 * s + hi) * NB + nt * 32 + li]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[nt], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int mrow0 = tile * 32;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            float s1 = 0.f, s2 = 0.f;
            if (OUTF32 && p.mse_tgt != nullptr) {           // heat-map head with the pixelwise MSE fused in (see conv_kernel)
                const float ginv = (float)(2.0 * p.mse_inv);
                const bool colok = col < p.Nout;
                const bool padcol = !colok && col < p.mse_ldd;
                float tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    tv[r] = ldg1(p.mse_tgt + (size_t)mm * p.ldy + (colok ? col : 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
            } else if (col < p.Nout && !(CUNET_DBG(p, 2048) && acc[nt][0] != 1234.5f)) {      // (2048: no output stores)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (OUTF32) {
                        p.y[(size_t)mm * p.ldy + col] = acc[nt][r];
                    } else {
                        const unsigned q = pack_bf16(acc[nt][r], 0.f);
                        reinterpret_cast<u16*>(p.y)[(size_t)mm * p.ldy + col] = (u16)(q & 0xffffu);
                        const float v = bf16_lo(q);                  // statistics of what the consumers will read
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
            }
            dsum[nt] += (double)s1;
            dsq[nt] += (double)s2;
        }
    }

    if (OUTF32 && p.mse_tgt != nullptr) {               // fused MSE: the block's sum of squared errors -> one fp64 atomic
        double t = 0.0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t += dsum[nt];
        for (int o = 32; o > 0; o >>= 1) t += shfl_xor_d(t, o);
        if (lane == 0) atomicAdd(&redbuf[0], t);
        __syncthreads();
        if (tid == 0) atomic_add_f64(p.mse_acc, redbuf[0] * p.mse_inv);
        return;
    }
    // ---- batch statistics of the output (training): lanes (l, l+32) -> waves through LDS -> one fp64 atomic per channel per block
    if (p.ystats != nullptr && !OUTF32) {
        double a1[NT], a2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {               // all lanes active here
            a1[nt] = dsum[nt] + shfl_xor_d16(dsum[nt]);
            a2[nt] = dsq[nt] + shfl_xor_d16(dsq[nt]);
        }
        if (hi == 0) {                                  // LDS fp64 atomics: one barrier instead of one per wave
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                atomicAdd(&redbuf[(nt * 32 + li) * 2 + 0], a1[nt]);
                atomicAdd(&redbuf[(nt * 32 + li) * 2 + 1], a2[nt]);
            }
        }
        __syncthreads();
        if (tid < NB) {
            const int col = n0 + tid;
            if (col < p.Nout) {
                __hip_atomic_fetch_add(p.ystats + col, redbuf[tid * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(p.ystats + p.Nout + col, redbuf[tid * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int TAPS, int NT, int OUTF32>
__global__ __launch_bounds__(b16_max_waves(NT) * 64) void conv_bf16_kernel(const ConvArgs p) {
    conv_bf16_body<TAPS, NT, OUTF32>(p, blockIdx.x, blockIdx.y, gridDim.x);
}
// two problems of one shape in one launch (the ahead / skip adapters: see conv_pair_kernel)
template <int TAPS, int NT, int OUTF32>
__global__ __launch_bounds__(b16_max_waves(NT) * 64) void conv_bf16_pair_kernel(const ConvPair q) {
    conv_bf16_body<TAPS, NT, OUTF32>(q.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Data gradient with bf16 MFMA (bf16 gradient-tensor storage, FusedTrainer(bf16_grads=True)):
//   dz[m][c] = relu'(BN(x)[m][c]) * sum_{tap, n} dY[m + tap][n] * W[n][c][flip(tap)]
// A = dY (bf16, no activation in front of it), B = the backward weight operand (bf16, [tap][Cout/8][Cin][8], taps
// flipped), accumulators fp32.  Epilogue as EP_BWD of conv_kernel: x (bf16, read through the node's segment table,
// up-sample map included) gives the ReLU mask and x-hat, dz is stored bf16, sum(dz) and sum(dz * xhat) go to `ystats`
// (= d beta, d gamma) in fp64.  Requirements: K (= Cout) and every segment multiples of 32, M of 32, W of 4.
struct Grp16 { const u16* ptr; int ld; int ups; };

// FZ (round 6, 1x1 over K = 128 only): the A operand is assembled on the load from the ONE consumer's dz slice and the tensor itself
// (ConvArgs::fz_*: the BatchNorm-backward gather of a single-consumer tensor folded into the data gradient that reads it) -- the
// coefficients come from a table built in the prologue, the result is rounded to bf16 exactly where grad_gather_rows_kernel<1, 0, 2, 8>
// would have stored it, and the blocks of column slice 0 write it to the gradient tensor as they go.
template <int TAPS, int NT, int NCK = 0, bool FZ = false>       // NCK: 1x1 only, K / 32 as a compile-time constant (1 ... 4)
__device__ __forceinline__ void dgrad_bf16_body(const ConvArgs& p, const int bidx, const int bidy, const int gdimx) {
    static_assert(!FZ || (TAPS == 1 && NCK == 4), "the fused gather exists for the 1x1 data gradient over K = 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = NT * 32;
    const int kq8 = p.Kpad >> 3;
    const int brows = TAPS * kq8;
    uint4* Bs = reinterpret_cast<uint4*>(smem);                      // [TAPS * K/8][NB]
    float* sc = reinterpret_cast<float*>(Bs + (size_t)brows * NB);    // [Ccat] each
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    Grp16* grp = reinterpret_cast<Grp16*>(is + p.Ccat);               // [Ccat / 4]
    double* redbuf = reinterpret_cast<double*>(grp + (p.Ccat >> 2));  // [NB][2]
    constexpr int TP = NB + 8;                                        // element pitch of the epilogue tiles (16-byte aligned rows)
    float* fzE = reinterpret_cast<float*>(redbuf + NB * 2);           // FZ: [3][128] coefficient tables E, D, A
    u16* tileT = reinterpret_cast<u16*>(fzE + (FZ ? 3 * 128 : 0));    // [waves][32][TP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the tile loop runs on scalar branches
    const int nwaves = blockDim.x >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    int bx = bidx, by = bidy, gxd = gdimx;      // see conv_kernel: column slices of a row block on one XCD
    if (p.xcd_gx > 0) {
        const int L = bidx, xcd = L & 7, slot = L >> 3;
        by = slot % p.xcd_gy;
        bx = (slot / p.xcd_gy) * 8 + xcd;
        gxd = p.xcd_gx;
        if (bx >= gxd) return;
    }
    const int n0 = by * NB;
    const u16* wB = reinterpret_cast<const u16*>(p.wB);
    const u16* dY = reinterpret_cast<const u16*>(p.a);

    {   // B operand -> LDS
        const int total = brows * NB;
        for (int base = tid; base < total; base += blockDim.x * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                const int ic = idx < total ? idx : 0;
                const int row = ic / NB;
                const int n = ic - row * NB;
                const int nn = (n0 + n < p.Npad) ? n0 + n : 0;
                v[u] = ldg16(wB + ((size_t)row * p.Npad + nn) * 8);
                if (n0 + n >= p.Npad) v[u] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * blockDim.x;
                if (idx < total) Bs[idx] = v[u];
            }
        }
    }
    for (int sgi = 0; sgi < p.nseg; ++sgi) {           // BN tables of the concat (batch statistics) and the segment table
        const Seg sg = p.seg[sgi];
        for (int lc = tid; lc < sg.C; lc += blockDim.x) {
            const int c = sg.choff + lc;
            const double mean = sg.stats[lc] / sg.count;
            double var = sg.stats[sg.C + lc] / sg.count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const double scale = (double)p.gamma[c] * istd;
            sc[c] = (float)scale;
            sh[c] = (float)((double)p.beta[c] - mean * scale);
            mu[c] = (float)mean;
            is[c] = (float)istd;
        }
        for (int g = tid; g < (sg.C >> 2); g += blockDim.x) {
            Grp16 e;
            e.ptr = reinterpret_cast<const u16*>(sg.x) + 4 * g;
            e.ld = sg.ld;
            e.ups = sg.ups;
            grp[(sg.choff >> 2) + g] = e;
        }
    }
    for (int i = tid; i < NB * 2; i += blockDim.x) redbuf[i] = 0.0;
    if constexpr (FZ) {      // the gather's coefficients, computed as grad_gather_rows_kernel<1, 0> computes them (one plain consumer)
        const double invM = 1.0 / (double)p.M;
        for (int c = tid; c < 128; c += blockDim.x) {
            const double mean = p.fz_stats[c] / p.fz_count;
            double var = p.fz_stats[128 + c] / p.fz_count - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double istd = 1.0 / sqrt(var + (double)BN_EPS);
            const int cc = p.fz_choff + c;
            const double scale = (double)p.fz_gamma[cc] * istd;
            const double c1 = p.fz_red[cc] * invM;
            const double c2 = p.fz_red[p.fz_lddz + cc] * invM;
            const double D = scale * c2 * istd;
            double Es = 0.0, Ds = 0.0;
            Es += 1.0 * (D * mean - scale * c1);
            Ds += 1.0 * D;
            fzE[c] = (float)Es;
            fzE[128 + c] = (float)Ds;
            fzE[256 + c] = (float)scale;
        }
    }
    __syncthreads();

    const int HW = p.H * p.W;
    const int nck = p.Kpad >> 5;
    const int nchunks = TAPS * nck;
    const int ntiles = p.M >> 5;
    double dsum[NT], dsq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dsum[nt] = 0.0; dsq[nt] = 0.0; }

    // Loader state of the tile whose A operand is being requested: the next tile's first chunk goes out BEFORE the current tile's
    // epilogue stores (vmcnt orders a later load behind them; see conv_kernel).
    int m = 0, py = 0, px = 0;
    auto set_tile = [&](int t) {
        m = t * 32 + li;
        if (TAPS == 9) {
            if (p.wshift >= 0) {
                const int rem = m & (HW - 1);
                py = rem >> p.wshift;
                px = rem & (p.W - 1);
            } else {
                const int nimg = m / HW;
                const int rem = m - nimg * HW;
                py = rem / p.W;
                px = rem - py * p.W;
            }
        }
    };
    int cl = 0, tap = 0;
    const u16* rowptr = nullptr;
    bool tvalid = true;
    auto enter = [&]() {
        if (TAPS == 1) {
            rowptr = dY + (size_t)m * p.lda + 8 * hi;
        } else {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int yy = py + dy, xx = px + dx;
            tvalid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
            rowptr = dY + (size_t)(tvalid ? m + dy * p.W + dx : m) * p.lda + 8 * hi;
        }
        cl = 0;
    };
    uint4 anext[2];
    bool vcur = true;
    auto begin_tile = [&](int t) {
        set_tile(t);
        tap = 0;
        enter();
        vcur = tvalid;
        anext[0] = ldg16(rowptr);
        anext[1] = ldg16(rowptr + 16);
    };
    // 1x1 (K <= 128): the waves of this kernel spend two thirds of their cycles waiting for their own loads (SQ_WAIT_ANY, round-3
    // counters) -- a one-chunk look-ahead keeps 2 KB per wave in flight.  Here ALL pieces of a tile's dY rows (<= 8 per lane, 8 KB per
    // wave) are requested together, and the next tile's right behind the epilogue's x requests.
    // K / 32 is a template constant there: with a run-time chunk count the requests sit behind branches, the compiler loses count
    // of what is outstanding and waits for vmcnt(0) -- i.e. for the look-ahead it has just issued -- before the epilogue's x.
    constexpr bool FULLA = (TAPS == 1);
    static_assert(!FULLA || (NCK >= 1 && NCK <= 4), "1x1: K / 32 in 1 ... 4");
    constexpr int NA = FULLA ? 2 * NCK : 1;
    uint4 abuf[NA];
    uint4 zbuf[FZ ? NA : 1];                          // FZ: the tensor's own pieces next to the consumer's dz pieces in abuf
    // the epilogue's x pieces (see below): this lane's piece column is the same for every tile
    constexpr int PPR = NB / 8;                       // pieces per tile row
    constexpr int NPJ = (32 * PPR) / 64;              // pieces per lane (4 for NB = 64, 2 for NB = 32)
    const int pc8 = lane % PPR;                       // this lane's piece column ...
    const int pr0 = lane / PPR;                       // ... and first row; further rows every 64 / PPR
    const int pcol = n0 + 8 * pc8;
    const bool pok = pcol < p.Nout;
    Grp16 pg;
    pg.ptr = dY; pg.ld = 0; pg.ups = 0;
    if (pok) pg = grp[pcol >> 2];
    uint4 xp[NPJ];
    auto request_x = [&](int t) {
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const int mm = t * 32 + pr0 + j * (64 / PPR);
            int row = mm;
            if (p.any_ups && pg.ups) {
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
            xp[j] = ldg16(pg.ptr + (size_t)row * pg.ld);
        }
    };
    // 1x1: a tile's dY AND its x pieces go out together, after the previous tile's dz stores.  One round trip per tile (a wave
    // with one tile -- every launch at 32 x 32 and below -- used to make two), the stores complete under it, and the request count
    // is the same on the loop's entry and back edge, so the waits are exact (vmcnt is one in-order counter of loads and stores).
    auto request_tile = [&](int t) {
        set_tile(t);
        const u16* rp = FZ ? reinterpret_cast<const u16*>(p.fz_dz) + (size_t)m * p.fz_lddz + p.fz_choff + 8 * hi : dY + (size_t)m * p.lda + 8 * hi;
#pragma unroll
        for (int c = 0; c < (FULLA ? NCK : 0); ++c) {
            abuf[(2 * c) % NA] = ldg16(rp + c * 32);
            abuf[(2 * c + 1) % NA] = ldg16(rp + c * 32 + 16);
        }
        if constexpr (FZ) {
            const u16* zp = reinterpret_cast<const u16*>(p.fz_x) + (size_t)m * p.fz_ldx + 8 * hi;
#pragma unroll
            for (int c = 0; c < NCK; ++c) {
                zbuf[2 * c] = ldg16(zp + c * 32);
                zbuf[2 * c + 1] = ldg16(zp + c * 32 + 16);
            }
        }
        request_x(t);
    };
    // FZ: pieces -> the gathered gradient, in place: r = fma(A, dz, E - D * x) (+ 0: the gather adds its zero "accumulate" operand),
    // rounded to bf16 as the gather's store rounds; slice 0 keeps the tensor
    auto assemble = [&](int t) {
        if constexpr (FZ) {
            u16* outp = reinterpret_cast<u16*>(const_cast<float*>(p.a)) + (size_t)(t * 32 + li) * p.lda + 8 * hi;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int k0 = (i >> 1) * 32 + (i & 1) * 16 + 8 * hi;
                const unsigned dw[4] = {abuf[i].x, abuf[i].y, abuf[i].z, abuf[i].w};
                const unsigned zw[4] = {zbuf[i].x, zbuf[i].y, zbuf[i].z, zbuf[i].w};
                unsigned ow[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    float r[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = k0 + 2 * h + e;
                        const float dzv = e ? bf16_bits_hi(dw[h]) : bf16_bits_lo(dw[h]);
                        const float xv = e ? bf16_bits_hi(zw[h]) : bf16_bits_lo(zw[h]);
                        float v = fzE[k] - fzE[128 + k] * xv;
                        v = fmaf(fzE[256 + k], dzv, v);
                        v += 0.f;
                        r[e] = v;
                    }
                    ow[h] = (unsigned)f32_to_bf16_rne(r[0]) | ((unsigned)f32_to_bf16_rne(r[1]) << 16);
                }
                abuf[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (by == 0) *reinterpret_cast<uint4*>(outp + (i >> 1) * 32 + (i & 1) * 16) = abuf[i];
            }
        }
    };
    const int tstride = gxd * nwaves;
    int tile = bx * nwaves + wave;
    if (tile < ntiles) {
        if constexpr (FULLA) request_tile(tile);
        else begin_tile(tile);
    }
    while (tile < ntiles) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        if constexpr (FULLA) {
            assemble(tile);
#pragma unroll
            for (int c = 0; c < NCK; ++c) {
                const uint4* bb = Bs + (size_t)c * 4 * NB;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, abuf[(2 * c + s) % NA]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 bv = __builtin_bit_cast(bf16x8, bb[(2 * s + hi) * NB + nt * 32 + li]);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[nt], 0, 0, 0);
                    }
                }
            }
        } else
        for (int ch = 0; ch < nchunks; ++ch) {
            uint4 acur[2] = {anext[0], anext[1]};
            const bool vthis = vcur;
            if (ch + 1 < nchunks) {
                if (++cl == nck) { ++tap; enter(); }
                vcur = tvalid;
                anext[0] = ldg16(rowptr + cl * 32);
                anext[1] = ldg16(rowptr + cl * 32 + 16);
            }
            const uint4* bb = Bs + (size_t)ch * 4 * NB;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 a = acur[s];
                if (TAPS == 9 && !vthis) a = make_uint4(0, 0, 0, 0);
                const bf16x8 av = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, bb[(2 * s + hi) * NB + nt * 32 + li]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[nt], 0, 0, 0);
                }
            }
        }

        // ---- epilogue.  x (for the ReLU mask and x-hat) and dz move through a wave-private LDS tile T[32][NB + 8] so that global
        //      memory sees 16-byte pieces: a lane requests four pieces (8 channels of one row each: 8 lanes cover 128 contiguous
        //      bytes of a row) instead of 16 x NT two-byte loads, the tile is read back in the accumulator layout (col = lane & 31,
        //      rows (r&3) + 8*(r>>2) + 4*hi), dz is written over the x it came from, and leaves as 16-byte pieces again.  The
        //      two-byte version issued 64 memory instructions per lane and tile next to 16 MFMAs; LDS takes the narrow ones.
        const int mrow0 = tile * 32;
        u16* T = tileT + (size_t)wave * 32 * TP;
        if constexpr (!FULLA) {
            request_x(tile);
            if (tile + tstride < ntiles) begin_tile(tile + tstride);      // next tile's first A chunk: behind the x requests, ahead of the stores
        }
#pragma unroll
        for (int j = 0; j < NPJ; ++j)
            *reinterpret_cast<uint4*>(T + (size_t)(pr0 + j * (64 / PPR)) * TP + 8 * pc8) = xp[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            const bool colok = col < p.Nout;
            float csc = 0.f, csh = 0.f, cmu = 0.f, cis = 0.f;
            if (colok) { csc = sc[col]; csh = sh[col]; cmu = mu[col]; cis = is[col]; }
            float s1 = 0.f, s2 = 0.f;
            u16* tcol = T + nt * 32 + li;
            // (three passes: a read next to a write of the same array is ordered by the compiler, which made this 16 dependent
            // LDS round trips per column tile)
            float xv[16];
            u16 dq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = bf16_lo((unsigned)tcol[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * TP]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(xv[r], csc, csh);
                const float dz = (colok && z > 0.f) ? acc[nt][r] : 0.f;
                dq[r] = (u16)(pack_bf16(dz, 0.f) & 0xffffu);
                s1 += dz;
                s2 = fmaf(dz, (xv[r] - cmu) * cis, s2);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) tcol[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * TP] = dq[r];
            dsum[nt] += (double)s1;
            dsq[nt] += (double)s2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (pok) {
#pragma unroll
            for (int j = 0; j < NPJ; ++j) {
                const int rr = pr0 + j * (64 / PPR);
                const uint4 q = *reinterpret_cast<const uint4*>(T + (size_t)rr * TP + 8 * pc8);
                *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.y) + (size_t)(mrow0 + rr) * p.ldy + pcol) = q;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next tile's pieces overwrite T)
        __builtin_amdgcn_wave_barrier();
        tile += tstride;
        if constexpr (FULLA) {
            if (tile < ntiles) request_tile(tile);
        }
    }

    if (p.ystats != nullptr) {
        double a1[NT], a2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            a1[nt] = dsum[nt] + shfl_xor_d16(dsum[nt]);
            a2[nt] = dsq[nt] + shfl_xor_d16(dsq[nt]);
        }
        if (hi == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                atomicAdd(&redbuf[(nt * 32 + li) * 2 + 0], a1[nt]);
                atomicAdd(&redbuf[(nt * 32 + li) * 2 + 1], a2[nt]);
            }
        }
        __syncthreads();
        if (tid < NB) {
            const int col = n0 + tid;
            if (col < p.Nout) {
                __hip_atomic_fetch_add(p.ystats + col, redbuf[tid * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(p.ystats + p.Nout + col, redbuf[tid * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// waves per block: 16 (128 VGPRs) except the two-channel-tile 1x1 variant, whose whole-tile dY look-ahead needs the 168 of 12 waves
// (FZ, the gather folded into the load: 8 more 16-byte pieces per lane -- one wave per SIMD fewer)
constexpr int dg16_max_waves(int taps, int nt, bool fz = false) { return fz ? (nt == 2 ? 8 : 12) : ((taps == 1 && nt == 2) ? 12 : B16_MAX_WAVES); }

template <int TAPS, int NT, int NCK = 0, bool FZ = false>
__global__ __launch_bounds__(dg16_max_waves(TAPS, NT, FZ) * 64) void dgrad_bf16_kernel(const ConvArgs p) {
    dgrad_bf16_body<TAPS, NT, NCK, FZ>(p, blockIdx.x, blockIdx.y, gridDim.x);
}
template <int TAPS, int NT, int NCK = 0>
__global__ __launch_bounds__(dg16_max_waves(TAPS, NT) * 64) void dgrad_bf16_pair_kernel(const ConvPair q) {
    dgrad_bf16_body<TAPS, NT, NCK>(q.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// 3x3 forward, tap-split, bf16 (norm2 -> relu2 -> conv2, 128 -> 32 channels; the bf16 twin of conv3x3_tapsplit_kernel).
// conv_bf16_kernel<9> walks 9 taps x 4 chunks in one wave: 36 dependent load -> MFMA steps, 19-25 us at every level from 4x4 to
// 32x32 (rocprofv3, round 3: 1.2 ms of a CU-Net-8 step).  Here a block is 9 waves and wave t owns tap t: its K x 32 slice of the
// weights lives in 32 registers, its lane requests the eight 16-byte pieces of its shifted row together, BatchNorm + ReLU in
// fp32, re-rounded, 8 MFMAs 32x32x16; the nine partial tiles meet in LDS where 512 threads add them, round to bf16, store and
// keep the statistics of the ROUNDED values.  The next tile's loads are in flight across the reduction.
__global__ __launch_bounds__(576) void conv3x3_tapsplit_bf16_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 128;
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
    const u16* xin = reinterpret_cast<const u16*>(sg.x);
    const u16* wB = reinterpret_cast<const u16*>(p.wB);

    // this wave's tap of the packed weights [tap][K/8][Npad][8]: piece (kq = 2j + hi, n = li) feeds MFMA j
    uint4 bw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bw[j] = ldg16(wB + ((size_t)(tap * (K / 8) + 2 * j + hi) * p.Npad + li) * 8);

    const int HW = p.H * p.W;
    const int ntiles = p.M >> 5;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    uint4 a[8];
    bool valid = false;
    auto fetch = [&](int tile) {                      // raw loads of this wave's shifted rows (always a valid address)
        const int m = tile * 32 + li;
        const int nimg = m / HW;
        const int rem = m - nimg * HW;
        const int py = rem / p.W;
        const int px = rem - py * p.W;
        const int yy = py + dy, xx = px + dx;
        valid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
        const u16* src = xin + (size_t)(valid ? m + dy * p.W + dx : m) * sg.ld + 8 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = ldg16(src + 16 * j);
    };
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);                   // requested before the BatchNorm tables are built

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
    __syncthreads();

    double dsum = 0.0, dsq = 0.0;
    for (; tile < ntiles; tile += gridDim.x) {        // block-uniform trip count
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool v = valid;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* scp = sc + 16 * j + 8 * hi;
            const float* shp = sh + 16 * j + 8 * hi;
            const float4 s0 = *reinterpret_cast<const float4*>(scp), s1 = *reinterpret_cast<const float4*>(scp + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(shp), h1 = *reinterpret_cast<const float4*>(shp + 4);
            const uint4 x = a[j];
            uint4 t;
            t.x = pack_bf16(fmaxf(fmaf(bf16_lo(x.x), s0.x, h0.x), 0.f), fmaxf(fmaf(bf16_hi(x.x), s0.y, h0.y), 0.f));
            t.y = pack_bf16(fmaxf(fmaf(bf16_lo(x.y), s0.z, h0.z), 0.f), fmaxf(fmaf(bf16_hi(x.y), s0.w, h0.w), 0.f));
            t.z = pack_bf16(fmaxf(fmaf(bf16_lo(x.z), s1.x, h1.x), 0.f), fmaxf(fmaf(bf16_hi(x.z), s1.y, h1.y), 0.f));
            t.w = pack_bf16(fmaxf(fmaf(bf16_lo(x.w), s1.z, h1.z), 0.f), fmaxf(fmaf(bf16_hi(x.w), s1.w, h1.w), 0.f));
            if (!v) t = make_uint4(0, 0, 0, 0);                         // zero padding is post-activation
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t), __builtin_bit_cast(bf16x8, bw[j]), acc, 0, 0, 0);
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
                const unsigned q = pack_bf16(vsum, 0.f);
                reinterpret_cast<u16*>(p.y)[(size_t)(tile * 32 + row) * p.ldy + li] = (u16)(q & 0xffffu);
                const float vr = bf16_lo(q);          // statistics of what the consumers will read
                dsum += (double)vr;
                dsq += (double)vr * (double)vr;
            }
        }
        __syncthreads();
    }
    if (p.ystats != nullptr) {                        // (threads 512..575 carry zeros)
        const double a1 = dsum + shfl_xor_d16(dsum);
        const double b1 = dsq + shfl_xor_d16(dsq);
        if (hi == 0) {
            atomicAdd(&redbuf[li * 2 + 0], a1);
            atomicAdd(&redbuf[li * 2 + 1], b1);
        }
        __syncthreads();
        if (tid < 32 && tid < p.Nout) {
            __hip_atomic_fetch_add(p.ystats + tid, redbuf[tid * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 forward on a ring of activated image rows, bf16 (128 -> 32 channels at 64 x 64 and 32 x 32; the bf16 twin of
// conv3x3_ring_kernel).  Both other bf16 kernels read every input row through nine shifted taps from L2 -- 226 MB of L2 -> CU traffic
// per 64 x 64 launch at batch 24 against 25 MB of input: 43 us where the bytes need 6.  Here a 4-wave workgroup walks image rows in
// steps of RB = 4 tiles' worth (2 rows of 64 pixels / 4 rows of 32): every row enters LDS ONCE -- BatchNorm + ReLU in fp32 and
// re-rounded on the way in, pixel pitch 136 bf16 (272 bytes: conflict-free 16-byte fragment reads), zero border columns -- and stays
// for the three output rows that need it.  The whole weight operand (72 KB) sits in LDS as well, so a wave computes a complete
// 32-pixel x 32-channel tile by itself: 72 MFMAs 32x32x16 whose A fragments are ds_read_b128 of row (y + dy) at pixel (x + dx) --
// no cross-wave reduction.  The next step's rows are requested from HBM before this step's MFMAs and committed after them.
constexpr int R16_PITCH = 136;          // bf16 elements per pixel in the ring

template <int WT>                        // W = 32 * WT pixels per image row (WT = 1, 2)
__global__ __launch_bounds__(256) void conv3x3_ring_bf16_kernel(const ConvArgs p, int steps_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 128;
    constexpr int W = 32 * WT;
    constexpr int RB = 4 / WT;                                    // output rows per step (one tile per wave)
    constexpr int S = RB + 2;                                     // ring slots
    constexpr int SLOT = (W + 2) * R16_PITCH;                     // elements per slot
    uint4* Bs = reinterpret_cast<uint4*>(smem);                   // [9][16][32] 16-byte pieces
    float* sc = reinterpret_cast<float*>(Bs + 9 * 16 * 32);       // [K]
    float* sh = sc + K;
    double* redbuf = reinterpret_cast<double*>(sh + K);           // [32][2]
    u16* ring = reinterpret_cast<u16*>(redbuf + 64);              // S slots
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const Seg sg = p.seg[0];
    const u16* xin = reinterpret_cast<const u16*>(sg.x);
    const u16* wB = reinterpret_cast<const u16*>(p.wB);
    const int H = p.H;
    const int NH = p.M / W;                                       // image rows in the batch

    {   // weights [tap][K/8][Npad = 32][8] -> LDS, 18 pieces per thread, all requested before the first is stored
        uint4 v[18];
#pragma unroll
        for (int u = 0; u < 18; ++u) v[u] = ldg16(wB + (size_t)(tid + 256 * u) * 8);
#pragma unroll
        for (int u = 0; u < 18; ++u) Bs[tid + 256 * u] = v[u];
    }
    for (int c = tid; c < K; c += 256) {
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
    for (int i = tid; i < S * 2 * 17; i += 256) {                 // the two border pixels of every slot stay zero (17 pieces of 16 bytes each)
        const int slot = i / 34, r = i - slot * 34;
        const int pix = r < 17 ? 0 : W + 1;
        reinterpret_cast<uint4*>(ring + (size_t)slot * SLOT + (size_t)pix * R16_PITCH)[r % 17] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // staging plan: a row is W * 16 pieces of 8 channels; thread t takes pieces t, t + 256, ...: its 8 channels never change
    const int c8 = tid & 15;
    float s8[8], h8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s8[e] = sc[8 * c8 + e]; h8[e] = sh[8 * c8 + e]; }
    constexpr int PPR = W * 16 / 256;                             // pieces per thread and row (4 / 2)
    constexpr int NLD = PPR * RB;                                 // = 8 loads per thread and step
    uint4 xv[NLD];
    auto issue_rows = [&](int g0, int nrows) {                    // rows g0 .. g0 + nrows - 1 (nrows <= RB); rows outside the batch are skipped
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < PPR; ++j) {
                const int g = g0 + r;
                const bool ok = r < nrows && g >= 0 && g < NH;
                const int pix = (tid >> 4) + 16 * j;
                xv[r * PPR + j] = ldg16(xin + ((size_t)(ok ? g : 0) * W + pix) * sg.ld + 8 * c8);
            }
    };
    auto commit_rows = [&](int g0, int nrows) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int g = g0 + r;
            if (!(r < nrows && g >= 0 && g < NH)) continue;       // (block-uniform)
            u16* slot = ring + (size_t)(((g % S) + S) % S) * SLOT;
#pragma unroll
            for (int j = 0; j < PPR; ++j) {
                const uint4 x = xv[r * PPR + j];
                uint4 t;
                t.x = pack_bf16(fmaxf(fmaf(bf16_lo(x.x), s8[0], h8[0]), 0.f), fmaxf(fmaf(bf16_hi(x.x), s8[1], h8[1]), 0.f));
                t.y = pack_bf16(fmaxf(fmaf(bf16_lo(x.y), s8[2], h8[2]), 0.f), fmaxf(fmaf(bf16_hi(x.y), s8[3], h8[3]), 0.f));
                t.z = pack_bf16(fmaxf(fmaf(bf16_lo(x.z), s8[4], h8[4]), 0.f), fmaxf(fmaf(bf16_hi(x.z), s8[5], h8[5]), 0.f));
                t.w = pack_bf16(fmaxf(fmaf(bf16_lo(x.w), s8[6], h8[6]), 0.f), fmaxf(fmaf(bf16_hi(x.w), s8[7], h8[7]), 0.f));
                const int pix = (tid >> 4) + 16 * j;
                *reinterpret_cast<uint4*>(slot + (size_t)(pix + 1) * R16_PITCH + 8 * c8) = t;
            }
        }
    };

    const int g_begin = blockIdx.x * steps_per_wg * RB;
    int g_end = g_begin + steps_per_wg * RB;
    if (g_end > NH) g_end = NH;
    // the first step needs rows g_begin - 1 .. g_begin + RB: two staging rounds
    issue_rows(g_begin - 1, 2); commit_rows(g_begin - 1, 2);
    issue_rows(g_begin + 1, RB); commit_rows(g_begin + 1, RB);
    __syncthreads();

    const int trow = WT == 2 ? (wave >> 1) : wave;                // this wave's output row within the step ...
    const int tcol = WT == 2 ? (wave & 1) * 32 : 0;               // ... and first pixel
    double dsum = 0.0, dsq = 0.0;
    for (int g0 = g_begin; g0 < g_end; g0 += RB) {
        const bool more = g0 + RB < g_end;
        if (more) issue_rows(g0 + RB + 1, RB);                    // rows g0 + RB + 1 .. g0 + 2 RB: in flight across this step's MFMAs
        const int g = g0 + trow;
        if (g < g_end) {                                          // (wave-uniform)
            const int y = g % H;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                if (y + dy < 0 || y + dy >= H) continue;          // zero padding is post-activation: the tap contributes nothing
                const u16* srow = ring + (size_t)(((g + dy) % S + S) % S) * SLOT;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int tap = (dy + 1) * 3 + (dx + 1);
                    const u16* ap = srow + (size_t)(tcol + li + dx + 1) * R16_PITCH + 8 * hi;
                    const uint4* bp = Bs + (size_t)(tap * 16 + hi) * 32 + li;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint4 a = *reinterpret_cast<const uint4*>(ap + 16 * j);
                        const uint4 b = bp[(size_t)(2 * j) * 32];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                    }
                }
            }
            // C layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            u16* yo = reinterpret_cast<u16*>(p.y) + ((size_t)g * W + tcol) * p.ldy + li;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const unsigned q = pack_bf16(acc[r], 0.f);
                yo[(size_t)row * p.ldy] = (u16)(q & 0xffffu);
                const float vr = bf16_lo(q);                      // statistics of what the consumers will read
                s1 += vr;
                s2 = fmaf(vr, vr, s2);
            }
            dsum += (double)s1;
            dsq += (double)s2;
        }
        __syncthreads();                                          // everyone is done with the rows this step no longer shares
        if (more) commit_rows(g0 + RB + 1, RB);
        __syncthreads();
    }
    if (p.ystats != nullptr) {
        const double a1 = dsum + shfl_xor_d16(dsum);
        const double b1 = dsq + shfl_xor_d16(dsq);
        if (hi == 0) {
            atomicAdd(&redbuf[li * 2 + 0], a1);
            atomicAdd(&redbuf[li * 2 + 1], b1);
        }
        __syncthreads();
        if (tid < 32 && tid < p.Nout) {
            __hip_atomic_fetch_add(p.ystats + tid, redbuf[tid * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ystats + p.Nout + tid, redbuf[tid * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static bool conv3x3_ring_bf16_supported(const ConvArgs& a, int out_f32) {
    return !out_f32 && a.taps == 9 && a.nseg == 1 && a.K == 128 && a.Kpad == 128 && a.Nout == 32 && a.Npad == 32 && !a.seg[0].ups &&
           a.seg[0].C == 128 && a.seg[0].ld % 8 == 0 && (a.W == 64 || a.W == 32) && a.M % a.W == 0 && a.ldy >= 32;
}

static hipError_t launch_conv3x3_ring_bf16(const ConvArgs& a, int num_cus, hipStream_t s) {
    const int WT = a.W / 32, RB = 4 / WT;
    const int NH = a.M / a.W;
    const int nsteps = (NH + RB - 1) / RB;
    int spw = (nsteps + num_cus - 1) / num_cus;                    // steps per workgroup: one workgroup per CU
    if (spw < 1) spw = 1;
    const int grid = (nsteps + spw - 1) / spw;
    const size_t smem = (size_t)9 * 16 * 32 * 16 + (size_t)2 * 128 * 4 + 64 * 8 + (size_t)(RB + 2) * (a.W + 2) * R16_PITCH * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (WT == 2) hipLaunchKernelGGL(conv3x3_ring_bf16_kernel<2>, dim3(grid), dim3(256), smem, s, a, spw);
    else hipLaunchKernelGGL(conv3x3_ring_bf16_kernel<1>, dim3(grid), dim3(256), smem, s, a, spw);
    return hipGetLastError();
}

static hipError_t launch_conv3x3_tapsplit_bf16(const ConvArgs& a, int num_cus, hipStream_t s) {
    const int ntiles = a.M / 32;
    static const int bpc = tune_int("CUNET_B16_TS_BPC", 2);         // blocks per CU (38 KB of LDS, 9 waves each)
    const int grid = ntiles < bpc * num_cus ? ntiles : bpc * num_cus;
    const size_t smem = (size_t)9 * 1024 * 4 + (size_t)128 * 8 + 64 * 8;
    hipLaunchKernelGGL(conv3x3_tapsplit_bf16_kernel, dim3(grid), dim3(576), smem, s, a);
    return hipGetLastError();
}

static size_t dgrad_bf16_smem(int NT, int taps, int Kpad, int Ccat) {
    return (size_t)taps * (Kpad / 8) * NT * 32 * 16 + (size_t)Ccat * 16 + (size_t)(Ccat / 4) * sizeof(Grp16) + (size_t)NT * 32 * 16 + 16;
}

template <int TAPS, int NT, int NCK = 0>
static hipError_t launch_dg16_pair_inst(const ConvArgs& a, const ConvArgs& b, dim3 grid, int threads, size_t smem, hipStream_t s) {
    if constexpr (TAPS == 1) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_bf16_pair_kernel<TAPS, NT, NCK>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        ConvPair q;
        q.a[0] = a;
        q.a[1] = b;
        copy_launch_geometry(q.a[1], a);
        grid.z = 2;
        hipLaunchKernelGGL((dgrad_bf16_pair_kernel<TAPS, NT, NCK>), grid, dim3(threads), smem, s, q);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int TAPS, int NT, int NCK = 0, bool FZ = false>
static hipError_t launch_dg16_inst(const ConvArgs& a, dim3 grid, int threads, size_t smem, hipStream_t s, const ConvArgs* pb = nullptr) {
    if (pb) {
        if constexpr (FZ) return hipErrorNotSupported;
        else return launch_dg16_pair_inst<TAPS, NT, NCK>(a, *pb, grid, threads, smem, s);
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_bf16_kernel<TAPS, NT, NCK, FZ>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((dgrad_bf16_kernel<TAPS, NT, NCK, FZ>), grid, dim3(threads), smem, s, a);
    return hipGetLastError();
}

// a.a (dY), a.seg[*].x, a.wB (backward operand), a.y (dz): bf16 behind float-typed pointers.  hipErrorInvalidValue when
// the shape is outside the kernel's requirements (the caller then uses conv_kernel's XB = 2 variant).
// LDS of a block: operand + tables (dgrad_bf16_smem) + one epilogue tile of 32 x (32 NT + 8) bf16 per wave, for the largest wave
// count the block may get at `bpc` blocks per CU.  Returns false when even one block per CU does not fit.
static bool dgrad_bf16_plan(int NT, int taps, int Kpad, int Ccat, int& bpc, size_t& smem, bool fz = false) {
    const size_t base = dgrad_bf16_smem(NT, taps, Kpad, Ccat) + (fz ? 3 * 128 * 4 : 0);
    const int maxw = dg16_max_waves(taps, NT, fz);
    for (bpc = 3; bpc >= 1; --bpc) {
        const int wmax = maxw / bpc < 4 ? 4 : maxw / bpc;
        smem = base + (size_t)wmax * 32 * (NT * 32 + 8) * 2;
        const size_t limit = bpc == 3 ? 50 * 1024 : (bpc == 2 ? 76 * 1024 : 150 * 1024);
        if (smem <= limit) return true;
    }
    return false;
}

static hipError_t launch_dgrad_bf16_impl(const ConvArgs& a, const ConvArgs* pb, int num_cus_all, hipStream_t s);
hipError_t launch_dgrad_bf16(const ConvArgs& a, int num_cus, hipStream_t s) { return launch_dgrad_bf16_impl(a, nullptr, num_cus, s); }
// a and b in one launch; hipErrorNotSupported (nothing launched) when they cannot share one
hipError_t launch_dgrad_bf16_pair(const ConvArgs& a, const ConvArgs& b, int num_cus, hipStream_t s) {
    if (!conv_pairable(a, b) || a.taps != 1) return hipErrorNotSupported;
    return launch_dgrad_bf16_impl(a, &b, num_cus, s);
}

static hipError_t launch_dgrad_bf16_impl(const ConvArgs& a, const ConvArgs* pb, int num_cus_all, hipStream_t s) {
    const int num_cus = pb ? num_cus_all / 2 : num_cus_all;      // a pair: each problem on half of the chip
    if (a.taps == 1 && a.Kpad > 128) return hipErrorInvalidValue;      // (the 1x1 variants hold a whole tile's dY rows in 8 registers per lane)
    if (a.M % 32 || a.K % 32 || a.K != a.Kpad || (a.taps != 1 && a.taps != 9) || (a.W & 3) || a.lda % 8 || a.Nout % 8 || a.ldy % 8) return hipErrorInvalidValue;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 32 || a.seg[i].ld % 8) return hipErrorInvalidValue;
    const int ntiles = a.M / 32;
    const int ncol32 = (a.Nout + 31) / 32;
    static const int ntmax = tune_int("CUNET_DG16_NT", 2);
    int NT = 0, blocks_per_cu = 1;
    size_t smem = 0;
    float best = 1e30f;
    const bool fz = a.fz_dz != nullptr;
    if (fz && (pb || a.taps != 1 || a.Kpad != 128 || a.fz_lddz % 8 || a.fz_choff % 8 || a.fz_ldx % 8)) return hipErrorNotSupported;
    for (int c = ntmax >= 2 ? 2 : 1; c >= 1; --c) {
        int bpc; size_t sm;
        if (!dgrad_bf16_plan(c, a.taps, a.Kpad, a.Ccat, bpc, sm, fz)) continue;
        const int slices = (ncol32 + c - 1) / c;
        const float cost = slices * ((float)c + 0.3f);
        if (cost < best) { best = cost; NT = c; blocks_per_cu = bpc; smem = sm; }
    }
    if (NT == 0) return hipErrorInvalidValue;
    const int gy = (ncol32 + NT - 1) / NT;
    const int max_blocks_x = (blocks_per_cu * num_cus + gy - 1) / gy;
    int waves = (ntiles + max_blocks_x - 1) / max_blocks_x;
    const int maxw = dg16_max_waves(a.taps, NT, fz);
    if (waves > maxw / blocks_per_cu) waves = maxw / blocks_per_cu;
    if (waves < 1) waves = 1;
    int gx = (ntiles + waves - 1) / waves;
    if (gx > max_blocks_x) gx = max_blocks_x;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, gy);
    const int threads = (waves < 4 ? 4 : waves) * 64;
    ConvArgs b = a;
    set_geometry_shifts(b);
    dim3 grid1 = grid;
    b.xcd_gx = b.xcd_gy = 0;
    if (gy > 1) { b.xcd_gx = gx; b.xcd_gy = gy; grid1 = dim3(8 * ((gx + 7) / 8) * gy, 1); }
    if (fz) {      // the single-consumer gather folded into the operand load (ConvArgs::fz_*)
        return NT == 2 ? launch_dg16_inst<1, 2, 4, true>(b, grid1, threads, smem, s) : launch_dg16_inst<1, 1, 4, true>(b, grid1, threads, smem, s);
    }
    if (a.taps == 1) {
        switch (a.Kpad / 32) {
#define CUNET_DG1(C) case C: return NT == 2 ? launch_dg16_inst<1, 2, C>(b, grid1, threads, smem, s, pb) : launch_dg16_inst<1, 1, C>(b, grid1, threads, smem, s, pb);
            CUNET_DG1(1) CUNET_DG1(2) CUNET_DG1(3) CUNET_DG1(4)
#undef CUNET_DG1
            default: return hipErrorInvalidValue;
        }
    }
    return NT == 2 ? launch_dg16_inst<9, 2>(b, grid1, threads, smem, s, pb) : launch_dg16_inst<9, 1>(b, grid1, threads, smem, s, pb);
}

static size_t conv_bf16_smem(int NT, int taps, int Kpad, int Ccat) {
    return (size_t)taps * (Kpad / 8) * NT * 32 * 16 + (size_t)(Ccat + 1) * 8 + (size_t)NT * 32 * 16;
}

template <int TAPS, int NT, int OUTF32>
static hipError_t launch_b16_pair_inst(const ConvArgs& a, const ConvArgs& b, dim3 grid, int threads, size_t smem, hipStream_t s) {
    if constexpr (TAPS == 1 && OUTF32 == 0) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_pair_kernel<TAPS, NT, OUTF32>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        ConvPair q;
        q.a[0] = a;
        q.a[1] = b;
        copy_launch_geometry(q.a[1], a);
        grid.z = 2;
        hipLaunchKernelGGL((conv_bf16_pair_kernel<TAPS, NT, OUTF32>), grid, dim3(threads), smem, s, q);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int TAPS, int NT, int OUTF32>
static hipError_t launch_b16_inst(const ConvArgs& a, dim3 grid, int threads, size_t smem, hipStream_t s, const ConvArgs* pb = nullptr) {
    if (pb) return launch_b16_pair_inst<TAPS, NT, OUTF32>(a, *pb, grid, threads, smem, s);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<TAPS, NT, OUTF32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_bf16_kernel<TAPS, NT, OUTF32>), grid, dim3(threads), smem, s, a);
    return hipGetLastError();
}

// a.seg[*].x, a.wB: bf16 data behind float-typed pointers; a.y: bf16 (out_f32 = 0) or fp32 (out_f32 = 1)
static hipError_t launch_conv_bf16_impl(const ConvArgs& a, const ConvArgs* pb, int out_f32, int num_cus_all, hipStream_t s);
hipError_t launch_conv_bf16(const ConvArgs& a, int out_f32, int num_cus, hipStream_t s) { return launch_conv_bf16_impl(a, nullptr, out_f32, num_cus, s); }
// a and b (1x1, bf16 out) in one launch; hipErrorNotSupported (nothing launched) when they cannot share one
hipError_t launch_conv_bf16_pair(const ConvArgs& a, const ConvArgs& b, int num_cus, hipStream_t s) {
    if (!conv_pairable(a, b) || a.taps != 1) return hipErrorNotSupported;
    return launch_conv_bf16_impl(a, &b, 0, num_cus, s);
}

static hipError_t launch_conv_bf16_impl(const ConvArgs& a, const ConvArgs* pb, int out_f32, int num_cus_all, hipStream_t s) {
    const int num_cus = pb ? num_cus_all / 2 : num_cus_all;      // a pair: each problem on half of the chip
    if (a.M % 32 || a.K % 32 || a.K != a.Kpad || (a.taps != 1 && a.taps != 9)) return hipErrorInvalidValue;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 32 || a.seg[i].ld % 8) return hipErrorInvalidValue;
    const int ntiles = a.M / 32;
    // 3x3 (128 -> 32) up to 12 tiles per CU (64 x 64 at batch 24): one tap per wave (tuning builds: CUNET_B16_TS = tiles per CU, 0 = off)
    // 3x3 (128 -> 32) at 64 x 64 / 32 x 32 with at least two image rows per CU: rows through an LDS ring (tuning builds: CUNET_B16_RING = 0 off)
    static const int use_ring = tune_int("CUNET_B16_RING", 1);
    if (use_ring && conv3x3_ring_bf16_supported(a, out_f32) && a.M / a.W >= (a.ring_min_rows > 0 ? a.ring_min_rows : 512))
        return launch_conv3x3_ring_bf16(a, num_cus, s);
    static const int use_ts = tune_int("CUNET_B16_TS", 12);
    if (use_ts && a.taps == 9 && !out_f32 && a.nseg == 1 && a.K == 128 && a.Nout == 32 && a.Npad == 32 && !a.seg[0].ups && a.ldy % 2 == 0 &&
        ntiles <= (long)use_ts * num_cus)
        return launch_conv3x3_tapsplit_bf16(a, num_cus, s);
    const int ncol32 = (a.Nout + 31) / 32;
    const long target = 2L * 4 * num_cus;
    int NT = 1;
    float best = 1e30f;
    for (int c = 4; c >= 1; c = (c == 4 ? 2 : c - 1)) {       // 4, 2, 1
        if (c == 4 && out_f32) continue;       // (heads: the fp32-output epilogue with the fused loss spilled 54 registers at four channel tiles)
        if (conv_bf16_smem(c, a.taps, a.Kpad, a.Ccat) > 150 * 1024) continue;
        const int slices = (ncol32 + c - 1) / c;
        if (c > 1 && (long)ntiles * slices < target) continue;
        const float cost = slices * ((float)c + 0.3f);
        if (cost < best) { best = cost; NT = c; }
    }
    const size_t smem = conv_bf16_smem(NT, a.taps, a.Kpad, a.Ccat);
    if (smem > 150 * 1024) return hipErrorInvalidValue;
    const int gy = (ncol32 + NT - 1) / NT;
    const int blocks_per_cu = smem > 76 * 1024 ? 1 : (smem > 50 * 1024 ? 2 : 3);
    const int max_blocks_x = (blocks_per_cu * num_cus + gy - 1) / gy;
    int waves = (ntiles + max_blocks_x - 1) / max_blocks_x;
    if (waves > b16_max_waves(NT) / blocks_per_cu) waves = b16_max_waves(NT) / blocks_per_cu;
    if (waves < 1) waves = 1;
    int gx = (ntiles + waves - 1) / waves;
    if (gx > max_blocks_x) gx = max_blocks_x;
    if (gx < 1) gx = 1;
    dim3 grid(gx, gy);
    ConvArgs b = a;
    static const int dbg16 = tune_int("CUNET_B16_DBG", 0);      // tuning builds only: work-skipping timing experiments (see the kernel)
    b.dbg = dbg16;
    static const int xcd_fwd = tune_int("CUNET_CONV_XCD_FWD", 1);
    b.xcd_gx = b.xcd_gy = 0;
    if (xcd_fwd && gy > 1) { b.xcd_gx = gx; b.xcd_gy = gy; grid = dim3(8 * ((gx + 7) / 8) * gy, 1); }
    const int threads = (waves < 4 ? 4 : waves) * 64;
#define CUNET_B16(T, N) \
    if (a.taps == T && NT == N) return out_f32 ? launch_b16_inst<T, N, 1>(b, grid, threads, smem, s, pb) : launch_b16_inst<T, N, 0>(b, grid, threads, smem, s, pb);
    CUNET_B16(1, 1) CUNET_B16(1, 2) CUNET_B16(9, 1) CUNET_B16(9, 2)
#undef CUNET_B16
    if (NT == 4 && !out_f32) return a.taps == 1 ? launch_b16_inst<1, 4, 0>(b, grid, threads, smem, s, pb) : launch_b16_inst<9, 4, 0>(b, grid, threads, smem, s, pb);
    return hipErrorInvalidValue;
}

}  // namespace cunet
