// Fused [concat -> BatchNorm -> ReLU] -> convolution as an MFMA GEMM, forward and backward-data.
//
// Replaces, for the reference's hot path:
//   models/cu_net.py:11-17   cat -> norm -> relu -> conv  (1x1 bottlenecks, adapters, heads)
//   models/cu_net.py:45-48,62 norm2 -> relu2 -> conv2     (3x3, pad 1)
//   models/cu_net.py:300     conv0 7x7/2                  (stem, im2col gather)
//   and the data-gradient half of autograd for those nodes (EP_BWD).
//
// Design (gfx950, fp32): one wave owns 32 output rows (pixels) x NT*32 output channels and
// contracts with v_mfma_f32_32x32x2_f32.  The A operand (activations) is read straight from
// HBM/L2 as one 16-byte NHWC piece per lane -- lane (row = l&31, half = l>>5) takes channels
// 8q+4*half..+3 of the current 32-channel chunk -- and BatchNorm+ReLU is applied in registers,
// so a concat is never materialised and each activation is normalised exactly once per use.
// The B operand (weights, pre-packed [tap][k/4][n][4]) is staged through double-buffered LDS
// and shared by the block's 4 waves.  Because the f32 MFMA issues once per 64 cycles the
// kernel is matrix-pipe bound; loads and the BN arithmetic hide underneath.
// Per-channel batch statistics of the OUTPUT (sum, sum of squares) are produced in the
// epilogue in fp64 so every consumer BatchNorm reuses them (a BN over a concat is per channel).
#include "common.h"

namespace cunet {

struct GrpEnt {            // one 4-channel group of the concat
    const float* ptr;      // segment base + local channel
    int ld;
    int ups;
};

__device__ __forceinline__ double shfl_xor_d(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Fill sc/sh (and mu/is) for every channel of the concat, and the group table.
template <bool NEED_MEAN>
__device__ __forceinline__ void setup_concat(const ConvArgs& p, GrpEnt* grp, float* sc, float* sh,
                                             float* mu, float* is) {
    const int tid = threadIdx.x;
    for (int s = 0; s < p.nseg; ++s) {
        const Seg sg = p.seg[s];
        for (int lc = tid; lc < sg.C; lc += 256) {
            const int c = sg.choff + lc;
            double mean, istd;
            if (p.training) {
                const double sum = sg.stats[lc], sq = sg.stats[sg.C + lc];
                mean = sum / sg.count;
                double var = sq / sg.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                istd = 1.0 / sqrt(var + (double)BN_EPS);
            } else {
                mean = (double)p.rmean[c];
                istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
            }
            const double scale = (double)p.gamma[c] * istd;
            sc[c] = (float)scale;
            sh[c] = (float)((double)p.beta[c] - mean * scale);
            if (NEED_MEAN) {
                mu[c] = (float)mean;
                is[c] = (float)istd;
            }
        }
        for (int g = tid; g < (sg.C >> 2); g += 256) {
            GrpEnt e;
            e.ptr = sg.x + 4 * g;
            e.ld = sg.ld;
            e.ups = sg.ups;
            grp[(sg.choff >> 2) + g] = e;
        }
    }
}

template <int LD, int EP, int NT>
__global__ __launch_bounds__(256) void conv_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = NT * 32;                      // output channels per block
    float4* Bs = reinterpret_cast<float4*>(smem);    // [2][8][NB]
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(Bs + 2 * 8 * NB);
    float* sc = reinterpret_cast<float*>(grp + (p.Ccat >> 2));
    float* sh = sc + p.Ccat;
    float* mu = sh + p.Ccat;
    float* is = mu + p.Ccat;
    double* redbuf = reinterpret_cast<double*>(is + p.Ccat + ((p.Ccat & 1) ? 1 : 0));  // [4][NB][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int n0 = blockIdx.y * NB;

    constexpr bool HAS_CONCAT = (LD == LD_SEG || LD == LD_3X3 || EP == EP_BWD);
    if (HAS_CONCAT) {
        setup_concat<EP == EP_BWD>(p, grp, sc, sh, mu, is);
        __syncthreads();
    }

    const int HW = p.H * p.W;
    const int nck = p.Kpad >> 5;                     // 32-channel chunks per tap
    const int nchunks = p.taps * nck;
    const int ntiles = (p.M + 127) >> 7;
    const int kq4 = p.Kpad >> 2;

    double dsum[NT], dsq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dsum[nt] = 0.0; dsq[nt] = 0.0; }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wave * 32 + li;   // this lane's A row
        const int mc = m < p.M ? m : p.M - 1;
        const int nimg = mc / HW;
        const int rem = mc - nimg * HW;
        const int py = rem / p.W;
        const int px = rem - py * p.W;
        const int rowU = nimg * (HW >> 2) + (py >> 1) * (p.W >> 1) + (px >> 1);

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        // ---- loaders -------------------------------------------------------------------
        auto tap_row = [&](int t, bool& valid) -> int {
            const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
            const int yy = py + dy, xx = px + dx;
            valid = (yy >= 0) && (yy < p.H) && (xx >= 0) && (xx < p.W);
            return valid ? mc + dy * p.W + dx : mc;
        };
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
                        const float* src = g.ptr + (size_t)(g.ups ? rowU : mc) * g.ld;
                        v = *reinterpret_cast<const float4*>(src);
                    } else if (LD == LD_3X3) {
                        bool valid;
                        const int row = tap_row(t, valid);
                        v = *reinterpret_cast<const float4*>(p.seg[0].x + (size_t)row * p.seg[0].ld + kk);
                    } else if (LD == LD_PLAIN) {
                        v = *reinterpret_cast<const float4*>(p.a + (size_t)mc * p.lda + kk);
                    } else if (LD == LD_PLAIN3) {
                        bool valid;
                        const int row = tap_row(t, valid);
                        v = *reinterpret_cast<const float4*>(p.a + (size_t)row * p.lda + kk);
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
                            e4[e] = ok ? p.img[((size_t)(nimg * 3 + ci) * p.IH + iy) * p.IW + ix] : 0.f;
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
                    } else {
                        a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        };
        auto load_b = [&](int ch, float4 (&b)[NT]) {
            const int t = ch / nck;
            const int c = ch - t * nck;
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const int j = tid + 256 * it;
                const int kq = j / NB;
                const int n = j - kq * NB;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n0 + n < p.Npad)
                    v = *reinterpret_cast<const float4*>(
                        p.wB + ((size_t)(t * kq4 + c * 8 + kq) * p.Npad + n0 + n) * 4);
                b[it] = v;
            }
        };
        auto store_b = [&](int buf, const float4 (&b)[NT]) {
#pragma unroll
            for (int it = 0; it < NT; ++it) Bs[buf * 8 * NB + tid + 256 * it] = b[it];
        };

        // ---- pipeline: A one chunk ahead in registers, B double-buffered in LDS ---------
        float4 anext[4], bnext[NT];
        load_a(0, anext);
        load_b(0, bnext);
        __syncthreads();            // previous tile's readers of Bs are done
        store_b(0, bnext);
        __syncthreads();

        for (int ch = 0; ch < nchunks; ++ch) {
            const int buf = ch & 1;
            float4 acur[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acur[q] = anext[q];
            activate(ch, acur);
            if (ch + 1 < nchunks) {
                load_a(ch + 1, anext);
                load_b(ch + 1, bnext);
            }
            const float4* bb = Bs + buf * 8 * NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bb[(2 * q + hi) * NB + nt * 32 + li];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].x, bv[nt].x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].y, bv[nt].y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].z, bv[nt].z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[q].w, bv[nt].w, acc[nt], 0, 0, 0);
                }
            }
            if (ch + 1 < nchunks) store_b(buf ^ 1, bnext);
            __syncthreads();
        }

        // ---- epilogue ---------------------------------------------------------------------
        // C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int mrow0 = tile * 128 + wave * 32;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = n0 + nt * 32 + li;
            const bool colok = col < p.Nout;
            float s1 = 0.f, s2 = 0.f;
            if (EP == EP_FWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (mm < p.M && colok) {
                        const float v = acc[nt][r];
                        p.y[(size_t)mm * p.ldy + col] = v;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
            } else {
                // BatchNorm/ReLU backward, first half: dz = relu'(z) * dA, reductions sum(dz), sum(dz*xhat)
                GrpEnt g;
                g.ptr = nullptr; g.ld = 0; g.ups = 0;
                float csc = 0.f, csh = 0.f, cmu = 0.f, cis = 0.f;
                if (colok) {
                    g = grp[col >> 2];
                    csc = sc[col]; csh = sh[col]; cmu = mu[col]; cis = is[col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (mm < p.M && colok) {
                        int row = mm;
                        if (g.ups) {
                            const int ni = mm / HW;
                            const int rm = mm - ni * HW;
                            const int yy = rm / p.W;
                            const int xx = rm - yy * p.W;
                            row = ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1);
                        }
                        const float xv = g.ptr[(size_t)row * g.ld + (col & 3)];
                        const float z = fmaf(xv, csc, csh);
                        const float dz = z > 0.f ? acc[nt][r] : 0.f;
                        p.y[(size_t)mm * p.ldy + col] = dz;
                        s1 += dz;
                        s2 = fmaf(dz, (xv - cmu) * cis, s2);
                    }
                }
            }
            dsum[nt] += (double)s1;
            dsq[nt] += (double)s2;
        }
    }

    // ---- per-channel reductions: lanes (l, l+32) -> 4 waves via LDS -> one fp64 atomic per channel
    if (p.ystats != nullptr) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double a = dsum[nt] + shfl_xor_d(dsum[nt], 32);
            const double b = dsq[nt] + shfl_xor_d(dsq[nt], 32);
            if (hi == 0) {
                redbuf[(wave * NB + nt * 32 + li) * 2 + 0] = a;
                redbuf[(wave * NB + nt * 32 + li) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < NB) {
            const int col = n0 + tid;
            if (col < p.Nout) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a += redbuf[(w * NB + tid) * 2 + 0];
                    b += redbuf[(w * NB + tid) * 2 + 1];
                }
                atomic_add_f64(p.ystats + col, a);
                atomic_add_f64(p.ystats + p.Nout + col, b);
            }
        }
    }
}

size_t conv_smem_bytes(int NT, int Ccat) {
    size_t b = (size_t)2 * 8 * NT * 32 * 16;           // Bs
    b += (size_t)(Ccat / 4) * sizeof(GrpEnt);          // group table
    b += (size_t)(Ccat + (Ccat & 1)) * 4 * 4;          // sc, sh, mu, is (padded to keep 8-B alignment)
    b += 16;
    b += (size_t)4 * NT * 32 * 2 * 8;                  // reduction scratch
    return b;
}

template <int LD, int EP>
static hipError_t launch_nt(const ConvArgs& a, int NT, dim3 grid, size_t smem, hipStream_t s) {
    switch (NT) {
        case 1: hipLaunchKernelGGL((conv_kernel<LD, EP, 1>), grid, dim3(256), smem, s, a); break;
        case 2: hipLaunchKernelGGL((conv_kernel<LD, EP, 2>), grid, dim3(256), smem, s, a); break;
        default: hipLaunchKernelGGL((conv_kernel<LD, EP, 4>), grid, dim3(256), smem, s, a); break;
    }
    return hipGetLastError();
}

// Host launcher. Chooses the channel tile (NT) so that the grid fills the chip: big-M nodes
// take all output channels per block, small-M nodes split channels over blockIdx.y.
hipError_t launch_conv(const ConvArgs& a, int load, int epi, int num_cus, hipStream_t s) {
    const int ntiles = (a.M + 127) / 128;
    const int ncol32 = (a.Nout + 31) / 32;
    int NT = ncol32 >= 4 ? 4 : (ncol32 >= 2 ? 2 : 1);
    while (NT > 1 && (long)ntiles * ((ncol32 + NT - 1) / NT) < 2L * num_cus) NT >>= 1;
    const int gy = (ncol32 + NT - 1) / NT;
    int gx = ntiles;
    const int cap = 4 * num_cus;                       // persistent: bounds the fp64 atomics per node
    if (gx * gy > cap) gx = (cap + gy - 1) / gy;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, gy);
    const size_t smem = conv_smem_bytes(NT, a.Ccat);
#define CUNET_CASE(L, E) \
    if (load == L && epi == E) return launch_nt<L, E>(a, NT, grid, smem, s);
    CUNET_CASE(LD_SEG, EP_FWD)
    CUNET_CASE(LD_3X3, EP_FWD)
    CUNET_CASE(LD_STEM, EP_FWD)
    CUNET_CASE(LD_PLAIN, EP_BWD)
    CUNET_CASE(LD_PLAIN3, EP_BWD)
#undef CUNET_CASE
    return hipErrorInvalidValue;
}

}  // namespace cunet
