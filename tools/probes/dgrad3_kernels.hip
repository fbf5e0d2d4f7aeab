// NEGATIVE RESULT (round 2), kept out of the library: an LDS-staged 1x1 data gradient in the style of wgrad3.  Correct
// (it passed tests/test_gpu_nodes.py when wired in) but SLOWER than conv_kernel's EP_BWD: 2.74 ms per CU-Net-2 step for
// the class alone against 1.89 ms (2756 vs 3050 img/s).  The accumulators of a 64-row x 320-column workgroup tile leave
// room for two 4-wave workgroups per CU only, and the epilogue (five dependent rounds of 16 X loads -> mask -> 16 dz
// stores per wave) is then exposed: conv_kernel's twelve independent one-tile waves per CU hide that latency better.
//
// 1x1 data gradient of the fused [concat -> BN -> ReLU] -> conv nodes, LDS-staged and software-pipelined:
//     dz[m][c] = relu'(z[m][c]) * sum_n dY[m][n] * W[n][c]        z = bn(x),   Cout = 128 -> Ccat = 128 ... 320
//     red[0][c] = sum_m dz,  red[1][c] = sum_m dz * xhat          (= dbeta, dgamma and the BN-backward coefficients)
// (the data-gradient half of autograd for models/cu_net.py:11-17,24,43; EP_BWD of conv_kernel for these shapes).
//
// Why.  conv_kernel runs these nodes with one 32-column slice per block (every slice re-reads dY: 5-10 passes), one
// accumulator per wave and an epilogue that the same wave serialises behind its MFMAs: rocprof shows the matrix pipe
// 33 % busy with 52 % of the wave cycles in issue stalls (profiles/r02_pmc_sq.txt).  Here a 256-thread workgroup owns
// 64 rows x ALL Ccat columns: the 16-channel K chunks of dY [64][16] and of the weight operand [16][Ccat] are staged in
// LDS (double buffered; the next chunk's global loads fly across the MFMAs), each wave keeps 1 row tile x <= 5 column
// tiles of accumulators fed by conflict-free ds_read_b128 one step ahead of the MFMAs, dY is read from HBM once, and
// two workgroups per CU interleave one's epilogue (X loads, mask, dz stores, reductions) with the others' MFMAs.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

namespace cunet {

constexpr int DG3_ROWS = 64;         // rows of a workgroup tile (2 row tiles of 32)
constexpr int DG3_K = 128;           // contraction length = output channels of the forward conv
constexpr int DG3_KC = 16;           // K chunk
constexpr int DG3_AP = DG3_KC + 4;   // LDS pitch of the A chunk in floats: 16-byte rows, conflict-free ds_read_b128
constexpr int DG3_THREADS = 256;

template <int CTW, int XB>           // CTW = column tiles per wave = ceil(Ccat / 64); XB: x of the concat stored as bf16
__global__ __launch_bounds__(DG3_THREADS, 2) void dgrad3_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = p.Nout;                                 // = Ccat, multiple of 32
    const int ct = N >> 5;
    GrpEnt* grp = reinterpret_cast<GrpEnt*>(smem);         // [N / 4]
    float* sc = reinterpret_cast<float*>(grp + (N >> 2));
    float* sh = sc + N;
    float* mu = sh + N;
    float* is = mu + N;
    double* redbuf = reinterpret_cast<double*>(is + N);    // [N][2]
    float* stage0 = reinterpret_cast<float*>(redbuf + 2 * N);
    const int bstage = DG3_ROWS * DG3_AP;                  // floats of the A part of a stage; the B part has 16 * N
    const int stagesz = bstage + DG3_KC * N;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int hi = lane >> 5;
    const int HW = p.H * p.W;

    setup_concat<true, XB>(p, grp, sc, sh, mu, is);
    bool any_ups = false;
    for (int i = 0; i < p.nseg; ++i) any_ups |= p.seg[i].ups != 0;
    for (int i = tid; i < 2 * N; i += DG3_THREADS) redbuf[i] = 0.0;

    // ---- staging plan: one float4 of dY (row tid/4, channels 4*(tid%4)), CTW float4 of the weight operand per chunk
    const int arow = tid >> 2, ak4 = tid & 3;
    const int nb4 = 4 * N;                                 // float4 items of a B chunk: [4 k-quads][N]
    int bidx[CTW];
#pragma unroll
    for (int j = 0; j < CTW; ++j) { const int i = tid + DG3_THREADS * j; bidx[j] = i < nb4 ? i : nb4 - 1; }

    const int rt = wave & 1;                               // row tile of this wave
    const int half = wave >> 1;
    const int cb = half * CTW;                             // first column tile
    int ctile[CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t) ctile[t] = (cb + t < ct) ? cb + t : ct - 1;

    const int ntiles = (p.M + DG3_ROWS - 1) / DG3_ROWS;
    float4 av, bv[CTW];
    bool aok = false;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * DG3_ROWS;
        auto issue = [&](int chunk) {                      // global -> registers
            const int m = m0 + arow;
            aok = m < p.M;
            av = ldg4(p.a + (size_t)(aok ? m : p.M - 1) * p.lda + chunk * DG3_KC + 4 * ak4);
#pragma unroll
            for (int j = 0; j < CTW; ++j) {
                const int k4 = bidx[j] / N, n = bidx[j] - k4 * N;
                bv[j] = ldg4(p.wB + ((size_t)(chunk * 4 + k4) * p.Npad + n) * 4);
            }
        };
        auto commit = [&](float* st) {
            *reinterpret_cast<float4*>(st + arow * DG3_AP + 4 * ak4) = aok ? av : make_float4(0.f, 0.f, 0.f, 0.f);
            float4* B = reinterpret_cast<float4*>(st + bstage);
#pragma unroll
            for (int j = 0; j < CTW; ++j) B[bidx[j]] = bv[j];
        };

        f32x16 acc[CTW];
#pragma unroll
        for (int t = 0; t < CTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        issue(0);
        __syncthreads();                                   // tables ready (first tile) / previous tile's stage reads done
        commit(stage0);
        __syncthreads();
        constexpr int NCH = DG3_K / DG3_KC;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const float* st = stage0 + (chunk & 1) * stagesz;
            if (chunk + 1 < NCH) issue(chunk + 1);
            // operands of k-quad pair q: A fragment = dY[row][8q + 4 hi ..+3], B fragment = quad (2q + hi) of the column
            const float4* A4 = reinterpret_cast<const float4*>(st + (rt * 32 + li) * DG3_AP + 4 * hi);
            const float4* B4 = reinterpret_cast<const float4*>(st + bstage) + hi * N + li;
#pragma unroll
            for (int q = 0; q < DG3_KC / 8; ++q) {
                const float4 a_cur = A4[2 * q];
                float4 b_cur[CTW];
#pragma unroll
                for (int t = 0; t < CTW; ++t) b_cur[t] = B4[(2 * q) * N + ctile[t] * 32];
#pragma unroll
                for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[t].y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur[t].z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur[t].w, acc[t], 0, 0, 0);
            }
            if (chunk + 1 < NCH) commit(stage0 + ((chunk + 1) & 1) * stagesz);
            __syncthreads();
        }

        // ---- epilogue: BatchNorm / ReLU backward, first half.  C layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        float xq[16];
        auto fetch = [&](int t, float (&xo)[16]) {         // X[row][col] of this lane's 16 rows, column tile t (always valid addresses)
            const int col = ctile[t] * 32 + li;
            const GrpEnt g = grp[col >> 2];
            const size_t coff = (size_t)(col & 3);
            if (any_ups && g.ups) {                        // nearest-upsample index map (y >> 1, x >> 1); only the first nodes of the up blocks
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int mm = m0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    mm = mm < p.M ? mm : p.M - 1;
                    const int ni = mm / HW;
                    const int rm = mm - ni * HW;
                    const int yy = rm / p.W;
                    const int xx = rm - yy * p.W;
                    xo[r] = ldx1<XB>(g.ptr, coff + (size_t)(ni * (HW >> 2) + (yy >> 1) * (p.W >> 1) + (xx >> 1)) * g.ld);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int mm = m0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    mm = mm < p.M ? mm : p.M - 1;
                    xo[r] = ldx1<XB>(g.ptr, coff + (size_t)mm * g.ld);
                }
            }
        };
#pragma unroll
        for (int t = 0; t < CTW; ++t) {                    // 16 loads in flight per tile; the other workgroups of the CU cover the latency
            if (cb + t >= ct) continue;                    // the clamped duplicate of an odd tile count
            __builtin_amdgcn_sched_barrier(0);             // keep the tiles' load bursts apart (hoisting all 80 loads spills the accumulators)
            fetch(t, xq);
            const int col = (cb + t) * 32 + li;
            const float csc = sc[col], csh = sh[col], cmu = mu[col], cis = is[col];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (mm < p.M) {
                    const float xv = xq[r];
                    const float z = fmaf(xv, csc, csh);
                    const float dz = (z > 0.f && (p.qin_bits == 0 || z < 1.f)) ? acc[t][r] : 0.f;
                    p.y[(size_t)mm * p.ldy + col] = dz;
                    s1 += dz;
                    s2 = fmaf(dz, (xv - cmu) * cis, s2);
                }
            }
            // per-column partial sums of this tile: lanes (l, l + 32), then the block's fp64 accumulators in LDS
            const double a = (double)s1 + shfl_xor_d((double)s1, 32), b = (double)s2 + shfl_xor_d((double)s2, 32);
            if (hi == 0) {
                atomicAdd(&redbuf[col * 2 + 0], a);
                atomicAdd(&redbuf[col * 2 + 1], b);
            }
        }
    }

    // ---- one fp64 atomic per column per block
    __syncthreads();
    for (int c = tid; c < N; c += DG3_THREADS) {
        atomic_add_f64(p.ystats + c, redbuf[c * 2 + 0]);
        atomic_add_f64(p.ystats + N + c, redbuf[c * 2 + 1]);
    }
}

bool dgrad3_supported(const ConvArgs& a) {
    if (a.taps != 1 || a.K != DG3_K || a.Kpad != DG3_K || a.lda != DG3_K || a.xbf16 == 2) return false;
    if (a.Nout % 32 || a.Nout < 64 || a.Nout > 320 || a.Nout != a.Ccat || a.ldy != a.Nout || a.Npad < a.Nout || a.ystats == nullptr) return false;
    for (int i = 0; i < a.nseg; ++i)
        if (a.seg[i].C % 4) return false;
    return true;
}

static size_t dgrad3_smem(int N) {
    return (size_t)(N / 4) * sizeof(GrpEnt) + (size_t)4 * N * 4 + (size_t)2 * N * 8 + (size_t)2 * (DG3_ROWS * DG3_AP + DG3_KC * N) * 4;
}

hipError_t launch_dgrad3(const ConvArgs& a, int num_cus, hipStream_t s) {
    if (!dgrad3_supported(a)) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[] = {(const void*)&dgrad3_kernel<1, 0>, (const void*)&dgrad3_kernel<2, 0>, (const void*)&dgrad3_kernel<3, 0>,
                             (const void*)&dgrad3_kernel<4, 0>, (const void*)&dgrad3_kernel<5, 0>,
                             (const void*)&dgrad3_kernel<1, 1>, (const void*)&dgrad3_kernel<2, 1>, (const void*)&dgrad3_kernel<3, 1>,
                             (const void*)&dgrad3_kernel<4, 1>, (const void*)&dgrad3_kernel<5, 1>};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    const int ct = a.Nout / 32;
    const int ctw = (ct + 1) / 2;
    const int ntiles = (a.M + DG3_ROWS - 1) / DG3_ROWS;
    const size_t smem = dgrad3_smem(a.Nout);
    int per_cu = (int)((160 * 1024) / smem);
    if (per_cu > 2) per_cu = 2;                             // <= 256 VGPRs per lane: two 4-wave workgroups per CU
    if (per_cu < 1) per_cu = 1;
    int grid = per_cu * num_cus;
    if (grid > ntiles) grid = ntiles;
#define CUNET_DG3(C) do { if (a.xbf16) hipLaunchKernelGGL((dgrad3_kernel<C, 1>), dim3(grid), dim3(DG3_THREADS), smem, s, a); \
                          else hipLaunchKernelGGL((dgrad3_kernel<C, 0>), dim3(grid), dim3(DG3_THREADS), smem, s, a); } while (0)
    switch (ctw) {
        case 1: CUNET_DG3(1); break;
        case 2: CUNET_DG3(2); break;
        case 3: CUNET_DG3(3); break;
        case 4: CUNET_DG3(4); break;
        case 5: CUNET_DG3(5); break;
        default: return hipErrorInvalidValue;
    }
#undef CUNET_DG3
    return hipGetLastError();
}

}  // namespace cunet
