// Device helpers shared by the convolution kernels (conv_kernels.hip, dgrad3_kernels.hip): the per-4-channel group table
// of a virtual concat, the BatchNorm scale / shift (+ mean / inverse std) tables every consumer derives from the
// producers' batch statistics, and fp64 reduction primitives.
#pragma once
#include "common.h"

namespace cunet {

// host: log2 of W and H*W when both are powers of two (ConvArgs::wshift / hwshift), and whether any segment is up-sampled
inline void set_geometry_shifts(ConvArgs& a) {
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (v > 0 && (1 << l) == v) ? l : -1; };
    const int lw = lg2(a.W), lhw = lg2(a.H * a.W);
    a.wshift = (lw >= 0 && lhw >= 0) ? lw : -1;
    a.hwshift = (lw >= 0 && lhw >= 0) ? lhw : -1;
    a.any_ups = 0;
    for (int i = 0; i < a.nseg; ++i) a.any_ups |= a.seg[i].ups;
}


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
template <bool NEED_MEAN, int XB>
__device__ __forceinline__ void setup_concat(const ConvArgs& p, GrpEnt* grp, float* sc, float* sh,
                                             float* mu, float* is) {
    const int tid = threadIdx.x;
    for (int s = 0; s < p.nseg; ++s) {
        const Seg sg = p.seg[s];
        for (int lc = tid; lc < sg.C; lc += blockDim.x) {
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
        for (int g = tid; g < (sg.C >> 2); g += blockDim.x) {
            GrpEnt e;
            e.ptr = XB ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(sg.x) + 4 * g)     // bf16 storage
                       : sg.x + 4 * g;
            e.ld = sg.ld;
            e.ups = sg.ups;
            grp[(sg.choff >> 2) + g] = e;
        }
    }
}

}  // namespace cunet
