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

// ---- an fp32 contraction on the bf16 matrix pipe (planner option f32_split) --------------------------------------------------------
// Eight fp32 values -> three operand registers of v_mfma_f32_32x32x16_bf16: x = h + m + l, each piece rounded to nearest even from
// what the pieces before it left (8 + 8 + 8 significand bits: h + m + l == x exactly unless l underflows).  Elements 2 j, 2 j + 1 of
// the input are the low / high half of dword j, the operand order of the instruction.
// Non-finite inputs: x = +-Inf (or finite above the bf16 maximum, 3.39e38) gives h = +-Inf and x - h = NaN, so such an operand contributes
// NaN where the fp32 MFMA would contribute +-Inf (NaN, not 0, against a zero-padded column too); a diverged run shows as a NaN loss
// instead of an Inf one.  A select per element would restore Inf at one more VALU operation in loops that are VALU-bound already (DESIGN
// section 4a); include/cunet.h documents the limitation next to the option, f32_split = 0 keeps the fp32 instruction.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_op __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_op __attribute__((ext_vector_type(2)));
typedef float f32x2_op __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_bf16x3_pair(const f32x2_op a, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_op));                          // v_cvt_pk_bf16_f32
    const f32x2_op hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const f32x2_op r1 = a - hf;                                                                      // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_op));
    const f32x2_op mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const f32x2_op r2 = r1 - mf;                                                                     // exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_op));
}
__device__ __forceinline__ void split_bf16x3(const float (&f)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hu, mu, lu;
        split_bf16x3_pair(f32x2_op{f[2 * j], f[2 * j + 1]}, hu, mu, lu);
        h[j] = hu; m[j] = mu; l[j] = lu;
    }
}
__device__ __forceinline__ f32x16 mfma_bf16x8(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_op, a), __builtin_bit_cast(bf16x8_op, b), c, 0, 0, 0);
}
// the six products of two split operands, the small terms first
__device__ __forceinline__ f32x16 mfma_split6(const u32x4& ah, const u32x4& am, const u32x4& al, const u32x4& bh, const u32x4& bm,
                                              const u32x4& bl, f32x16 c) {
    c = mfma_bf16x8(ah, bl, c);
    c = mfma_bf16x8(al, bh, c);
    c = mfma_bf16x8(am, bm, c);
    c = mfma_bf16x8(ah, bm, c);
    c = mfma_bf16x8(am, bh, c);
    return mfma_bf16x8(ah, bh, c);
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
            if (CUNET_DBG(p, 65536)) {      // tuning builds (timing only, wrong results): tables without the statistics loads and the fp64 arithmetic
                sc[c] = 1.f; sh[c] = 0.f;
                if (NEED_MEAN) { mu[c] = 0.f; is[c] = 1.f; }
                continue;
            }
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
