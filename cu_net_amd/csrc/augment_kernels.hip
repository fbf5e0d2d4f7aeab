// Training-sample preparation on the device, the reference's way: data/mpii_for_mpii_22.py:127-141 (flip, colour gain, clamp) and
// pylib/HumanAug.py:115-172 `crop`, whose two resamplers -- scipy.misc.imresize / imrotate -- were 8-bit PIL operations behind a
// data-dependent byte-scale.  The stages of crop() are kept as stages (the 8-bit intermediates ARE the semantics):
//   [pre-shrink, scale * 200 / 256 >= 2]  byte-scale the whole image (min / max over it) -> PIL resize to int(W / sf) x int(H / sf)
//   window of the (shrunk) image on a zero canvas -> byte-scale (min / max over the canvas) -> 8-bit canvas
//   [rot != 0]  PIL rotate(BILINEAR) about the canvas centre, padding removed
//   PIL resize(BILINEAR) to res x res (horizontal pass, uint8 intermediate, vertical pass) -> uint8 -> / 255 (utils/imutils.py:31-36)
// PIL's arithmetic is restated exactly (Pillow Resample.c / Geometry.c; oracle/augment_ref.py holds the same restatement in
// numpy and tools/gen_golden.py pins that to the executed reference crop() byte for byte, G16): triangle filter of support
// max(1, in / out) with coefficients normalised in double and rounded to 22-bit fixed point, clip8((2^21 + sum) >> 22) per pass;
// rotation = inverse affine map in double, 2 x 2 bilinear with edge clamping, truncation to uint8.  Every floating-point
// expression below keeps the reference's operation order and is compiled without fma contraction.
// One thread per output element of a stage, blockIdx.y = sample; nothing here is on the train step's critical path.
#include "common.h"
#include "kernels.h"

namespace cunet {

typedef unsigned char u8;

__device__ __forceinline__ float aug_pixel(const AugSample& a, int c, int y, int x) {      // flip -> gain -> clamp, float32 as torch does
#pragma clang fp contract(off)
    const int xs = a.flip ? a.sw - 1 - x : x;
    const float v = a.src[((size_t)c * a.sh + y) * a.sw + xs] * a.gain[c];
    return fminf(fmaxf(v, 0.f), 1.f);
}

// min / max slots: values are >= 0, so their float bit patterns order like unsigned integers
__global__ void aug_init_kernel(const AugSample* tab, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned* mm = tab[i].mm;
    mm[0] = 0x7f800000u; mm[1] = 0u; mm[2] = 0x7f800000u; mm[3] = 0u; mm[4] = 0u;
}

__device__ __forceinline__ void block_minmax_commit(float lo, float hi, unsigned* slot) {
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(slot, __float_as_uint(lo));
        atomicMax(slot + 1, __float_as_uint(hi));
    }
}

// (pre-shrink samples) min / max of the gain-clamped image
__global__ __launch_bounds__(256) void aug_image_minmax_kernel(const AugSample* tab) {
    const AugSample a = tab[blockIdx.y];
    if (!a.pre) return;
    const long total = (long)3 * a.sh * a.sw;
    float lo = __uint_as_float(0x7f800000u), hi = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / ((long)a.sh * a.sw));
        const long r = i - (long)c * a.sh * a.sw;
        const float v = aug_pixel(a, c, (int)(r / a.sw), (int)(r % a.sw));
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    block_minmax_commit(lo, hi, a.mm);
}

// (pre-shrink samples) scipy.misc.bytescale of the float32 H x W x 3 image: float32 arithmetic, as numpy evaluates it
__global__ __launch_bounds__(256) void aug_image_bytescale_kernel(const AugSample* tab) {
#pragma clang fp contract(off)
    const AugSample a = tab[blockIdx.y];
    if (!a.pre) return;
    const float cmin = __uint_as_float(a.mm[0]), cmax = __uint_as_float(a.mm[1]);
    float cscale = cmax - cmin;
    if (cscale == 0.f) cscale = 1.f;
    const float scale = 255.f / cscale;
    const long total = (long)a.sh * a.sw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int y = (int)(i / a.sw), x = (int)(i % a.sw);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float b = (aug_pixel(a, c, y, x) - cmin) * scale;
            b = fminf(fmaxf(b, 0.f), 255.f) + 0.5f;
            a.i8[i * 3 + c] = (u8)b;
        }
    }
}

// One pass of Image.resize(BILINEAR) along x over an H x W x 3 uint8 image -> H x Wout x 3 (Pillow precompute_coeffs,
// normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc); `transpose` walks columns instead (the vertical pass).
__device__ __forceinline__ void pil_resample_pixel(const u8* __restrict__ src, long src_pitch /*bytes between taps*/, int in_size, int out_size,
                                                    int xx, u8* __restrict__ dst) {
#pragma clang fp contract(off)
    if (in_size == out_size) {      // Image.resize to the same size along this axis: a copy
        const u8* p = src + (long)xx * src_pitch;
        dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
        return;
    }
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        double w = t < 1.0 ? 1.0 - t : 0.0;
        if (ww != 0.0) w = w / ww;
        const int k = (int)(0.5 + w * 4194304.0);
        const u8* p = src + (long)(x + xmin) * src_pitch;
        s0 += p[0] * k; s1 += p[1] * k; s2 += p[2] * k;
    }
    s0 >>= 22; s1 >>= 22; s2 >>= 22;
    dst[0] = (u8)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
    dst[1] = (u8)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
    dst[2] = (u8)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
}

// stage: 0 = pre-shrink horizontal (i8 -> t1), 1 = pre-shrink vertical (t1 -> i1), 2 = final horizontal (c8 / r8 -> t2), 3 = final vertical (t2 -> o8)
__global__ __launch_bounds__(256) void aug_resize_kernel(const AugSample* tab, int stage, int res) {
    const AugSample a = tab[blockIdx.y];
    if (stage < 2 && !a.pre) return;
    const u8* src; u8* dst; int H, W, Hout, Wout;
    if (stage == 0) { src = a.i8; dst = a.t1; H = a.sh; W = a.sw; Hout = a.sh; Wout = a.sw1; }
    else if (stage == 1) { src = a.t1; dst = a.i1; H = a.sh; W = a.sw1; Hout = a.sh1; Wout = a.sw1; }
    else if (stage == 2) { src = a.rotated ? a.r8 : a.c8; dst = a.t2; H = a.win_h; W = a.win_w; Hout = a.win_h; Wout = res; }
    else { src = a.t2; dst = a.o8; H = a.win_h; W = res; Hout = res; Wout = res; }
    const long total = (long)Hout * Wout;
    unsigned omax = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int y = (int)(i / Wout), x = (int)(i % Wout);
        u8* d = dst + i * 3;
        if ((stage & 1) == 0) pil_resample_pixel(src + (long)y * W * 3, 3, W, Wout, x, d);             // along x
        else pil_resample_pixel(src + (long)x * 3, (long)W * 3, H, Hout, y, d);                         // along y
        if (stage == 3) omax = max(omax, max((unsigned)d[0], max((unsigned)d[1], (unsigned)d[2])));
    }
    if (stage == 3) {
        for (int o = 32; o > 0; o >>= 1) omax = max(omax, (unsigned)__shfl_xor((int)omax, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(a.mm + 4, omax);
    }
}

// value of canvas cell (y, x): the window of the (shrunk) image, zero outside it (pylib/HumanAug.py:144-159)
__device__ __forceinline__ bool canvas_src(const AugSample& a, int y, int x, int& iy, int& ix) {
    iy = y + a.uly; ix = x + a.ulx;
    const int ih = a.pre ? a.sh1 : a.sh, iw = a.pre ? a.sw1 : a.sw;
    return iy >= 0 && iy < ih && ix >= 0 && ix < iw;
}

__global__ __launch_bounds__(256) void aug_canvas_minmax_kernel(const AugSample* tab) {
    const AugSample a = tab[blockIdx.y];
    const long total = (long)a.ch * a.cw;
    float lo = __uint_as_float(0x7f800000u), hi = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int iy, ix;
        if (canvas_src(a, (int)(i / a.cw), (int)(i % a.cw), iy, ix)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = a.pre ? (float)a.i1[((long)iy * a.sw1 + ix) * 3 + c] : aug_pixel(a, c, iy, ix);
                lo = fminf(lo, v); hi = fmaxf(hi, v);
            }
        } else {
            lo = fminf(lo, 0.f);
        }
    }
    block_minmax_commit(lo, hi, a.mm + 2);
}

// scipy.misc.bytescale of the float64 canvas
__global__ __launch_bounds__(256) void aug_canvas_bytescale_kernel(const AugSample* tab) {
#pragma clang fp contract(off)
    const AugSample a = tab[blockIdx.y];
    const double cmin = (double)__uint_as_float(a.mm[2]), cmax = (double)__uint_as_float(a.mm[3]);
    double cscale = cmax - cmin;
    if (cscale == 0.0) cscale = 1.0;
    const double scale = 255.0 / cscale;
    const long total = (long)a.ch * a.cw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int iy, ix;
        const bool in = canvas_src(a, (int)(i / a.cw), (int)(i % a.cw), iy, ix);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = !in ? 0.0 : (a.pre ? (double)a.i1[((long)iy * a.sw1 + ix) * 3 + c] : (double)aug_pixel(a, c, iy, ix));
            double b = (v - cmin) * scale;
            b = fmin(fmax(b, 0.0), 255.0) + 0.5;
            a.c8[i * 3 + c] = (u8)b;
        }
    }
}

// Image.rotate(rot, BILINEAR) of the canvas, inner [pad, -pad) region only (Pillow affine_transform + bilinear_filter32RGB)
__global__ __launch_bounds__(256) void aug_rotate_kernel(const AugSample* tab) {
#pragma clang fp contract(off)
    const AugSample a = tab[blockIdx.y];
    if (!a.rotated) return;
    const long total = (long)a.win_h * a.win_w;
    const int W = a.cw, H = a.ch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / a.win_w) + a.pad, ox = (int)(i % a.win_w) + a.pad;
        const double xo = ox + 0.5, yo = oy + 0.5;
        double xin = a.rm[0] * xo + a.rm[1] * yo + a.rm[2];
        double yin = a.rm[3] * xo + a.rm[4] * yo + a.rm[5];
        u8* d = a.r8 + i * 3;
        if (xin < 0.0 || xin >= W || yin < 0.0 || yin >= H) { d[0] = d[1] = d[2] = 0; continue; }
        xin -= 0.5; yin -= 0.5;
        const double fx = floor(xin), fy = floor(yin);
        const int x = (int)fx, y = (int)fy;
        const double dx = xin - fx, dy = yin - fy;
        const int x0 = min(max(x, 0), W - 1), x1 = min(max(x + 1, 0), W - 1);
        const int r0 = min(max(y, 0), H - 1);
        const bool has2 = (y + 1 >= 0) && (y + 1 < H);
        const u8* p0 = a.c8 + (long)r0 * W * 3;
        const u8* p1 = a.c8 + (long)(has2 ? y + 1 : r0) * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a0 = p0[x0 * 3 + c], b0 = p0[x1 * 3 + c];
            const double v1 = a0 + (b0 - a0) * dx;
            double v2 = v1;
            if (has2) {
                const double a1 = p1[x0 * 3 + c], b1 = p1[x1 * 3 + c];
                v2 = a1 + (b1 - a1) * dx;
            }
            const double v = v1 + (v2 - v1) * dy;
            d[c] = (u8)v;
        }
    }
}

// uint8 res x res x 3 -> float32 3 x res x res: utils/imutils.py:31-36 im_to_torch (`if img.max() > 1: img /= 255`)
__global__ __launch_bounds__(256) void aug_finish_kernel(const AugSample* tab, float* __restrict__ out, int res) {
    const AugSample a = tab[blockIdx.y];
    const bool divide = a.mm[4] > 1u;
    const long total = (long)res * res;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)a.o8[i * 3 + c];
            out[((size_t)blockIdx.y * 3 + c) * total + i] = divide ? v / 255.f : v;
        }
    }
}

hipError_t launch_augment(const AugSample* tab_dev, const AugSample* tab_host, int n, float* out, int res, hipStream_t s) {
    long max_img = 1, max_t1 = 1, max_i1 = 1, max_canvas = 1, max_win = 1, max_t2 = 1;
    bool any_pre = false, any_rot = false;
    for (int i = 0; i < n; ++i) {
        const AugSample& a = tab_host[i];
        if (a.sh < 1 || a.sw < 1 || a.cw < 1 || a.ch < 1 || a.win_w < 1 || a.win_h < 1 || !a.src || !a.mm || !a.c8 || !a.t2 || !a.o8) return hipErrorInvalidValue;
        if (a.pre && (a.sh1 < 1 || a.sw1 < 1 || !a.i8 || !a.t1 || !a.i1)) return hipErrorInvalidValue;
        if (a.rotated && (!a.r8 || a.win_w != a.cw - 2 * a.pad || a.win_h != a.ch - 2 * a.pad)) return hipErrorInvalidValue;
        if (!a.rotated && (a.win_w != a.cw || a.win_h != a.ch)) return hipErrorInvalidValue;
        if (a.pre) {
            any_pre = true;
            max_img = std::max(max_img, (long)a.sh * a.sw);
            max_t1 = std::max(max_t1, (long)a.sh * a.sw1);
            max_i1 = std::max(max_i1, (long)a.sh1 * a.sw1);
        }
        any_rot |= a.rotated != 0;
        max_canvas = std::max(max_canvas, (long)a.ch * a.cw);
        max_win = std::max(max_win, (long)a.win_h * a.win_w);
        max_t2 = std::max(max_t2, (long)a.win_h * res);
    }
    auto gx = [](long elems) { long g = (elems + 255) / 256; return (unsigned)std::min<long>(std::max<long>(g, 1), 4096); };
    hipLaunchKernelGGL(aug_init_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tab_dev, n);
    if (any_pre) {
        hipLaunchKernelGGL(aug_image_minmax_kernel, dim3(gx(3 * max_img), n), dim3(256), 0, s, tab_dev);
        hipLaunchKernelGGL(aug_image_bytescale_kernel, dim3(gx(max_img), n), dim3(256), 0, s, tab_dev);
        hipLaunchKernelGGL(aug_resize_kernel, dim3(gx(max_t1), n), dim3(256), 0, s, tab_dev, 0, res);
        hipLaunchKernelGGL(aug_resize_kernel, dim3(gx(max_i1), n), dim3(256), 0, s, tab_dev, 1, res);
    }
    hipLaunchKernelGGL(aug_canvas_minmax_kernel, dim3(gx(max_canvas), n), dim3(256), 0, s, tab_dev);
    hipLaunchKernelGGL(aug_canvas_bytescale_kernel, dim3(gx(max_canvas), n), dim3(256), 0, s, tab_dev);
    if (any_rot) hipLaunchKernelGGL(aug_rotate_kernel, dim3(gx(max_win), n), dim3(256), 0, s, tab_dev);
    hipLaunchKernelGGL(aug_resize_kernel, dim3(gx(max_t2), n), dim3(256), 0, s, tab_dev, 2, res);
    hipLaunchKernelGGL(aug_resize_kernel, dim3(gx((long)res * res), n), dim3(256), 0, s, tab_dev, 3, res);
    hipLaunchKernelGGL(aug_finish_kernel, dim3(gx((long)res * res), n), dim3(256), 0, s, tab_dev, out, res);
    return hipGetLastError();
}

}  // namespace cunet
