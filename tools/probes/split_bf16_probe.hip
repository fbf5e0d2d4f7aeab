// Probe (round 4): an fp32 contraction on the bf16 matrix pipe.  a = h + m + l with h, m, l bf16 (8 + 8 + 8 significand bits: an fp32
// value exactly), a * b ~ hh' + hm' + mh' + mm' + hl' + lh' (six v_mfma_f32_32x32x16_bf16 per 16 k, 6 x 32 cycles) against eight
// v_mfma_f32_32x32x2_f32 (8 x 64 cycles) for the same 16 k.  Two questions:
//   1. numerics: max / rms relative error of a 32 x K x 32 product against fp64, native fp32 MFMA vs 3 / 6 / 9 products
//   2. rate: a weight-stationary loop as the conv kernels run it (A chunk in registers with a BatchNorm + ReLU, B fragments from
//      LDS, NT channel tiles per wave, 12 or 8 waves per CU), native vs split, in fp32-equivalent TFLOP/s
// build: hipcc -O3 --offload-arch=gfx950 -o split_bf16_probe.bin split_bf16_probe.hip ; run: ./split_bf16_probe.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// eight fp32 -> three packed-bf16 operand registers (round to nearest even at every level)
__device__ __forceinline__ void split8(const float (&f)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 a = {f[2 * j], f[2 * j + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
        const f32x2 hf = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
        const f32x2 r1 = a - hf;
        const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 mf = {__uint_as_float(mu << 16), __uint_as_float(mu & 0xffff0000u)};
        const f32x2 r2 = r1 - mf;
        const unsigned lu = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
        h[j] = hu; m[j] = mu; l[j] = lu;
    }
}
__device__ __forceinline__ f32x16 mm(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- numerics: C[32][32] = A[32][K] * B[K][32], one wave, MODE 0 native fp32, 3 / 6 / 9 = products of the split
template <int MODE>
__global__ void product_kernel(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kh = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int s = 0; s < K / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + 2 * s + kh], B[(2 * s + kh) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int s = 0; s < K / 16; ++s) {
            float a[8], b[8];
            for (int j = 0; j < 8; ++j) { a[j] = A[i * K + 16 * s + 8 * kh + j]; b[j] = B[(16 * s + 8 * kh + j) * 32 + i]; }
            u32x4 ah, am, al, bh, bm, bl;
            split8(a, ah, am, al);
            split8(b, bh, bm, bl);
            // smallest terms first
            if (MODE >= 9) { acc = mm(al, bl, acc); acc = mm(am, bl, acc); acc = mm(al, bm, acc); }
            if (MODE >= 6) { acc = mm(ah, bl, acc); acc = mm(al, bh, acc); acc = mm(am, bm, acc); }
            acc = mm(ah, bm, acc);
            acc = mm(am, bh, acc);
            acc = mm(ah, bh, acc);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r >> 2) * 8 + kh * 4 + (r & 3)) * 32 + i] = acc[r];
}

// ---- rate: the conv kernels' inner loop -------------------------------------------------------------------------------------
// fp32: B fragments [chunk 0..3][8 rows][NT * 32 columns] float4 (16 KB x NT); split: [chunk][step 0..1][plane 0..2][NT][64 lanes] uint4
template <int MODE, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate_kernel(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NB = NT * 32;
    float4* Bf = reinterpret_cast<float4*>(smem);
    u32x4* Bp = reinterpret_cast<u32x4*>(smem);
    const int words = (MODE == 0 ? 4 * 8 * NB : 4 * 2 * 3 * NT * 64) * 4;
    for (int i = tid; i < words; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = MODE == 0 ? 1e-3f * (i & 15) : __uint_as_float(0x3c003c00u);
    __syncthreads();
    f32x16 acc[NT];
    for (int nt = 0; nt < NT; ++nt) for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    float4 a[4];
    for (int q = 0; q < 4; ++q) a[q] = make_float4(seed + lane, seed - lane, seed * lane, seed + q);
    const float sc = 0.999f + seed, sh = 1e-3f * seed;
    for (int it = 0; it < iters; ++it) {
        int lo = lane;
        asm volatile("" : "+v"(lo));                  // (B fragments are re-read every pass, as in the kernels: nothing hoisted out of the loop)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            // the A chunk of the iteration: BatchNorm + ReLU on the 16 values in hand (what the loaders do)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q].x = fmaxf(fmaf(a[q].x, sc, sh), 0.f); a[q].y = fmaxf(fmaf(a[q].y, sc, sh), 0.f);
                a[q].z = fmaxf(fmaf(a[q].z, sc, sh), 0.f); a[q].w = fmaxf(fmaf(a[q].w, sc, sh), 0.f);
            }
            if (MODE == 0) {
                const float4* bb = Bf + (size_t)ch * 8 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 bv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = bb[(2 * q + (lo >> 5)) * NB + nt * 32 + (lo & 31)];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bv[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bv[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bv[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bv[nt].w, acc[nt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float f[8] = {a[2 * t].x, a[2 * t].y, a[2 * t].z, a[2 * t].w, a[2 * t + 1].x, a[2 * t + 1].y, a[2 * t + 1].z, a[2 * t + 1].w};
                    u32x4 ah, am, al;
                    split8(f, ah, am, al);
                    const u32x4* bb = Bp + (size_t)((ch * 2 + t) * 3) * NT * 64 + lo;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const u32x4 bh = bb[(0 * NT + nt) * 64], bm = bb[(1 * NT + nt) * 64], bl = bb[(2 * NT + nt) * 64];
                        acc[nt] = mm(ah, bl, acc[nt]);
                        acc[nt] = mm(al, bh, acc[nt]);
                        acc[nt] = mm(am, bm, acc[nt]);
                        acc[nt] = mm(ah, bm, acc[nt]);
                        acc[nt] = mm(am, bh, acc[nt]);
                        acc[nt] = mm(ah, bh, acc[nt]);
                    }
                }
            }
        }
    }
    float s = 0.f;
    for (int nt = 0; nt < NT; ++nt) for (int r = 0; r < 16; ++r) s += acc[nt][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int NT, int WAVES>
static void rate(const char* what, float* out, int blocks, int iters) {
    const size_t smem = (MODE == 0 ? (size_t)4 * 8 * NT * 32 * 16 : (size_t)4 * 2 * 3 * NT * 64 * 16);
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<MODE, NT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((rate_kernel<MODE, NT, WAVES>), dim3(blocks), dim3(WAVES * 64), smem, 0, out, iters, 1e-6f * rep);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = (double)blocks * WAVES * iters * 4 * NT * 65536.0;
    printf("%-34s NT %d waves %2d blocks %4d: %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", what, NT, WAVES, blocks, best, flop / (best * 1e-3) * 1e-12);
}

int main() {
    // ---- numerics
    for (int K : {32, 128, 512}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> A(32 * K), B(K * 32);
            srand(7 + K + dist);
            auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
            for (auto& v : A) v = dist ? rnd() * std::exp2f((float)(rand() % 17) - 8.f) : std::fmax(rnd() + 0.3f, 0.f);      // post-ReLU-like / wide exponents
            for (auto& v : B) v = dist ? rnd() * std::exp2f((float)(rand() % 17) - 8.f) : 0.1f * rnd();
            float *dA, *dB, *dC;
            CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dC, 1024 * 4));
            CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
            CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            std::vector<double> ref(1024), mag(1024);
            for (int i = 0; i < 32; ++i)
                for (int n = 0; n < 32; ++n) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + n]; m += std::fabs((double)A[i * K + k] * B[k * 32 + n]); }
                    ref[i * 32 + n] = s; mag[i * 32 + n] = m;
                }
            for (int mode : {0, 3, 6, 9}) {
                if (mode == 0) hipLaunchKernelGGL(product_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                if (mode == 3) hipLaunchKernelGGL(product_kernel<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                if (mode == 6) hipLaunchKernelGGL(product_kernel<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                if (mode == 9) hipLaunchKernelGGL(product_kernel<9>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                std::vector<float> C(1024);
                CHK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
                double worst = 0, sq = 0;
                for (int e = 0; e < 1024; ++e) {
                    const double err = std::fabs(C[e] - ref[e]) / mag[e];       // relative to sum |a b|: the scale rounding errors live on
                    worst = std::fmax(worst, err); sq += err * err;
                }
                printf("K %3d %-14s %-22s max err / sum|ab| %.3e  rms %.3e\n", K, dist ? "wide exponents" : "relu x small w",
                       mode == 0 ? "native fp32 MFMA" : mode == 3 ? "3 products (hh hm mh)" : mode == 6 ? "6 products" : "9 products", worst, std::sqrt(sq / 1024));
            }
            CHK(hipFree(dA)); CHK(hipFree(dB)); CHK(hipFree(dC));
        }
    }
    // ---- rate
    float* out;
    CHK(hipMalloc(&out, (size_t)4096 * 1024 * 4));
    const int iters = 400;
    rate<0, 1, 12>("native fp32 32x32x2", out, 256, iters);
    rate<1, 1, 12>("split bf16, 6 products", out, 256, iters);
    rate<0, 2, 12>("native fp32 32x32x2", out, 256, iters);
    rate<1, 2, 12>("split bf16, 6 products", out, 256, iters);
    rate<0, 4, 12>("native fp32 32x32x2", out, 256, iters);
    rate<1, 4, 12>("split bf16, 6 products", out, 256, iters);
    rate<0, 4, 8>("native fp32 32x32x2", out, 256, iters);
    rate<1, 4, 8>("split bf16, 6 products", out, 256, iters);
    rate<0, 1, 16>("native fp32 32x32x2", out, 256, iters);
    rate<1, 1, 16>("split bf16, 6 products", out, 256, iters);
    printf("done\n");
    return 0;
}
