// Stand-alone probe (not part of libcunet_hip.so): how fast is a 1x1 weight gradient on bf16 MFMA when both operands are
// stored pixel-OCT-major  T[m/8][channel][8]  (8 consecutive pixels of one channel = one 16-byte lane load)?
//   dW[n][c] = sum_m dY[m][n] * relu(sc[c] * X[m][c] + sh[c])        M pixels, N = 128 output channels, C input channels
// v_mfma_f32_32x32x16_bf16: A lane (i = n, k-group g) holds dY of pixels 8g..8g+7 of a 16-pixel step, B lane (g, j = c)
// holds the activated X of the same pixels; both are single coalesced 16-byte loads in the oct layout.
// One wave: 4 n-tiles x CT c-tiles accumulators over its share of the pixels; 4 waves reduced through LDS; fp32 atomics.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/probes/wgrad_oct_probe.hip -o /tmp/wgrad_oct_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo16(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi16(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

template <int CT>
__global__ __launch_bounds__(256, 2) void wgrad_oct_kernel(const u16* __restrict__ dYo, const u16* __restrict__ Xo,
                                                           const float* __restrict__ sc, const float* __restrict__ sh,
                                                           float* __restrict__ dW, int M, int N, int C, int rows_per_block) {
    extern __shared__ float lds[];                 // [4 waves][1024] + [4*CT][1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
    const int c0 = blockIdx.y * 32 * CT;
    const int row_begin = blockIdx.x * rows_per_block;
    int row_end = row_begin + rows_per_block;
    if (row_end > M) row_end = M;
    float s_[CT], h_[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) { s_[t] = sc[c0 + 32 * t + li]; h_[t] = sh[c0 + 32 * t + li]; }
    f32x16 acc[4][CT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // wave w takes 16-pixel steps  row_begin + 16*(w + 4*k)
    for (int m0 = row_begin + 16 * wave; m0 < row_end; m0 += 64) {
        const size_t oct = (size_t)(m0 >> 3) + hi;
        uint4 av[4], xv[CT];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = *reinterpret_cast<const uint4*>(dYo + (oct * N + 32 * a + li) * 8);
#pragma unroll
        for (int t = 0; t < CT; ++t) xv[t] = *reinterpret_cast<const uint4*>(Xo + (oct * C + c0 + 32 * t + li) * 8);
#pragma unroll
        for (int t = 0; t < CT; ++t) {            // BN + ReLU with lane-constant scale / shift, back to bf16
            const unsigned q[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w};
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack2(fmaxf(fmaf(lo16(q[e]), s_[t], h_[t]), 0.f), fmaxf(fmaf(hi16(q[e]), s_[t], h_[t]), 0.f));
            xv[t] = make_uint4(o[0], o[1], o[2], o[3]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int t = 0; t < CT; ++t)
                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[a]), __builtin_bit_cast(bf16x8, xv[t]), acc[a][t], 0, 0, 0);
    }
    float* red = lds;
    float* sum = lds + 4 * 1024;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < CT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[a][t][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) sum[(a * CT + t) * 1024 + e] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            __syncthreads();
        }
    // C element (row n_local, col c_local) of tile (a, t): r*64 + l with row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l & 31
    const int width = 32 * CT;
    for (int idx = tid; idx < 128 * width; idx += 256) {
        const int n = idx / width, cc = idx - n * width;
        const int a = n >> 5, i = n & 31, t = cc >> 5, j = cc & 31;
        const int r = (i & 3) | ((i >> 3) << 2);
        const int e = r * 64 + ((i >> 2) & 1) * 32 + j;
        atomicAdd(dW + (size_t)n * C + c0 + cc, sum[(a * CT + t) * 1024 + e]);
    }
}

static u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(u16 v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 98304, N = 128, C = argc > 2 ? atoi(argv[2]) : 320;
    const int CT = 2;
    std::vector<u16> dY((size_t)M * N), X((size_t)M * C);
    std::vector<float> sc(C), sh(C);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.f - 0.5f; };
    // logical tensors dYl[m][n], Xl[m][c] are stored oct-major
    for (size_t m = 0; m < (size_t)M; ++m) for (int n = 0; n < N; ++n) dY[((m >> 3) * N + n) * 8 + (m & 7)] = f2bf(rnd());
    for (size_t m = 0; m < (size_t)M; ++m) for (int c = 0; c < C; ++c) X[((m >> 3) * C + c) * 8 + (m & 7)] = f2bf(rnd() * 2.f);
    for (int c = 0; c < C; ++c) { sc[c] = 0.5f + 0.01f * (c % 7); sh[c] = 0.1f * ((c % 5) - 2); }
    u16 *ddY, *dX; float *dsc, *dsh, *ddW;
    CK(hipMalloc(&ddY, dY.size() * 2)); CK(hipMalloc(&dX, X.size() * 2));
    CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4)); CK(hipMalloc(&ddW, (size_t)N * C * 4));
    CK(hipMemcpy(ddY, dY.data(), dY.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, sc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), C * 4, hipMemcpyHostToDevice));
    const int groups = C / (32 * CT);
    int chunks = (512 + groups - 1) / groups;                  // ~2 blocks per CU
    int rpb = (M + chunks - 1) / chunks; rpb = (rpb + 63) / 64 * 64; chunks = (M + rpb - 1) / rpb;
    const size_t smem = (size_t)(4 + 4 * CT) * 1024 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_oct_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    auto run = [&]() { hipLaunchKernelGGL((wgrad_oct_kernel<CT>), dim3(chunks, groups), dim3(256), smem, 0, ddY, dX, dsc, dsh, ddW, M, N, C, rpb); };
    CK(hipMemset(ddW, 0, (size_t)N * C * 4));
    run(); CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)N * C);
    CK(hipMemcpy(got.data(), ddW, got.size() * 4, hipMemcpyDeviceToHost));
    // reference on a few output elements (full M)
    double worst = 0;
    for (int k = 0; k < 24; ++k) {
        const int n = (k * 37) % N, c = (k * 53 + 3) % C;
        double ref = 0;
        for (size_t m = 0; m < (size_t)M; ++m) {
            const float x = bf2f(X[((m >> 3) * C + c) * 8 + (m & 7)]);
            const float a = bf2f(f2bf(fmaxf(fmaf(x, sc[c], sh[c]), 0.f)));
            ref += (double)bf2f(dY[((m >> 3) * N + n) * 8 + (m & 7)]) * a;
        }
        const double err = fabs(got[(size_t)n * C + c] - ref) / (fabs(ref) + 1e-3 * sqrt((double)M));
        if (err > worst) worst = err;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) run();
    CK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 50, flops = 2.0 * M * N * C;
    printf("M %d N %d C %d: grid %d x %d, %.1f us per launch, %.1f TFLOP/s, operand bytes %.0f MB -> %.2f TB/s, worst rel err %.2e\n",
           M, N, C, chunks, groups, us, flops / us / 1e6, ((double)M * (N * groups + C) * 2) / 1e6, ((double)M * (N * groups + C) * 2) / us / 1e6, worst);
    return 0;
}
