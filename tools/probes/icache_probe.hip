// Does a kernel's CODE SIZE show up in the duration of a small dependent launch?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/icp tools/probes/icache_probe.hip && /tmp/icp
// Kernels whose body is N independent straight-line VALU instructions executed ONCE by one workgroup per CU (N x 8 bytes of code).  A burst of
// launches queued behind a long kernel (host far ahead) gives GPU time per launch: the SAME kernel back to back (its code stays in the
// instruction caches) against two DIFFERENT kernels of the same size alternating (each launch finds the other one's code in the cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int N, int SALT>
__global__ void k_code(float* p, float a) {
    float v0 = a, v1 = a + 1.f, v2 = a + 2.f, v3 = a + 3.f;
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {      // 4 independent chains: issue-bound, one instruction per cycle when the code is there
        v0 = __builtin_fmaf(v0, a, 0.5f + SALT);
        v1 = __builtin_fmaf(v1, a, 0.25f);
        v2 = __builtin_fmaf(v2, a, 0.125f);
        v3 = __builtin_fmaf(v3, a, 2.0f);
    }
    if (v0 + v1 + v2 + v3 == 123.456f) p[0] = v0;
}
// the same arithmetic as a LOOP of 16 instructions (its code is fetched once)
template <int N>
__global__ void k_loop(float* p, float a) {
    float v0 = a, v1 = a + 1.f, v2 = a + 2.f, v3 = a + 3.f;
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0 = __builtin_fmaf(v0, a, 0.5f);
            v1 = __builtin_fmaf(v1, a, 0.25f);
            v2 = __builtin_fmaf(v2, a, 0.125f);
            v3 = __builtin_fmaf(v3, a, 2.0f);
        }
    }
    if (v0 + v1 + v2 + v3 == 123.456f) p[0] = v0;
}
__global__ void k_spin(long ticks) { const long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2); }
template <int N>
int run(hipStream_t s, float* buf, hipEvent_t a, hipEvent_t b) {
    float ms_same, ms_alt, ms_loop;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_loop<N>), dim3(256), dim3(256), 0, s, buf, 1.f);
    CHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000L);
    CHK(hipEventRecord(a, s));
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL((k_loop<N>), dim3(256), dim3(256), 0, s, buf, 1.f);
    CHK(hipEventRecord(b, s));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms_loop, a, b));
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL((k_code<N, 0>), dim3(256), dim3(256), 0, s, buf, 1.f); hipLaunchKernelGGL((k_code<N, 1>), dim3(256), dim3(256), 0, s, buf, 1.f); }
    CHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000L);
    CHK(hipEventRecord(a, s));
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL((k_code<N, 0>), dim3(256), dim3(256), 0, s, buf, 1.f);
    CHK(hipEventRecord(b, s));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms_same, a, b));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 300000L);
    CHK(hipEventRecord(a, s));
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL((k_code<N, 0>), dim3(256), dim3(256), 0, s, buf, 1.f); hipLaunchKernelGGL((k_code<N, 1>), dim3(256), dim3(256), 0, s, buf, 1.f); }
    CHK(hipEventRecord(b, s));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms_alt, a, b));
    printf("%6d v_fma (straight line: %4d KB of code), 256 workgroups x 256: as a 16-instruction loop %6.2f us per launch; straight line, same kernel back to back %6.2f us, two kernels alternating %6.2f us\n",
           N, N * 8 / 1024, 1e3f * ms_loop / 400, 1e3f * ms_same / 400, 1e3f * ms_alt / 400);
    return 0;
}
int main() {
    float* buf; CHK(hipMalloc(&buf, 4096));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    if (run<256>(s, buf, a, b)) return 1;
    if (run<1024>(s, buf, a, b)) return 1;
    if (run<4096>(s, buf, a, b)) return 1;
    if (run<8192>(s, buf, a, b)) return 1;
    if (run<16384>(s, buf, a, b)) return 1;
    return 0;
}
