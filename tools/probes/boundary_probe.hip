// What does a dependent kernel boundary cost on this stack, by kernel shape?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/boundary_probe tools/probes/boundary_probe.hip && /tmp/boundary_probe
// Chains of N dependent launches on one stream, wall time per launch (hipEvent pair around the chain, host far ahead):
// an empty kernel by grid size, by kernel-argument size (ConvArgs is ~700 bytes by value), with dynamic LDS, with a tail of
// same-address fp64 atomics (the BatchNorm statistics every producer ends with), and a small streaming body.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { float v[176]; };   // 704 bytes
__global__ void k_empty(float* p) { if (p == nullptr) asm volatile(""); }
__global__ void k_big(Big b, float* p) { if (b.v[17] == 123.f) p[0] = 1.f; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; if (p == nullptr) sm[threadIdx.x] = 0.f; }
__global__ void k_atom(double* acc) { if (threadIdx.x < 64) atomicAdd(acc + threadIdx.x, 1.0); }
__global__ void k_spin(Big b, long ticks, float* p) {      // every wave busy for `ticks` of the 100 MHz wall clock: the host runs far ahead, what is left is the GPU's own boundary
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (b.v[17] == 123.f) p[0] = 1.f;
}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float* buf; double* acc;
    CHK(hipMalloc(&buf, 64 << 20)); CHK(hipMalloc(&acc, 4096)); CHK(hipMemset(buf, 0, 64 << 20)); CHK(hipMemset(acc, 0, 4096));
    hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int N = 2000;
    Big big{}; 
    auto run = [&](const char* name, auto launch) -> int {
        for (int i = 0; i < 50; ++i) launch();
        CHK(hipStreamSynchronize(s));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(a, s));
            for (int i = 0; i < N; ++i) launch();
            CHK(hipEventRecord(b, s));
            CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("%-64s %6.2f us per launch\n", name, 1e3f * best / N);
        return 0;
    };
    for (int g : {1, 12, 48, 256, 1024, 4096}) {
        char nm[96]; snprintf(nm, sizeof nm, "empty kernel, %d workgroups x 256", g);
        run(nm, [&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, buf); });
    }
    run("empty kernel, 256 x 256, 704-byte by-value argument", [&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big, buf); });
    run("empty kernel, 48 x 256, 704-byte by-value argument", [&] { hipLaunchKernelGGL(k_big, dim3(48), dim3(256), 0, s, big, buf); });
    CHK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    run("empty kernel, 256 x 256, 128 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 128 << 10, s, buf); });
    run("empty kernel, 256 x 512, 128 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 128 << 10, s, buf); });
    run("64 same-address fp64 atomics per block, 48 blocks", [&] { hipLaunchKernelGGL(k_atom, dim3(48), dim3(256), 0, s, acc); });
    run("64 same-address fp64 atomics per block, 256 blocks", [&] { hipLaunchKernelGGL(k_atom, dim3(256), dim3(256), 0, s, acc); });
    run("64 same-address fp64 atomics per block, 1024 blocks", [&] { hipLaunchKernelGGL(k_atom, dim3(1024), dim3(256), 0, s, acc); });
    for (int mb : {1, 4, 16}) {
        char nm[96]; snprintf(nm, sizeof nm, "read-modify-write of %d MB (dirty lines at the boundary)", mb);
        int n = mb << 18;
        run(nm, [&] { hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, s, buf, n); });
    }
    for (int g : {1, 48, 256, 768}) for (long us : {10, 20}) {
        char nm[128]; snprintf(nm, sizeof nm, "GPU-side boundary: %d workgroups x 256 busy for %ld us, 704-byte argument  (minus the busy time)", g, us);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_spin, dim3(g), dim3(256), 0, s, big, us * 100, buf);
        CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(a, s));
        for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k_spin, dim3(g), dim3(256), 0, s, big, us * 100, buf);
        CHK(hipEventRecord(b, s));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        printf("%-118s %6.2f us per launch\n", nm, 1e3f * ms / 500 - (float)us);
    }
    // a burst of trivial kernels behind ONE long kernel (the host has enqueued all of them before the long one ends): GPU-side cost per trivial kernel
    for (int g : {1, 48, 256}) {
        float ms0, ms1;
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, big, 300000L, buf);      // 3 ms
        CHK(hipEventRecord(a, s));
        CHK(hipEventRecord(b, s));
        CHK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, big, 300000L, buf);      // 3 ms
        CHK(hipEventRecord(a, s));
        for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, s, big, buf);
        CHK(hipEventRecord(b, s));
        CHK(hipEventSynchronize(b));
        CHK(hipEventElapsedTime(&ms1, a, b));
        (void)ms0;
        printf("burst of 400 trivial kernels (%3d workgroups, 704-byte argument) queued behind a 3 ms kernel: %6.2f us per kernel on the GPU\n", g, 1e3f * ms1 / 400);
    }
    // the same empty chain on the NULL stream
    hipStream_t keep = s; s = nullptr;
    run("empty kernel, 256 x 256, NULL stream", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, buf); });
    s = keep;
    return 0;
}
