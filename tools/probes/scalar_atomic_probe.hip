// Does gfx950 execute scalar-memory atomics (s_atomic_add ... glc: returns the old value through lgkmcnt, not vmcnt)?  (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sap tools/probes/scalar_atomic_probe.hip && /tmp/sap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* ctr, unsigned* out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    const int N = 4096;
    unsigned *ctr, *out;
    if (hipMalloc(&ctr, 256) != hipSuccess || hipMalloc(&out, N * 4) != hipSuccess) return 1;
    hipMemset(ctr, 0, 256);
    hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, ctr, out);      // one wave per block: one scalar atomic per block
    std::vector<unsigned> h(N);
    unsigned c = 0;
    if (hipMemcpy(h.data(), out, N * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == (unsigned)N;
    for (int i = 0; i < N; ++i) ok = ok && h[i] == (unsigned)i;
    printf("s_atomic_add on gfx950: counter %u (expected %d), returned values are %s permutation of 0..N-1\n", c, N, ok ? "a" : "NOT a");
    return ok ? 0 : 2;
}
