// Probe (tools only, not part of the library): what does a grid-wide barrier cost on this GPU, against a dependent
// kernel boundary?  Decides whether a persistent "bottom of the U" launch (one co-resident grid walking the r <= 16
// nodes behind grid barriers) can beat one launch per node.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/grid_barrier_probe.bin tools/probes/grid_barrier_probe.hip
//   tools/probes/grid_barrier_probe.bin
// Prints, for G = 32 / 64 / 128 / 256 workgroups of 256 threads (one per CU):
//   * XCD-hierarchical barrier (per-XCD arrival counter -> top counter -> per-XCD generation word), us per barrier,
//     empty phases and phases that publish 4 KB per workgroup and read a neighbour's 4 KB (checked);
//   * single-counter barrier;
//   * a chain of dependent launches of a trivial kernel of the same grid (us per boundary, eager).
// Every spin is bounded: a barrier that does not complete sets a timeout word and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct BarState {
    unsigned xcd_cnt[8][32];     // one 128-byte line per XCD
    unsigned top[32];
    unsigned xcd_gen[8][32];
    unsigned timeout[32];
    unsigned one[32];
};

constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-hierarchical barrier; gen counts completed barriers (0, 1, ...); nx = workgroups on this XCD; nxcd = XCDs in use
__device__ __forceinline__ void barrier_xcd(BarState* st, int xcd, unsigned nx, unsigned nxcd, unsigned gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(&st->xcd_cnt[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == nx * (gen + 1)) {               // last arriver of this XCD: the leader
            __hip_atomic_fetch_add(&st->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld_relaxed(&st->top[0]) < nxcd * (gen + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) { st->timeout[0] = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&st->xcd_gen[xcd][0], gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (ld_relaxed(&st->xcd_gen[xcd][0]) < gen + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) { st->timeout[0] = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_one(BarState* st, unsigned G, unsigned gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&st->one[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_relaxed(&st->one[0]) < G * (gen + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_MAX) { st->timeout[0] = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// mode 0: xcd barrier, empty phases; 1: xcd barrier, publish 4 KB + read the neighbour's (checked); 2: single counter
__global__ __launch_bounds__(256) void barrier_kernel(BarState* st, float* slab, unsigned* errors, int nbar, int mode) {
    const int G = gridDim.x, b = blockIdx.x, xcd = b & 7;
    const unsigned nxcd = G < 8 ? G : 8;
    const unsigned nx = (G - xcd + 7) / 8;
    unsigned bad = 0;
    for (int it = 0; it < nbar; ++it) {
        if (mode == 1) {
            float4* mine = reinterpret_cast<float4*>(slab + ((size_t)(it & 1) * G + b) * 1024);
            mine[threadIdx.x] = make_float4((float)(it * 1000 + b), 1.f, 2.f, 3.f);
        }
        if (mode == 2) barrier_one(st, G, it); else barrier_xcd(st, xcd, nx, nxcd, it);
        if (ld_relaxed(&st->timeout[0])) return;
        if (mode == 1) {
            const int nb = (b + 37) % G;                        // a workgroup of another XCD
            const float4 v = reinterpret_cast<const float4*>(slab + ((size_t)(it & 1) * G + nb) * 1024)[threadIdx.x];
            if (v.x != (float)(it * 1000 + nb)) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(256) void trivial_kernel(float* slab) {
    if (threadIdx.x == 0) slab[blockIdx.x * 1024] += 1.f;
}

int main() {
    BarState* st; float* slab; unsigned* errors;
    CHK(hipMalloc(&st, sizeof(BarState)));
    CHK(hipMalloc(&slab, (size_t)2 * 256 * 1024 * 4));
    CHK(hipMalloc(&errors, 4));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipStream_t s; CHK(hipStreamCreate(&s));
    const int NBAR = 200;
    for (int G : {32, 64, 128, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f; unsigned err = 0, to = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipMemsetAsync(st, 0, sizeof(BarState), s));
                CHK(hipMemsetAsync(errors, 0, 4, s));
                CHK(hipEventRecord(a, s));
                hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, s, st, slab, errors, NBAR, mode);
                CHK(hipEventRecord(b, s));
                CHK(hipStreamSynchronize(s));
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                if (rep > 0 && ms < best) best = ms;
                unsigned e; CHK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost)); err += e;
                BarState h; CHK(hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost)); to += h.timeout[0];
            }
            printf("G=%3d %-34s %6.2f us per barrier (kernel of %d barriers: %.1f us) errors=%u timeouts=%u\n", G,
                   mode == 0 ? "xcd barrier, empty phases" : mode == 1 ? "xcd barrier, 4 KB publish + read" : "single-counter barrier",
                   1e3f * best / NBAR, NBAR, 1e3f * best, err, to);
        }
        {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipStreamSynchronize(s));
                CHK(hipEventRecord(a, s));
                for (int i = 0; i < NBAR; ++i) hipLaunchKernelGGL(trivial_kernel, dim3(G), dim3(256), 0, s, slab);
                CHK(hipEventRecord(b, s));
                CHK(hipStreamSynchronize(s));
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("G=%3d %-34s %6.2f us per launch (chain of %d dependent launches, eager)\n", G, "trivial kernel chain", 1e3f * best / NBAR, NBAR);
        }
    }
    return 0;
}
