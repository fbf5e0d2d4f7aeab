// Probe of the two gfx950 data-movement instructions the round-4 weight-gradient kernel relies on (run on the GPU box):
//   (1) ds_read_b64_tr_b16: which four bf16 elements does lane l receive for a given per-lane LDS address?
//   (2) global_load_lds_dwordx4: where in LDS do the 16 bytes a lane requests land (M0 base, instruction offset)?
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/probes/cdna4_lds_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void tr_probe(unsigned short* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // hypothesis (guide): a 16-lane group reads a [4 k][16 col] row-major block; lane (l & 15) gets column l & 15, rows 0..3.
    // Address given per lane: base of ITS group's block + its own column?  Try: addr = ((l >> 4) * 4 * pitch + (l & 15)) elements.
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)(((l >> 4) * 4 * pitch_elems + (l & 15)) * 2);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[l * 4 + 1] = (unsigned short)(v.x >> 16);
    out[l * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[l * 4 + 3] = (unsigned short)(v.y >> 16);
}

// variant: every lane of a 16-lane group passes the SAME address (the block base): what comes back?
__global__ void tr_probe_uniform(unsigned short* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)(((l >> 4) * 4 * pitch_elems) * 2);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[l * 4 + 1] = (unsigned short)(v.x >> 16);
    out[l * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[l * 4 + 3] = (unsigned short)(v.y >> 16);
}

// variant: lane l passes the address of ITS OWN row-major 8-byte piece: row (l & 3) + 4 * (l >> 4)?, i.e. natural "each lane points at
// 4 contiguous bf16" addressing: addr = row r = (l >> 4) * 4 + ((l & 15) >> 2), col 4 * (l & 3)
__global__ void tr_probe_rowpiece(unsigned short* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int r = (l >> 4) * 4 + ((l & 15) >> 2), c = 4 * (l & 3);
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)((r * pitch_elems + c) * 2);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[l * 4 + 1] = (unsigned short)(v.x >> 16);
    out[l * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[l * 4 + 3] = (unsigned short)(v.y >> 16);
}

__global__ void glds_probe(const unsigned* src, unsigned* out, int lds_base_bytes, int reverse) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l asks for the 16 bytes at src + 4 * (reverse ? 63 - l : l) dwords
    const unsigned* g = src + 4 * (reverse ? 63 - l : l);
    const unsigned m0v = (unsigned)(size_t)lds + (unsigned)lds_base_bytes;      // LDS byte address of the wave's 1 KB destination (the array starts at LDS 0 here)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(m0v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    for (int variant = 0; variant < 3; ++variant)
        for (int pitch : {16, 64}) {
            if (variant == 0) hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, pitch);
            else if (variant == 1) hipLaunchKernelGGL(tr_probe_uniform, dim3(1), dim3(64), 0, 0, d, pitch);
            else hipLaunchKernelGGL(tr_probe_rowpiece, dim3(1), dim3(64), 0, 0, d, pitch);
            if (hipDeviceSynchronize() != hipSuccess) { printf("tr probe failed\n"); return 1; }
            hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
            printf("ds_read_b64_tr_b16 variant %d (0 = own column, 1 = group-uniform base, 2 = own row piece), pitch %d elements: lane -> 4 x (row, col)\n", variant, pitch);
            for (int l = 0; l < 64; ++l) {
                printf("  l%02d:", l);
                for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
                if ((l & 3) == 3) printf("\n");
            }
        }
    unsigned *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 4096 * 4);
    std::vector<unsigned> hs(4096), ho(4096);
    for (int i = 0; i < 4096; ++i) hs[i] = 0x1000 + i;
    hipMemcpy(src, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int rev = 0; rev < 2; ++rev)
        for (int base : {0, 2048}) {
            hipLaunchKernelGGL(glds_probe, dim3(1), dim3(64), 0, 0, src, out, base, rev);
            if (hipDeviceSynchronize() != hipSuccess) { printf("glds probe failed\n"); return 1; }
            hipMemcpy(ho.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
            printf("global_load_lds_dwordx4, M0 = %d, lane order %s: LDS dword index -> source dword index (only written dwords)\n", base, rev ? "reversed" : "identity");
            int shown = 0;
            for (int i = 0; i < 4096 && shown < 24; ++i)
                if (ho[i] != 0xdeadbeefu) { printf("  lds[%d] <- src[%d]\n", i, (int)ho[i] - 0x1000); ++shown; if (shown == 8) { i = (i / 64 + 1) * 64 * 2 - 1; } }
            int written = 0, lo = 1 << 30, hi2 = -1;
            for (int i = 0; i < 4096; ++i) if (ho[i] != 0xdeadbeefu) { ++written; if (i < lo) lo = i; if (i > hi2) hi2 = i; }
            bool linear = true;
            for (int i = lo; i <= hi2 && lo <= hi2; ++i) { const int k = i - lo; const int lane = k / 4; if ((int)ho[i] - 0x1000 != 4 * (rev ? 63 - lane : lane) + (k & 3)) linear = false; }
            printf("  %d dwords written, LDS dword range [%d, %d], lane-linear destination (M0 base + 16 * lane): %s\n", written, lo, hi2, linear ? "yes" : "NO");
        }
    return 0;
}
