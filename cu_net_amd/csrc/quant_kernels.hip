// Weight / gradient quantisers and the multiplier-free ternary convolution.
//
// Replaces, on the flat parameter / gradient arenas (one launch per phase for ALL target convs):
//   utils/quantize.py:104-149  QuanOp.quantization  (mean-centre, clamp, save, quantise)
//   utils/quantize.py:151-153  QuanOp.restore
//   utils/quantize.py:156-175  QuanOp.updateQuanGradWeight
//   models/cu_net_prev_version.py:45-92  BinOp (keep_scale = 1, bits_g = 32)
// and, as the non-MFMA alternative for conv with weights in {-1,0,+1} on 8-bit activations
// (QuanInput2d placement, models/cu_net_prev_version_wig.py:96-98,277-279), an AND-popcount kernel.
#include "common.h"

namespace cunet {


__device__ __forceinline__ float q_scale(int bits) { return exp2f((float)(bits - 1)); }     // S(bits)
__device__ __forceinline__ float q_clamp(float x, int bits) {                                // C(x, bits)
    const float delta = (bits > 15 || bits == 1 || bits == 2) ? 0.f : 1.f / q_scale(bits);
    return fminf(fmaxf(x, -1.f + delta), 1.f - delta);
}
__device__ __forceinline__ float q_sign(float x) { return (float)((x > 0.f) - (x < 0.f)); }
__device__ __forceinline__ float q_round(float x, int bits) {                                // Q(x, bits)
    if (bits > 15) return x;
    if (bits == 1) return q_sign(x);
    if (bits == 2) return rintf(x);                 // torch.round: half to even
    const float sc = q_scale(bits);
    return rintf(x * sc) / sc;
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {     // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// One block per (conv, filter).  The filter (n = I*KK floats) lives in LDS while it is processed.
__global__ __launch_bounds__(256) void quant_prepare_kernel(const QuantEntry* tab, float* params, float* saved,
                                                            int bits_w, int bits_g, int keep_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = reinterpret_cast<float*>(smem);
    __shared__ float scratch[4];
    __shared__ float pmean[64];
    const QuantEntry e = tab[blockIdx.y];
    const int o = blockIdx.x;
    if (o >= e.O) return;
    const int n = e.I * e.KK;
    float* src = params + e.off + (size_t)o * n;
    float* sv = saved + e.off + (size_t)o * n;
    for (int i = threadIdx.x; i < n; i += 256) w[i] = src[i];
    __syncthreads();
    // mean over the INPUT channels for every kernel position (W.mean(1, keepdim))
    for (int p = threadIdx.x >> 6; p < e.KK; p += 4) {       // one wave per position
        float s = 0.f;
        for (int i = threadIdx.x & 63; i < e.I; i += 64) s += w[i * e.KK + p];
        for (int of = 32; of > 0; of >>= 1) s += __shfl_xor(s, of, 64);
        if ((threadIdx.x & 63) == 0) pmean[p] = s / (float)e.I;
    }
    __syncthreads();
    float asum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float c = q_clamp(w[i] + (-pmean[i % e.KK]), bits_g);
        w[i] = c;
        sv[i] = q_round(c, bits_g);
        asum += fabsf(c);
    }
    const float m = block_sum(asum, scratch) / (float)n;     // mean |W| of the filter
    for (int i = threadIdx.x; i < n; i += 256) {
        const float c = w[i];
        float out;
        if (bits_w == 1) {
            const float mq = q_round(m, bits_g);
            const float t = q_sign(c) * mq;
            out = keep_scale ? t : q_round(q_clamp(t, 1), 1);         // the reference falls through to Q(C(.,1),1)
        } else if (bits_w == 2) {
            const float d = m * 0.7f;
            out = (float)(c > d) + -1.f * (float)(c < -d);
        } else {
            out = q_round(q_clamp(c, bits_w), bits_w);
        }
        src[i] = out;
    }
}

__global__ __launch_bounds__(256) void quant_restore_kernel(const QuantEntry* tab, float* params, const float* saved) {
    const QuantEntry e = tab[blockIdx.y];
    const long n = (long)e.O * e.I * e.KK;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        params[e.off + i] = saved[e.off + i];
}

__global__ __launch_bounds__(256) void quant_grad_kernel(const QuantEntry* tab, const float* params, float* grads,
                                                         int bits_w, int bits_g, int keep_scale) {
    __shared__ float scratch[4];
    const QuantEntry e = tab[blockIdx.y];
    const int o = blockIdx.x;
    if (o >= e.O) return;
    const int n = e.I * e.KK;
    const float* w = params + e.off + (size_t)o * n;
    float* g = grads + e.off + (size_t)o * n;
    if (bits_w != 1) {
        if (!keep_scale)
            for (int i = threadIdx.x; i < n; i += 256) g[i] = q_round(q_clamp(g[i], bits_g), bits_g);
        return;
    }
    float asum = 0.f, ssum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        asum += fabsf(w[i]);
        ssum += q_sign(w[i]) * g[i];
    }
    const float m = block_sum(asum, scratch) / (float)n;
    const float sadd = block_sum(ssum, scratch) / (float)n;
    const float mq = q_round(m, bits_g);
    const float c1 = (float)(1.0 - 1.0 / (double)e.I);
    for (int i = threadIdx.x; i < n; i += 256) {
        const float wi = w[i];
        const float mi = (wi < -1.f || wi > 1.f) ? q_round(0.f, bits_g) : mq;
        float v = ((mi * g[i] + sadd * q_sign(wi)) * c1) * (float)n;
        if (!keep_scale) v = q_round(q_clamp(v, bits_g), bits_g);     // BinOp has no gradient rounding
        g[i] = v;
    }
}

hipError_t launch_quant_prepare(const QuantEntry* tab, int nconv, int maxO, int maxN, float* params, float* saved,
                                int bits_w, int bits_g, int keep_scale, hipStream_t s) {
    hipLaunchKernelGGL(quant_prepare_kernel, dim3(maxO, nconv), dim3(256), (size_t)maxN * 4, s, tab, params, saved,
                       bits_w, bits_g, keep_scale);
    return hipGetLastError();
}
hipError_t launch_quant_restore(const QuantEntry* tab, int nconv, float* params, const float* saved, hipStream_t s) {
    hipLaunchKernelGGL(quant_restore_kernel, dim3(16, nconv), dim3(256), 0, s, tab, params, saved);
    return hipGetLastError();
}
hipError_t launch_quant_grad(const QuantEntry* tab, int nconv, int maxO, const float* params, float* grads,
                             int bits_w, int bits_g, int keep_scale, hipStream_t s) {
    hipLaunchKernelGGL(quant_grad_kernel, dim3(maxO, nconv), dim3(256), 0, s, tab, params, grads, bits_w, bits_g, keep_scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ternary convolution by AND + popcount (1x1 or 3x3, NHWC):
//     y[p][o] = sum_{tap,c} w[o][c][tap] * a[p (+) tap][c],  w in {-1,0,+1},
//     a = QuanInput_b(relu(x*scale + shift)) = q / 2^(b-1),  q in [0, 2^(b-1) - 1]
//   => y * 2^(b-1) = sum_bit 2^bit * ( popc(P_o & plane_bit) - popc(N_o & plane_bit) )
// Lanes = output channels (their +1 / -1 masks P_o, N_o stay in registers); the wave walks pixels: 64
// lanes read 64 channels of one pixel (one coalesced 256-B row piece), quantise, and 7 wave ballots
// turn the 64 values into 7 uniform 64-bit bit-planes held in scalar registers.  No multiplier and no
// MFMA is used; the result is exact integer arithmetic, bit-identical to an fp32 convolution of the
// quantised activations (every partial sum is a multiple of 2^-7 below 2^17).
constexpr int TC_MAXG1 = 6;     // 1x1: 64-channel groups kept in registers (C <= 384)
constexpr int TC_MAXG9 = 2;     // 3x3: C <= 128 (9 taps x 2 groups x 2 masks)


template <int TAPS, int MAXG>
__global__ __launch_bounds__(256) void ternary_conv_kernel(const TernArgs p) {
    __shared__ float s_sc[64 * MAXG], s_sh[64 * MAXG];
    __shared__ double s_red[2][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = (p.C + 63) >> 6;
    // ---- BatchNorm folded to scale / shift: given (stand-alone operator) or derived here from the input tensor's batch
    //      statistics (training) / the running statistics (eval), exactly as every other consumer kernel derives them
    for (int c = tid; c < 64 * MAXG; c += 256) {
        float sc = 0.f, sh = 0.f;
        if (c < p.C) {
            if (p.scale != nullptr) { sc = p.scale[c]; sh = p.shift[c]; }
            else {
                double mean, istd;
                if (p.training) {
                    const double sum = p.xstats[c], sq = p.xstats[p.C + c];
                    mean = sum / p.count;
                    double var = sq / p.count - mean * mean;
                    var = var < 0.0 ? 0.0 : var;
                    istd = 1.0 / sqrt(var + (double)BN_EPS);
                } else {
                    mean = (double)p.rmean[c];
                    istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
                }
                const double scale = (double)p.gamma[c] * istd;
                sc = (float)scale;
                sh = (float)((double)p.beta[c] - mean * scale);
            }
        }
        s_sc[c] = sc; s_sh[c] = sh;
    }
    if (tid < 128) s_red[tid >> 6][tid & 63] = 0.0;
    __syncthreads();

    const int o = blockIdx.y * 64 + lane;                 // this lane's output channel
    uint64_t P[TAPS][MAXG], N[TAPS][MAXG];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            P[t][g] = 0; N[t][g] = 0;
            if (g < G && o < p.Opad) {
                P[t][g] = p.wpos[((size_t)t * G + g) * p.Opad + o];
                N[t][g] = p.wneg[((size_t)t * G + g) * p.Opad + o];
            }
        }
    const int nb = p.bits_i - 1;                           // bit-planes (7 for 8-bit inputs)
    const float qs = exp2f((float)nb);
    const int HW = p.H * p.W;
    const int nwaves = gridDim.x * 4;
    double d1 = 0.0, d2 = 0.0;
    for (int m = blockIdx.x * 4 + wave; m < p.M; m += nwaves) {
        const int ni = m / HW;
        const int rem = m - ni * HW;
        const int py = rem / p.W, px = rem - py * p.W;
        int acc = 0;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            int row = m;
            bool valid = true;
            if (TAPS == 9) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const int yy = py + dy, xx = px + dx;
                valid = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;      // zero padding is post-activation
                row = valid ? m + dy * p.W + dx : m;
            }
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                if (g >= G) break;
                const int c = 64 * g + tern_chan_of_bit(lane);             // (ballot bit `lane` = this channel: the mask words' bit order)
                int q = 0;
                if (valid && c < p.C) {
                    float a = fmaxf(fmaf(p.x[(size_t)row * p.ldx + c], s_sc[c], s_sh[c]), 0.f);
                    a = fminf(a, 1.f - 1.f / qs);                          // C(x, bits_i); relu already >= 0
                    q = (int)rintf(a * qs);                                // Q(x, bits_i) * 2^(bits_i-1)
                }
                for (int b = 0; b < nb; ++b) {
                    const uint64_t plane = __ballot((q >> b) & 1);         // uniform: bit c = bit b of channel 64g+c
                    const int d = __popcll(P[t][g] & plane) - __popcll(N[t][g] & plane);
                    acc += d * (1 << b);
                }
            }
        }
        const float yv = (float)acc / qs;
        if (o < p.O) {
            p.y[(size_t)m * p.ldy + o] = yv;
            d1 += (double)yv;
            d2 += (double)yv * (double)yv;
        }
    }
    if (p.ystats != nullptr) {        // batch statistics of the output for the consumer BatchNorms
        atomicAdd(&s_red[0][lane], d1);
        atomicAdd(&s_red[1][lane], d2);
        __syncthreads();
        if (tid < 64 && o < p.O) {
            __hip_atomic_fetch_add(p.ystats + o, s_red[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ystats + p.O + o, s_red[1][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second-generation AND-popcount forward (plan path, C <= 128, bits_i <= 8): the activation is quantised and cut into bit-planes
// ONCE per tensor instead of once per tap, and a wave works on 64 / LP pixels at a time.
//   ternary_planes_kernel        x -> BatchNorm -> ReLU -> QuanInput -> one 128-byte record per pixel (TERN_REC_WORDS)
//   ternary_conv_planes_kernel   a wave per pixel, plane words broadcast by v_readlane, + / - masks on the two halves of the wave (see there)
// Every quantity is an integer below 2^24: the result is exact, bit-identical to the MFMA forward of the same node.
__global__ __launch_bounds__(256) void ternary_planes_kernel(const TernArgs p) {
    __shared__ float s_sc[128], s_sh[128];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = (p.C + 63) >> 6;
    for (int c = tid; c < 128; c += 256) {
        float sc = 0.f, sh = 0.f;
        if (c < p.C && p.scale != nullptr) {                  // (the stand-alone operator: a folded scale / shift handed in)
            sc = p.scale[c]; sh = p.shift[c];
        } else if (c < p.C) {
            double mean, istd;
            if (p.training) {
                mean = p.xstats[c] / p.count;
                double var = p.xstats[p.C + c] / p.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                istd = 1.0 / sqrt(var + (double)BN_EPS);
            } else {
                mean = (double)p.rmean[c];
                istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
            }
            const double scale = (double)p.gamma[c] * istd;
            sc = (float)scale;
            sh = (float)((double)p.beta[c] - mean * scale);
        }
        s_sc[c] = sc; s_sh[c] = sh;
    }
    __syncthreads();
    const int nb = p.bits_i - 1;
    const float qs = exp2f((float)nb);
    if (blockIdx.x == 0 && wave == 0 && lane < TERN_REC_WORDS) p.planes[(size_t)p.M * TERN_REC_WORDS + lane] = 0;      // the zero record
    const int nwaves = gridDim.x * 4;
    for (int m0 = (blockIdx.x * 4 + wave) * 2; m0 < p.M; m0 += nwaves * 2) {      // two pixels per iteration: four loads in flight
        float xv[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int m = m0 + u < p.M ? m0 + u : m0;
                const int c = 64 * g + tern_chan_of_bit(lane);             // (ballot bit `lane` = this channel)
                xv[u][g] = (g < G && c < p.C) ? ldg1(p.x + (size_t)m * p.ldx + c) : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (m0 + u >= p.M) break;
            uint64_t mine = 0;                  // lanes 0..13 end up holding plane word (7 g + b), lane 14 the sum
            long long ssum = 0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 64 * g + tern_chan_of_bit(lane);
                int q = 0;
                if (g < G && c < p.C) {
                    float a = fmaxf(fmaf(xv[u][g], s_sc[c], s_sh[c]), 0.f);
                    a = fminf(a, 1.f - 1.f / qs);                          // C(x, bits_i); relu already >= 0
                    q = (int)rintf(a * qs);                                // Q(x, bits_i) * 2^(bits_i-1)
                }
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const uint64_t plane = b < nb ? __ballot((q >> b) & 1) : 0;
                    if (lane == 7 * g + b) mine = plane;
                    ssum += (long long)__popcll(plane) << b;
                }
            }
            if (lane == 14) mine = (uint64_t)ssum;
            if (lane < TERN_REC_WORDS) p.planes[(size_t)(m0 + u) * TERN_REC_WORDS + lane] = mine;
        }
    }
}

// The plane records with LANE = PIXEL (round 5, with the lane-per-pixel counting kernel).  The ballot kernel above spends ~150 wave
// instructions per pixel (a wave per pixel: 14 ballots, 14 scalar popcounts, the selects that route words to lanes 0..14) -- 22 us per launch
// on average, 3.5 ms of a CU-Net-16 step, as much as the counting it serves.  Here a wave takes 64 pixels: phase 1, lanes along CHANNELS
// (a 16-byte piece of 4 channels per lane, two pixels per load instruction: coalesced), BatchNorm + ReLU + QuanInput, the four quantised
// bytes packed into a dword of a wave-private LDS tile [64 pixels][32 dwords + 1]; phase 2, lane = pixel: its 32 dwords come back by
// conflict-free ds_read_b32, and bit b of the four channels of dword k lands in plane b by (d >> b) & 0x01010101 shifted left by k -- three
// operations per dword and plane, the bit order of tern_bit_of_chan (common.h).  The pixel sum is v_sad_u8 over the same dwords.
// ~23 instead of ~150 wave instructions per pixel; the records are bit-identical to the ballot kernel's.
__global__ __launch_bounds__(256) void ternary_planes_rows_kernel(const TernArgs p) {
    __shared__ float s_sc[128], s_sh[128];
    __shared__ unsigned s_q[4][64 * 33];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int c = tid; c < 128; c += 256) {
        float sc = 0.f, sh = 0.f;
        if (c < p.C && p.scale != nullptr) {
            sc = p.scale[c]; sh = p.shift[c];
        } else if (c < p.C) {
            double mean, istd;
            if (p.training) {
                mean = p.xstats[c] / p.count;
                double var = p.xstats[p.C + c] / p.count - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                istd = 1.0 / sqrt(var + (double)BN_EPS);
            } else {
                mean = (double)p.rmean[c];
                istd = 1.0 / sqrt((double)p.rvar[c] + (double)BN_EPS);
            }
            const double scale = (double)p.gamma[c] * istd;
            sc = (float)scale;
            sh = (float)((double)p.beta[c] - mean * scale);
        }
        s_sc[c] = sc; s_sh[c] = sh;
    }
    __syncthreads();
    const int nb = p.bits_i - 1;
    const float qs = exp2f((float)nb);
    const float lim = 1.f - 1.f / qs;
    if (blockIdx.x == 0 && tid < 2 * TERN_REC_WORDS) reinterpret_cast<unsigned*>(p.planes + (size_t)p.M * TERN_REC_WORDS)[tid] = 0u;      // the zero record
    const int c4 = 4 * (lane & 31);                       // phase 1: this lane's four channels (the same in every load)
    const bool chan_ok = c4 < p.C;                         // (C is a multiple of 4: a piece is inside or outside as a whole)
    const float4 sc4 = *reinterpret_cast<const float4*>(s_sc + c4), sh4 = *reinterpret_cast<const float4*>(s_sh + c4);
    unsigned* Q = s_q[wave];
    const int groups = (p.M + 63) >> 6;
    for (int grp = blockIdx.x * 4 + wave; grp < groups; grp += gridDim.x * 4) {
        const int m0 = grp * 64;
        // ---- phase 1: quantise, pack, LDS (eight loads in flight per lane)
#pragma unroll
        for (int j0 = 0; j0 < 32; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = m0 + 2 * (j0 + u) + (lane >> 5);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (chan_ok && m < p.M) v[u] = ldg4(p.x + (size_t)m * p.ldx + c4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                const float s4[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, h4[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
                unsigned packed = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = fmaxf(fmaf(f[e], s4[e], h4[e]), 0.f);
                    a = fminf(a, lim);                                     // C(x, bits_i); relu already >= 0
                    packed |= (unsigned)(int)rintf(a * qs) << (8 * e);     // Q(x, bits_i) * 2^(bits_i-1): at most 127
                }
                Q[(2 * (j0 + u) + (lane >> 5)) * 33 + (lane & 31)] = packed;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: lane = pixel m0 + lane
        unsigned rec[2 * TERN_REC_WORDS];
#pragma unroll
        for (int i = 0; i < 2 * TERN_REC_WORDS; ++i) rec[i] = 0u;
        unsigned ssum = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {                       // 32-channel group h: word 7 (h >> 1) + b, half h & 1
            unsigned A[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned d = Q[lane * 33 + 8 * h + k];
                ssum = __builtin_amdgcn_sad_u8(d, 0u, ssum);
#pragma unroll
                for (int b = 0; b < 7; ++b) A[b] |= ((d >> b) & 0x01010101u) << k;
            }
#pragma unroll
            for (int b = 0; b < 7; ++b) rec[2 * (7 * (h >> 1) + b) + (h & 1)] = A[b];
        }
        rec[28] = ssum;                                    // word 14: the pixel's sum of quantised activations
        if (m0 + lane < p.M) {
            uint4* dst = reinterpret_cast<uint4*>(p.planes + (size_t)(m0 + lane) * TERN_REC_WORDS);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = make_uint4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the next group's bytes overwrite the tile)
        __builtin_amdgcn_wave_barrier();
    }
}

// A wave owns ONE pixel at a time.  Its nine neighbour records (32 dwords each) arrive as FIVE coalesced vector loads -- lane l reads
// dword l & 31 of the record of tap 2k + (l >> 5) -- requested one pixel ahead; a plane word then reaches all lanes as a scalar operand
// through v_readlane.  (History: per-lane 16-byte loads of the records sat on the vector-memory issue rate, 72 load instructions per
// pixel; s_load through the scalar cache had nothing in flight behind its latency: 120 us per 64 x 64 launch either way.)
// Lane = (output channel o = lane & 31, sign = lane >> 5): lanes 0..31 hold the +1 masks of their output channel and count
// popc(P & plane), lanes 32..63 the -1 masks and count popc(N & plane); one cross-lane subtraction at the end gives
// sum_b 2^b (popc(P & plane_b) - popc(N & plane_b)).  32 output channels per block column (blockIdx.y).
// 16 waves per block, about one block per CU: every block ends in 64 fp64 atomics on the SAME addresses, which the L2 serialises at
// ~12 ns each -- with 2048 four-wave blocks that tail alone was 40 - 50 us, more than the counting.
constexpr int TCP_THREADS = 1024;

template <int TAPS>
__global__ __launch_bounds__(TCP_THREADS) void ternary_conv_planes_kernel(const TernArgs p) {
    __shared__ double s_red[2][32];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ol = lane & 31;
    const int sgn = lane >> 5;
    const int o = blockIdx.y * 32 + ol;                   // this lane's output channel
    const int G = (p.C + 63) >> 6;
    if (tid < 64) s_red[tid >> 5][tid & 31] = 0.0;
    __syncthreads();

    unsigned Ml[TAPS][2], Mh[TAPS][2];                    // this lane's masks (+1 masks for sgn = 0, -1 masks for sgn = 1)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint64_t mm = 0;
            if (g < G && o < p.Opad) mm = (sgn ? p.wneg : p.wpos)[((size_t)t * G + g) * p.Opad + o];
            Ml[t][g] = (unsigned)mm; Mh[t][g] = (unsigned)(mm >> 32);
        }
    const float qs = exp2f((float)(p.bits_i - 1));
    const int HW = p.H * p.W;
    const int stride = gridDim.x * (TCP_THREADS / 64);
    const unsigned* planes = reinterpret_cast<const unsigned*>(p.planes);
    constexpr int NL = (TAPS + 1) / 2;                    // record loads per pixel
    auto fetch = [&](int m, unsigned (&v)[NL]) {          // m is wave-uniform
        const int ni = m / HW;
        const int rem = m - ni * HW;
        const int py = rem / p.W, px = rem - py * p.W;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            int row = m;
            if (TAPS == 9) {
                const int t = 2 * k + sgn;                // (t = 9 for the upper half of the last load: the zero record)
                const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
                const int yy = py + dy, xx = px + dx;
                const bool valid = t < 9 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;      // zero padding is post-activation
                row = valid ? m + dy * p.W + dx : p.M;                                      // record M is all zero
            }
            v[k] = (unsigned)__builtin_nontemporal_load(planes + (size_t)row * (TERN_REC_WORDS * 2) + ol);
        }
    };
    double d1 = 0.0, d2 = 0.0;
    unsigned vnext[NL];
    int m = blockIdx.x * (TCP_THREADS / 64) + wave;
    if (m < p.M) fetch(m, vnext);
    for (; m < p.M; m += stride) {
        unsigned v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = vnext[k];
        if (m + stride < p.M) fetch(m + stride, vnext);   // the next pixel's records are in flight during this pixel's counting
        int cnt[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int k = t >> 1, base = (t & 1) * 32;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const unsigned vl = (unsigned)__builtin_amdgcn_readlane((int)v[k], base + 2 * (7 * g + b));
                    const unsigned vh = (unsigned)__builtin_amdgcn_readlane((int)v[k], base + 2 * (7 * g + b) + 1);
                    cnt[b] += __popc(Ml[t][g] & vl) + __popc(Mh[t][g] & vh);
                }
        }
        int acc = 0;
#pragma unroll
        for (int b = 0; b < 7; ++b) acc += cnt[b] << b;
        const int other = __shfl_xor(acc, 32, 64);
        const float yv = (float)(acc - other) / qs;          // (lanes 0..31: P - N)
        if (sgn == 0 && o < p.O) {
            p.y[(size_t)m * p.ldy + o] = yv;
            d1 += (double)yv;
            d2 += (double)yv * (double)yv;
        }
    }
    if (p.ystats != nullptr) {        // batch statistics of the output for the consumer BatchNorms
        if (sgn == 0) {
            atomicAdd(&s_red[0][ol], d1);
            atomicAdd(&s_red[1][ol], d2);
        }
        __syncthreads();
        if (tid < 32 && blockIdx.y * 32 + tid < p.O) {
            __hip_atomic_fetch_add(p.ystats + blockIdx.y * 32 + tid, s_red[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ystats + p.O + blockIdx.y * 32 + tid, s_red[1][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Third-generation AND-popcount forward (round 5, planner option popcount_pixels = 1, the default): LANE = PIXEL, the weight masks are
// SCALAR operands.  The wave-per-pixel kernel above spends one v_readlane per plane dword to turn a VECTOR value (the plane word of its
// pixel) into the scalar operand of v_and -- 252 readlanes + 252 v_and + 252 v_bcnt per pixel and 32 output channels, both signs.  Here
// a wave owns 64 consecutive pixels (lane l = pixel m0 + l) and TPX_OC = 8 output channels: a lane keeps the 14 plane words of ITS pixel's
// neighbour record in registers (seven 16-byte loads per tap; the records of a tensor stay in L2), the masks of the 8 output channels
// arrive through the scalar cache (the address is wave-uniform) and sit in SGPRs: v_and_b32 v, s, v + v_bcnt_u32_b32 per dword, nothing
// else.  And only ONE mask per weight word is counted: with P / N the +1 / -1 masks and Z = ~(P | N) the zero weights,
//     popc(P & x) - popc(N & x) = 2 popc(P & x) - popc(x) + popc(Z & x),
// where sum_b 2^b popc(x_b) over a record is the pixel sum the plane kernel already stores in word 14, and the Z term is skipped by a
// scalar branch when a word has no zero weight (binary weights, utils/quantize.py:125-149 with bits_w = 1: always).  Per 64 pixels and
// output channel: 9 taps x (28 v_and + 28 v_bcnt + 7 shift-adds) = 567 vector instructions = 8.9 per pixel and channel against 23.6.
// Work item = (group of 64 pixels, chunk of 8 output channels); every integer stays below 2^24: bit-identical to the kernels above and to
// the MFMA forward of the same node.  The output's batch statistics: per item a reduce-scatter over the 64 lanes (8 values -> one per
// lane group) in exact integer / fp64 arithmetic, two LDS atomics on 8 lanes, one pair of global atomics per channel and block.
constexpr int TPX_THREADS = 256;
// popc(v) + acc in ONE instruction (v_bcnt_u32_b32's second operand): left to itself hipcc counts into a fresh register and adds
// afterwards (v_bcnt ..., 0 + v_add3: 72 instead of 63 vector instructions per output channel and tap)
__device__ __forceinline__ int tpx_count(unsigned v, int acc) {
#ifdef CUNET_TPX_NOASM      // (probe builds: the compiler's own selection)
    return __popc(v) + acc;
#else
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
#endif
}

// TPX_OC output channels per work item: 8 where the launch has work items to spare (64 x 64), 4 / 2 below -- a 16 x 16 launch of batch 24 is
// 96 pixel groups: with 8 channels per item 384 items sit on 1024 SIMDs and the launch lasts one item's latency (25 us for 10 us of work).
template <int TAPS, int TPX_OC>
__global__ __launch_bounds__(TPX_THREADS) void ternary_conv_pixels_kernel(const TernArgs p) {
    __shared__ double s_red[2][32];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (p.C + 63) >> 6;
    const int ocol0 = blockIdx.y * 32;                                  // this block column's 32 output channels
    const int ocols = p.O - ocol0 < 32 ? p.O - ocol0 : 32;
    const int nch = (ocols + TPX_OC - 1) / TPX_OC;                      // chunks of 8 output channels
    if (tid < 64) s_red[tid >> 5][tid & 31] = 0.0;
    __syncthreads();
    const float qs = exp2f((float)(p.bits_i - 1));
    const int HW = p.H * p.W;
    const int groups = (p.M + 63) >> 6;
    const int items = groups * nch;
    const int stride = gridDim.x * (TPX_THREADS / 64);
    const uint4* planes = reinterpret_cast<const uint4*>(p.planes);    // a record = 8 uint4 (TERN_REC_WORDS * 8 bytes)
    typedef const __attribute__((address_space(4))) uint64_t* cmask_t;
    const cmask_t cpos = (cmask_t)(uintptr_t)p.wpos, cneg = (cmask_t)(uintptr_t)p.wneg;
    // valid-channel masks per group (channels >= C carry zero planes; their "zero weights" must not trigger the Z term)
    uint64_t valid0 = 0, valid1 = 0;
    for (int b = 0; b < 64; ++b) {                                      // (bit b of a word = channel tern_chan_of_bit(b) of its group)
        if (tern_chan_of_bit(b) < p.C) valid0 |= 1ull << b;
        if (64 + tern_chan_of_bit(b) < p.C) valid1 |= 1ull << b;
    }

    for (int item = blockIdx.x * (TPX_THREADS / 64) + wave; item < items; item += stride) {
        const int grp = item / nch;
        const int oc = item - grp * nch;
        const int o0 = ocol0 + oc * TPX_OC;                             // wave-uniform
        const int m = grp * 64 + lane;
        const bool live = m < p.M;
        const int mc = live ? m : p.M - 1;
        const int ni = mc / HW;
        const int rem = mc - ni * HW;
        const int py = rem / p.W, px = rem - py * p.W;
        int A[TPX_OC], Zt[TPX_OC];
#pragma unroll
        for (int j = 0; j < TPX_OC; ++j) { A[j] = 0; Zt[j] = 0; }
        int S = 0;
        auto fetch = [&](int t, uint4 (&w)[7], unsigned& sum) {               // the record of tap t's neighbour pixel (t is wave-uniform)
            int row = mc;
            if (TAPS == 9) {
                const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
                const int yy = py + dy, xx = px + dx;
                const bool inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;      // zero padding is post-activation
                row = inside ? mc + dy * p.W + dx : p.M;                            // record M is all zero
            }
            const uint4* rec = planes + (size_t)row * (TERN_REC_WORDS / 2);
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = rec[k];                              // plane words 0 .. 13 (7 g + b)
            sum = reinterpret_cast<const unsigned*>(rec)[28];                       // word 14: the record's sum of quantised activations
        };
        uint4 wn[7];
        unsigned sn;
        fetch(0, wn, sn);
#pragma unroll 1
        for (int t = 0; t < TAPS; ++t) {                                          // (not unrolled: nine taps of 28 plane dwords do not fit the registers)
            uint4 w[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = wn[k];
            S += (int)sn;
            if (t + 1 < TAPS) fetch(t + 1, wn, sn);                                 // the next tap's record is in flight during this tap's counting
            // dwords of plane word i: lo = xs[2 i], hi = xs[2 i + 1]
            const unsigned xs[28] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y, w[1].z, w[1].w, w[2].x, w[2].y, w[2].z, w[2].w,
                                     w[3].x, w[3].y, w[3].z, w[3].w, w[4].x, w[4].y, w[4].z, w[4].w, w[5].x, w[5].y, w[5].z, w[5].w,
                                     w[6].x, w[6].y, w[6].z, w[6].w};
            // this tap's masks of the 8 output channels: wave-uniform addresses in the CONSTANT address space (nothing in this kernel writes
            // the mask region) -- scalar loads, the words live in SGPRs and enter v_and as scalar operands
            uint64_t Pm[2][TPX_OC], Zm[2][TPX_OC];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < TPX_OC; ++j) {
                    // (unconditional loads -- a one-group conv re-reads group 0 and masks the result: no branch between the scalar loads)
                    const size_t wi = ((size_t)t * G + (g < G ? g : 0)) * p.Opad + o0 + j;
                    const uint64_t P = cpos[wi], N = cneg[wi];
                    const uint64_t keep = g < G ? ~0ull : 0ull;
                    Pm[g][j] = P & keep;
                    Zm[g][j] = ~(P | N) & (g == 0 ? valid0 : valid1) & keep;
                }
#pragma unroll
            for (int j = 0; j < TPX_OC; ++j) {
                int c[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const unsigned Pl = (unsigned)Pm[g][j], Ph = (unsigned)(Pm[g][j] >> 32);
#pragma unroll
                    for (int b = 0; b < 7; ++b) c[b] = tpx_count(Ph & xs[2 * (7 * g + b) + 1], tpx_count(Pl & xs[2 * (7 * g + b)], c[b]));
                }
                int a = 0;
#pragma unroll
                for (int b = 0; b < 7; ++b) a += c[b] << b;
                A[j] += a;
                if ((Zm[0][j] | Zm[1][j]) != 0) {                                   // (scalar branch: a weight word with zeros -- ternary weights)
                    int z[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const unsigned Zl = (unsigned)Zm[g][j], Zh = (unsigned)(Zm[g][j] >> 32);
#pragma unroll
                        for (int b = 0; b < 7; ++b) z[b] = tpx_count(Zh & xs[2 * (7 * g + b) + 1], tpx_count(Zl & xs[2 * (7 * g + b)], z[b]));
                    }
                    int zz = 0;
#pragma unroll
                    for (int b = 0; b < 7; ++b) zz += z[b] << b;
                    Zt[j] += zz;
                }
            }
        }
        int yi[TPX_OC];
#pragma unroll
        for (int j = 0; j < TPX_OC; ++j) yi[j] = live ? 2 * A[j] - S + Zt[j] : 0;
        // ---- store (8 consecutive channels per pixel)
        if (live) {
            float* yrow = p.y + (size_t)m * p.ldy + o0;
            if (TPX_OC >= 4 && o0 + TPX_OC <= p.O && (p.ldy & 3) == 0) {
#pragma unroll
                for (int j = 0; j + 3 < TPX_OC; j += 4)
                    *reinterpret_cast<float4*>(yrow + j) = make_float4((float)yi[j] / qs, (float)yi[j + 1] / qs, (float)yi[j + 2] / qs, (float)yi[j + 3] / qs);
            } else if (TPX_OC == 2 && o0 + 2 <= p.O && (p.ldy & 1) == 0) {
                *reinterpret_cast<float2*>(yrow) = make_float2((float)yi[0] / qs, (float)yi[1] / qs);
            } else {
#pragma unroll
                for (int j = 0; j < TPX_OC; ++j)
                    if (o0 + j < p.O) yrow[j] = (float)yi[j] / qs;
            }
        }
        // ---- batch statistics of the output: sum y = sum yi / qs, sum y^2 = sum yi^2 / qs^2 (all exact)
        if (p.ystats != nullptr) {
            int v[TPX_OC];
            double q[TPX_OC];
#pragma unroll
            for (int j = 0; j < TPX_OC; ++j) { v[j] = yi[j]; q[j] = (double)yi[j] * (double)yi[j]; }
            constexpr int LG = TPX_OC == 8 ? 3 : (TPX_OC == 4 ? 2 : 1);
            int jo = 0;                                                             // the channel this lane group ends up summing
#pragma unroll
            for (int step = 0; step < LG; ++step) {                                 // reduce-scatter: OC -> OC / 2 -> ... -> 1 values per lane
                const int mask = 32 >> step, half = TPX_OC >> (step + 1);
                const bool up = (lane & mask) != 0;
                jo += up ? half : 0;
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const int sendv = up ? v[j] : v[j + half], keepv = up ? v[j + half] : v[j];
                    const double sendq = up ? q[j] : q[j + half], keepq = up ? q[j + half] : q[j];
                    v[j] = keepv + __shfl_xor(sendv, mask, 64);
                    q[j] = keepq + __shfl_xor(sendq, mask, 64);
                }
            }
#pragma unroll
            for (int mask = 32 >> LG; mask > 0; mask >>= 1) {
                v[0] += __shfl_xor(v[0], mask, 64);
                q[0] += __shfl_xor(q[0], mask, 64);
            }
            if ((lane & ((64 >> LG) - 1)) == 0 && o0 + jo < p.O) {
                atomicAdd(&s_red[0][oc * TPX_OC + jo], (double)v[0] / (double)qs);
                atomicAdd(&s_red[1][oc * TPX_OC + jo], q[0] / ((double)qs * (double)qs));
            }
        }
    }
    if (p.ystats != nullptr) {
        __syncthreads();
        if (tid < 32 && ocol0 + tid < p.O) {
            __hip_atomic_fetch_add(p.ystats + ocol0 + tid, s_red[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.ystats + p.O + ocol0 + tid, s_red[1][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// every conv of a table in one launch: blockIdx.y = table entry (same packing as ternary_pack_kernel)
__global__ __launch_bounds__(256) void ternary_pack_all_kernel(const TernPackEntry* __restrict__ tab, const float* __restrict__ params,
                                                               uint64_t* __restrict__ masks) {
    const TernPackEntry e = tab[blockIdx.y];
    const int G = (e.C + 63) >> 6;
    const long total = (long)e.taps * G * e.Opad;
    const float* w = params + e.src;
    uint64_t* wpos = masks + e.dst;
    uint64_t* wneg = wpos + total;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % e.Opad);
        const int g = (int)((i / e.Opad) % G);
        const int t = (int)(i / ((long)e.Opad * G));
        uint64_t pp = 0, nn = 0;
        if (o < e.O)
            for (int c = 0; c < 64; ++c) {
                const int ch = 64 * g + c;
                if (ch >= e.C) break;
                const float v = w[((size_t)o * e.C + ch) * e.taps + t];
                if (v > 0.f) pp |= (uint64_t)1 << tern_bit_of_chan(c);
                if (v < 0.f) nn |= (uint64_t)1 << tern_bit_of_chan(c);
            }
        wpos[i] = pp;
        wneg[i] = nn;
    }
}

// packs torch-layout ternary weights [O][C][taps] (values -1, 0, +1) into the two bit-mask tensors
__global__ __launch_bounds__(256) void ternary_pack_kernel(const float* __restrict__ w, uint64_t* wpos, uint64_t* wneg,
                                                           int O, int C, int taps, int Opad) {
    const int G = (C + 63) >> 6;
    const long total = (long)taps * G * Opad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % Opad);
        const int g = (int)((i / Opad) % G);
        const int t = (int)(i / ((long)Opad * G));
        uint64_t pp = 0, nn = 0;
        if (o < O)
            for (int c = 0; c < 64; ++c) {
                const int ch = 64 * g + c;
                if (ch >= C) break;
                const float v = w[((size_t)o * C + ch) * taps + t];
                if (v > 0.f) pp |= (uint64_t)1 << tern_bit_of_chan(c);
                if (v < 0.f) nn |= (uint64_t)1 << tern_bit_of_chan(c);
            }
        wpos[i] = pp;
        wneg[i] = nn;
    }
}

hipError_t launch_ternary_pack(const float* w, uint64_t* wpos, uint64_t* wneg, int O, int C, int taps, int Opad, hipStream_t s) {
    const long total = (long)taps * ((C + 63) / 64) * Opad;
    hipLaunchKernelGGL(ternary_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wpos, wneg, O, C, taps, Opad);
    return hipGetLastError();
}

hipError_t launch_ternary_pack_all(const TernPackEntry* tab, int n, const float* params, uint64_t* masks, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ternary_pack_all_kernel, dim3(5, n), dim3(256), 0, s, tab, params, masks);      // 9*2*64 = 1152 words at most
    return hipGetLastError();
}

hipError_t launch_ternary_conv(const TernArgs& a, int num_cus, hipStream_t s) {
    if ((a.taps != 1 && a.taps != 9) || a.bits_i < 2 || a.bits_i > 15) return hipErrorInvalidValue;
    if (a.planes != nullptr && a.C <= 128 && a.bits_i <= 8) {      // plan path (and cunet_ternary_conv_ex): bit-planes once per tensor
        if (a.variant != 0 && a.C % 4 == 0 && a.ldx % 4 == 0 && ((uintptr_t)a.x & 15) == 0) {      // lane = pixel (16-byte pieces of 4 channels)
            int gp = ((a.M + 63) / 64 + 3) / 4;
            if (gp > 8 * num_cus) gp = 8 * num_cus;
            hipLaunchKernelGGL(ternary_planes_rows_kernel, dim3(gp), dim3(256), 0, s, a);
        } else {
            int gp = (a.M + 7) / 8;
            if (gp > 8 * num_cus) gp = 8 * num_cus;
            hipLaunchKernelGGL(ternary_planes_kernel, dim3(gp), dim3(256), 0, s, a);
        }
        if (a.variant != 0) {            // lane = pixel, masks as scalar operands (planner option popcount_pixels)
            const int cols = (a.O + 31) / 32;
            const int ocols = a.O < 32 ? a.O : 32;
            const long groups = (a.M + 63) / 64;
            // output channels per work item: the most that still leaves an item per SIMD (1024 SIMDs; 32 x 32 at 8 channels: 34.7 us, at 4: 40.9 -- the plane loads are amortised over fewer channels)
#ifndef CUNET_TPX_MAX_OC
#define CUNET_TPX_MAX_OC 8
#endif
            int oc = CUNET_TPX_MAX_OC;
            while (oc > 2 && groups * ((ocols + oc - 1) / oc) * cols < 1024) oc >>= 1;
            const int nch = (ocols + oc - 1) / oc;
            const long items = groups * nch;
            long gxp = (items + TPX_THREADS / 64 - 1) / (TPX_THREADS / 64);
            const long cap = 6L * num_cus;                                   // six 4-wave blocks per CU: the kernel's occupancy
            if (gxp > cap) {
                const long rounds = (gxp + cap - 1) / cap;                   // every wave walks the same number of items
                gxp = (gxp + rounds - 1) / rounds;
            }
            if (gxp < 1) gxp = 1;
            const dim3 gridp((unsigned)gxp, cols);
#define CUNET_TPX(T_, OC_) hipLaunchKernelGGL((ternary_conv_pixels_kernel<T_, OC_>), gridp, dim3(TPX_THREADS), 0, s, a)
            if (a.taps == 9) { if (oc == 8) CUNET_TPX(9, 8); else if (oc == 4) CUNET_TPX(9, 4); else CUNET_TPX(9, 2); }
            else { if (oc == 8) CUNET_TPX(1, 8); else if (oc == 4) CUNET_TPX(1, 4); else CUNET_TPX(1, 2); }
#undef CUNET_TPX
            return hipGetLastError();
        }
        int gx = (a.M + TCP_THREADS / 64 - 1) / (TCP_THREADS / 64);
        if (gx > num_cus) gx = num_cus;
        const dim3 grid(gx, (a.O + 31) / 32);
        if (a.taps == 9) hipLaunchKernelGGL((ternary_conv_planes_kernel<9>), grid, dim3(TCP_THREADS), 0, s, a);
        else hipLaunchKernelGGL((ternary_conv_planes_kernel<1>), grid, dim3(TCP_THREADS), 0, s, a);
        return hipGetLastError();
    }
    if ((a.C + 63) / 64 > (a.taps == 1 ? TC_MAXG1 : TC_MAXG9)) return hipErrorInvalidValue;
    int gx = (a.M + 3) / 4;
    if (gx > 8 * num_cus) gx = 8 * num_cus;
    const dim3 grid(gx, a.Opad / 64);
    if (a.taps == 1) hipLaunchKernelGGL((ternary_conv_kernel<1, TC_MAXG1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ternary_conv_kernel<9, TC_MAXG9>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace cunet
