// GroupNorm(32, eps) + optional SiLU on NHWC rows -- the HBM-bound half of the encoder
// (SURVEY.md section 2, K4).  Three launches:
//   stats    : one read of x  -> per (image, pixel-chunk, group) partial (n, mean, M2), pivot-shifted fp32
//              (conv epilogues emit the same triples for their outputs, so this pass is only needed for conv_in)
//   finalize : deterministic fixed-order merge in fp64 -> per (image, channel) (scale, shift)
//   apply    : y = act(x*scale + shift), one read of x, one bf16 write (the MFMA operand of the next conv)
// x is the fp32 residual stream or a bf16 conv output.  16-B vector accesses, 8 channels per lane.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_CHUNK_PIX = 1024;     // pixels per stats block (upper bound)

template <typename T> struct Load8;
template <> struct Load8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ void ld_nt(const float* p, float (&v)[8]) {
        const f32x4 a = __builtin_nontemporal_load((const f32x4*)p), b = __builtin_nontemporal_load((const f32x4*)(p + 4));
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
};
template <> struct Load8<f16_t> {
    static __device__ __forceinline__ void ld(const f16_t* p, float (&v)[8]) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        const f16x8 a = *(const f16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void ld_nt(const f16_t* p, float (&v)[8]) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        const f16x8 a = __builtin_nontemporal_load((const f16x8*)p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
};
template <> struct Load8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
        const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void ld_nt(const bf16_t* p, float (&v)[8]) {
        const bf16x8 a = __builtin_nontemporal_load((const bf16x8*)p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
};

// Numerically robust partials: every lane accumulates sums of (v - pivot), pivot = the first value it
// sees for that slot, so E[d^2] - E[d]^2 never cancels even when |mean| >> std; lanes, chunks and
// images are then merged as (n, mean, M2) triples with Chan's formula in a FIXED order (deterministic).
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
}

// SLOTS = groups covered by one lane's 8 channels = max(1, 8 / channels_per_group)
template <typename T, int SLOTS>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, int HW, int C, int cpg,
                                                              int chunk_pix, int nchunks,
                                                              float* __restrict__ partial) {
    __shared__ float red[GN_THREADS][SLOTS][3];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tpp = C >> 3;                       // lanes per pixel
    const int ppp = GN_THREADS / tpp;             // pixels per pass
    const int tc = threadIdx.x % tpp;             // this lane's channel chunk (fixed for the whole block)
    const int tp = threadIdx.x / tpp;
    const int pbeg = chunk * chunk_pix;
    const int pend = min(HW, pbeg + chunk_pix);
    const T* xb = x + ((long long)b * HW) * C + tc * 8;
    constexpr int PER = 8 / SLOTS;
    float s[SLOTS], ss[SLOTS], piv[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) s[i] = ss[i] = piv[i] = 0.f;
    int cnt = 0;
    if (pbeg + tp < pend) {
        float v[8];
        Load8<T>::ld(xb + (long long)(pbeg + tp) * C, v);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) piv[i] = v[i * PER];
    }
#pragma unroll 4
    for (int p = pbeg + tp; p < pend; p += ppp) {
        float v[8];
        Load8<T>::ld(xb + (long long)p * C, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = v[i] - piv[i / PER];
            s[i / PER] += d;
            ss[i / PER] = fmaf(d, d, ss[i / PER]);
        }
        ++cnt;
    }
    const float nt = (float)(cnt * PER);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const float ms = cnt ? s[i] / nt : 0.f;
        red[threadIdx.x][i][0] = nt;
        red[threadIdx.x][i][1] = piv[i] + ms;
        red[threadIdx.x][i][2] = cnt ? fmaxf(ss[i] - s[i] * ms, 0.f) : 0.f;
    }
    __syncthreads();
    // one lane per group merges its contributors in a fixed order
    const int groups = C / cpg;
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        if (SLOTS > 1) {
            const int ct = (g * cpg) >> 3, sl = ((g * cpg) & 7) / cpg;   // lane-chunk and slot holding group g
            for (int q = 0; q < ppp; ++q) chan_merge(n, mean, m2, red[q * tpp + ct][sl][0], red[q * tpp + ct][sl][1], red[q * tpp + ct][sl][2]);
        } else {
            const int nct = cpg >> 3, ct0 = (g * cpg) >> 3;              // group spans nct lane-chunks
            for (int q = 0; q < ppp; ++q)
                for (int c = 0; c < nct; ++c) chan_merge(n, mean, m2, red[q * tpp + ct0 + c][0][0], red[q * tpp + ct0 + c][0][1], red[q * tpp + ct0 + c][0][2]);
        }
        float* o = partial + (((long long)b * nchunks + chunk) * groups + g) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

// one workgroup of FIN_T threads per (image, group): merges [nparts] (n, mean, M2) triples -- written by gn_stats_kernel or by a
// conv epilogue -- with a fixed thread<-part assignment, in fp64, then xor-shuffle trees and a fixed-order merge of the waves
// => deterministic.  (Sixteen waves: the 4096-partial tensors of the 1024^2 stage took 66-71 us per call on four waves -- two dependent
// sweeps of 16 strided loads per thread, all latency -- and 4 launches of a step sit between convs with nothing to hide them under.)
constexpr int FIN_T = 1024;
__global__ __launch_bounds__(FIN_T) void gn_finalize_kernel(const float* __restrict__ partial, int nparts, int C,
                                                            int groups, float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ scale_shift, int* __restrict__ status) {
    __shared__ double red[FIN_T / 64][2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6;
    const int nt = blockDim.x;                                // 256, or FIN_T for the large partial tensors (the launcher's choice)
    const int cpg = C / groups;
    const float* pb = partial + ((long long)b * nparts * groups + g) * 3;
    auto block_sum2 = [&](double& a0, double& a1) {           // sums over the workgroup, identical in every thread
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
        __syncthreads();                                      // (the previous use of `red` is over)
        if ((tid & 63) == 0) { red[wave][0] = a0; red[wave][1] = a1; }
        __syncthreads();
        a0 = 0.0; a1 = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) { a0 += red[w][0]; a1 += red[w][1]; }
    };
    // Every partial is a 12-byte record 384 B (32 groups) away from the next one of this (image, group): one lane = one cache-line request per
    // load, which is what this kernel's time is (70 us for 4096 partials on 4 or on 16 waves alike, three dword loads and two sweeps) -- so a
    // record is ONE dwordx3 load and, up to FIN_CACHE records per thread, stays in registers for the second sweep.
    struct Rec { float n, mean, m2; };
    constexpr int FIN_CACHE = 4;
    const bool cached = nparts <= FIN_CACHE * nt;
    Rec reg[FIN_CACHE];
    double n = 0.0, s = 0.0;
    if (cached) {
#pragma unroll
        for (int i = 0; i < FIN_CACHE; ++i) {
            const int c = tid + i * nt;
            reg[i] = c < nparts ? *(const Rec*)(pb + (long long)c * groups * 3) : Rec{0.f, 0.f, 0.f};
            n += (double)reg[i].n;
            s += (double)reg[i].n * (double)reg[i].mean;
        }
    } else {
        for (int c = tid; c < nparts; c += nt) {
            const Rec t = *(const Rec*)(pb + (long long)c * groups * 3);
            n += (double)t.n;
            s += (double)t.n * (double)t.mean;
        }
    }
    block_sum2(n, s);
    const double mean = s / n;
    double m2 = 0.0, unused = 0.0;
    if (cached) {
#pragma unroll
        for (int i = 0; i < FIN_CACHE; ++i) {
            const double d = (double)reg[i].mean - mean;
            m2 += (double)reg[i].m2 + (double)reg[i].n * d * d;        // (an empty slot adds 0 + 0 * d * d)
        }
    } else {
        for (int c = tid; c < nparts; c += nt) {
            const Rec t = *(const Rec*)(pb + (long long)c * groups * 3);
            const double d = (double)t.mean - mean;
            m2 += (double)t.m2 + (double)t.n * d * d;
        }
    }
    block_sum2(m2, unused);
    const double var = m2 / n;
    // every tensor of the path passes through here: inf / NaN statistics (an fp16-stored activation beyond +-65504, or a
    // checkpoint with NaN weights) raise the sticky status bit the host reads with vt_status -- the outputs are garbage then
    if (status && tid == 0 && !(fabs(mean) <= 1.0e300 && fabs(var) <= 1.0e300)) atomicOr(status, 1);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int c = tid; c < cpg; c += nt) {
        const int ch = g * cpg + c;
        const float sc = rstd * gamma[ch];
        float* o = scale_shift + ((long long)b * C + ch) * 2;
        o[0] = sc;
        o[1] = beta[ch] - fmean * sc;
    }
}

// OUT: 0 = bf16 rows; 1 = e4m3(out_scale * y), saturated at +-448 (1 B per element: the operand of the fp8 conv); 2 = fp16 rows (the
// fp16-operand mode of the convs, vt_set_flag 18: 11 significand bits instead of bf16's 8 at the same 2 B).
// A lane handles 8 consecutive channels of a pixel and stores 8 B of e4m3 (measured on MI355X at 16 x 1024^2: 4.7 TB/s of read +
// write; 16 channels per lane with 16-B stores 3.8 TB/s; lane pairs exchanging through DPP so that half the lanes, or -- with two
// pixels per lane -- all lanes store 16 B: 4.4 TB/s and 1 % slower end to end; plain instead of nontemporal stores -0.5 %).
template <typename T, bool SILU, int OUT>
__device__ __forceinline__ void gn_apply_body(const T* __restrict__ x, const float* __restrict__ scale_shift, void* __restrict__ yv, int HW, int C,
                                              int pix_per_block, float out_scale, int* __restrict__ status) {
    constexpr int CPL = 8;
    constexpr bool OUT8 = OUT == 1;
    const int b = blockIdx.y;
    const int tpp = C / CPL, ppp = GN_THREADS / tpp;
    const int tc = threadIdx.x % tpp, tp = threadIdx.x / tpp;
    float sc[CPL], sh[CPL];
    const float* ssb = scale_shift + ((long long)b * C + tc * CPL) * 2;
#pragma unroll
    for (int i = 0; i < CPL / 2; ++i) {
        const f32x4 q = *(const f32x4*)(ssb + i * 4);
        sc[2 * i] = q[0]; sh[2 * i] = q[1]; sc[2 * i + 1] = q[2]; sh[2 * i + 1] = q[3];
    }
    const int pbeg = blockIdx.x * pix_per_block;
    const int pend = min(HW, pbeg + pix_per_block);
    const long long base = ((long long)b * HW) * C + tc * CPL;
    float amax = 0.f;                                   // OUT8: largest |scaled activation| this thread converted
#pragma unroll 4
    for (int p = pbeg + tp; p < pend; p += ppp) {
        float t[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k += 8) {
            float v[8];
            Load8<T>::ld_nt(x + base + (long long)p * C + k, v);      // streamed once: keep it out of the caches (+2 % measured)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[k + i] = fmaf(v[i], sc[k + i], sh[k + i]);
                if (SILU) t[k + i] = vt_silu(t[k + i]);
            }
        }
        if constexpr (OUT8) {
            int o0 = 0, o1 = 0;
#pragma unroll
            for (int i = 0; i < CPL; ++i) t[i] *= out_scale;
#pragma unroll
            for (int i = 0; i < CPL; i += 2) amax = fmaxf(fmaxf(fabsf(t[i]), fabsf(t[i + 1])), amax);       // v_max3_f32 with |.| modifiers
#pragma unroll
            for (int i = 0; i < CPL; ++i) t[i] = __builtin_amdgcn_fmed3f(t[i], -448.f, 448.f);
            o0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], o0, false);
            o0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], o0, true);
            o1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], o1, false);
            o1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], o1, true);
            typedef int i32x2 __attribute__((ext_vector_type(2)));
#ifdef GN8_PLAIN_STORE
            *(i32x2*)((unsigned char*)yv + base + (long long)p * C) = i32x2{o0, o1};
#else
            __builtin_nontemporal_store(i32x2{o0, o1}, (i32x2*)((unsigned char*)yv + base + (long long)p * C));
#endif
        } else if constexpr (OUT == 2) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16_t)t[i];
            __builtin_nontemporal_store(o, (f16x8*)((f16_t*)yv + base + (long long)p * C));
        } else {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16_t)t[i];
            __builtin_nontemporal_store(o, (bf16x8*)((bf16_t*)yv + base + (long long)p * C));
        }
    }
    // e4m3 saturates silently: tell the host once if anything was clamped (sticky bit, vt_status); NaN compares false and is the
    // finalize kernel's business
    if (OUT8 && status && amax > 448.f) atomicOr(status, 2);
}

template <typename T, bool SILU, int OUT>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, void* __restrict__ yv, int HW,
                                                              int C, int pix_per_block, float out_scale, int* __restrict__ status) {
    gn_apply_body<T, SILU, OUT>(x, scale_shift, yv, HW, C, pix_per_block, out_scale, status);
}
// fp32 input (the fp32 residual-stream storage of vt_set_flag(ctx, 4, 0)): the compiler gathers the (scale, shift) pairs of this instantiation with
// v_pk_mov_b32 ... op_sel:[1,0] -- a high register routed into the low lane, the operand form of the hazard in DESIGN.md 4.14 -- so this cold variant is
// built without packed fp32 (tests/test_isa_lint.py forbids the form in every kernel).
template <typename T, bool SILU, int OUT>
__global__ __launch_bounds__(GN_THREADS) VT_NO_PACKED_F32 void gn_apply_f32in_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift,
                                                                                     void* __restrict__ yv, int HW, int C, int pix_per_block, float out_scale,
                                                                                     int* __restrict__ status) {
    gn_apply_body<T, SILU, OUT>(x, scale_shift, yv, HW, C, pix_per_block, out_scale, status);
}

template <typename T, bool SILU, int OUT>
void launch_gn_apply(dim3 grid, dim3 block, hipStream_t s, const T* x, const float* scale_shift, void* y, int HW, int C, int ppb, float out_scale, int* status) {
    if constexpr (std::is_same<T, float>::value)
        hipLaunchKernelGGL((gn_apply_f32in_kernel<T, SILU, OUT>), grid, block, 0, s, x, scale_shift, y, HW, C, ppb, out_scale, status);
    else
        hipLaunchKernelGGL((gn_apply_kernel<T, SILU, OUT>), grid, block, 0, s, x, scale_shift, y, HW, C, ppb, out_scale, status);
}

bool gn_shape_ok(int C, int groups) {
    if (C <= 0 || groups <= 0 || C % groups) return false;
    const int cpg = C / groups;
    if (C % 8 || (GN_THREADS % (C / 8)) != 0 || C / 8 > GN_THREADS) return false;
    if (cpg & (cpg - 1)) return false;         // power of two
    if (cpg < 2) return false;
    return true;
}

int chunk_pix_for(int HW, int C) {
    const int ppp = GN_THREADS / (C / 8);
    int cp = GN_CHUNK_PIX;
    if (cp < ppp) cp = ppp;
    return cp;
}

}  // namespace

int vt_gn_max_chunks(int HW, int C) {
    const int cp = chunk_pix_for(HW, C);
    return (HW + cp - 1) / cp;
}

hipError_t vt_launch_gn_stats(const void* x, int x_dtype, int B, int HW, int C, int groups, float* partial,
                              int* nchunks_out, hipStream_t s) {
    if (!gn_shape_ok(C, groups) || B <= 0 || HW <= 0) return hipErrorInvalidValue;
    const int cpg = C / groups;
    const int cp = chunk_pix_for(HW, C);
    const int nchunks = (HW + cp - 1) / cp;
    if (nchunks_out) *nchunks_out = nchunks;
    dim3 grid(nchunks, B), block(GN_THREADS);
    const int slots = cpg >= 8 ? 1 : 8 / cpg;
#define GN_STATS(T, S) hipLaunchKernelGGL((gn_stats_kernel<T, S>), grid, block, 0, s, (const T*)x, HW, C, cpg, cp, nchunks, partial)
    if (x_dtype == 1) {
        if (slots == 1) GN_STATS(float, 1); else if (slots == 2) GN_STATS(float, 2); else GN_STATS(float, 4);
    } else if (x_dtype == 2) {
        if (slots == 1) GN_STATS(f16_t, 1); else if (slots == 2) GN_STATS(f16_t, 2); else GN_STATS(f16_t, 4);
    } else {
        if (slots == 1) GN_STATS(bf16_t, 1); else if (slots == 2) GN_STATS(bf16_t, 2); else GN_STATS(bf16_t, 4);
    }
#undef GN_STATS
    return hipGetLastError();
}

hipError_t vt_launch_gn_finalize(const float* partial, int nparts, int B, int C, int groups, float eps,
                                 const float* gamma, const float* beta, float* scale_shift, hipStream_t s, int* status) {
    if (C <= 0 || groups <= 0 || C % groups || nparts <= 0 || B <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(nparts >= 2048 ? FIN_T : 256), 0, s, partial, nparts, C, groups, eps, gamma,
                       beta, scale_shift, status);
    return hipGetLastError();
}

hipError_t vt_launch_gn_apply(const void* x, int x_dtype, const float* scale_shift, void* y, int B, int HW,
                              int C, int silu, hipStream_t s, float out_fp8_scale, int* status, int out_f16) {
    if (C % 8 || (GN_THREADS % (C / 8)) != 0 || C / 8 > GN_THREADS || B <= 0 || HW <= 0) return hipErrorInvalidValue;
    const bool o8 = out_fp8_scale > 0.f;
    if (o8 && out_f16) return hipErrorInvalidValue;
    const int ppp = GN_THREADS / (C / 8);
#ifndef GN_PASSES
#define GN_PASSES 4
#endif
    const int ppb = ppp * (out_fp8_scale > 0.f ? 2 * GN_PASSES : GN_PASSES);   // pixels per block: short blocks stream faster (bf16: 5.3 -> 5.9 TB/s at 4 passes; fp8 output: 8 passes +0.6 % images/s)
    dim3 grid((HW + ppb - 1) / ppb, B), block(GN_THREADS);
#define GN_APPLY(T, A, O) launch_gn_apply<T, A, O>(grid, block, s, (const T*)x, scale_shift, y, HW, C, ppb, out_fp8_scale, status)
#define GN_APPLY2(T) do { if (silu) { if (o8) GN_APPLY(T, true, 1); else if (out_f16) GN_APPLY(T, true, 2); else GN_APPLY(T, true, 0); } \
                          else { if (o8) GN_APPLY(T, false, 1); else if (out_f16) GN_APPLY(T, false, 2); else GN_APPLY(T, false, 0); } } while (0)
    if (x_dtype == 1) GN_APPLY2(float);
    else if (x_dtype == 2) GN_APPLY2(f16_t);
    else GN_APPLY2(bf16_t);
#undef GN_APPLY2
#undef GN_APPLY
    return hipGetLastError();
}
