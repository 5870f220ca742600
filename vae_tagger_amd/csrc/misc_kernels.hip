// conv_in (3 -> C0 channels, fp32 direct) and the row softmax of the unfused attention path.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// conv_in: x fp32 NCHW [B,3,H,W] (the tensor the reference hands to vae.encode, infer_full.py:98)
// -> NHWC rows [B][H*W][Cout].  K = 27 is too thin for MFMA and the op is bound by its 128-channel
// output write, so it is a direct fp32 VALU conv: exact fp32 inputs/weights (no bf16 rounding of the
// image).  One workgroup = 64 x 16 pixels x Cout: weights (27 x Cout) and the 18-row input halo are
// staged in LDS once and reused for all 16 rows.  The epilogue also emits GroupNorm (n, mean, M2)
// partials of the output for the first resnet's norm1.
// packed weight layout: wp[k][cout], k = ci*9 + ky*3 + kx.
constexpr int CI_PIX = 64, CI_ROWS = 16;
__global__ __launch_bounds__(256) VT_NO_PACKED_F32 void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                      const float* __restrict__ bias, float* __restrict__ o32,
                                                      bf16_t* __restrict__ o16, f16_t* __restrict__ oh,
                                                      float* __restrict__ gn_partial,
                                                      int gn_cpg, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int RW = CI_PIX + 2, RH = CI_ROWS + 2;
    float* sw = sm;                       // [27][Cout]
    float* sin = sm + 27 * Cout;          // [3 ch][RH rows][RW]
    float* red = sin + 3 * RH * RW;       // [Cout/4][16][2] stats scratch
    const int tid = threadIdx.x;
    const int b = blockIdx.z, y0 = blockIdx.y * CI_ROWS, x0 = blockIdx.x * CI_PIX;
    for (int i = tid; i < 27 * Cout; i += 256) sw[i] = wp[i];
    for (int i = tid; i < 3 * RH * RW; i += 256) {
        const int c = i / (RH * RW), r = (i / RW) % RH, xx = i % RW;
        const int iy = y0 - 1 + r, ix = x0 - 1 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long long)b * 3 + c) * H + iy) * W + ix];
        sin[i] = v;
    }
    __syncthreads();
    const int cgi = tid & 15, ps = tid >> 4;          // 16 cout-groups of 8, 16 pixel slots of 4
    for (int cg = cgi; cg * 8 < Cout; cg += 16) {
        const f32x4 b0 = *(const f32x4*)(bias + cg * 8), b1 = *(const f32x4*)(bias + cg * 8 + 4);
        // shifted sums per 4-cout half, pivot = the half's first bias (close to the output mean: E[x] ~ 0)
        float s0 = 0.f, ss0 = 0.f, s1 = 0.f, ss1 = 0.f;
        int cnt = 0;
        for (int row = 0; row < CI_ROWS; ++row) {
            const int y = y0 + row;
            if (y >= H) break;
            float acc[4][8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acc[p][0] = b0[0]; acc[p][1] = b0[1]; acc[p][2] = b0[2]; acc[p][3] = b0[3];
                acc[p][4] = b1[0]; acc[p][5] = b1[1]; acc[p][6] = b1[2]; acc[p][7] = b1[3];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float in6[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) in6[i] = sin[(c * RH + row + ky) * RW + ps * 4 + i];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* wr = sw + (c * 9 + ky * 3 + kx) * Cout + cg * 8;
                        const f32x4 w0 = *(const f32x4*)wr, w1 = *(const f32x4*)(wr + 4);
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float v = in6[p + kx];
                            acc[p][0] = fmaf(v, w0[0], acc[p][0]); acc[p][1] = fmaf(v, w0[1], acc[p][1]);
                            acc[p][2] = fmaf(v, w0[2], acc[p][2]); acc[p][3] = fmaf(v, w0[3], acc[p][3]);
                            acc[p][4] = fmaf(v, w1[0], acc[p][4]); acc[p][5] = fmaf(v, w1[1], acc[p][5]);
                            acc[p][6] = fmaf(v, w1[2], acc[p][6]); acc[p][7] = fmaf(v, w1[3], acc[p][7]);
                        }
                    }
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int px = x0 + ps * 4 + p;
                if (px >= W) continue;
                const long long o = (((long long)b * H + y) * W + px) * Cout + cg * 8;
                if (o32) {
                    *(f32x4*)(o32 + o) = f32x4{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                    *(f32x4*)(o32 + o + 4) = f32x4{acc[p][4], acc[p][5], acc[p][6], acc[p][7]};
                }
                if (o16) {
                    bf16x8 h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (bf16_t)acc[p][i];
                    *(bf16x8*)(o16 + o) = h;
                }
                if (oh) {
                    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                    f16x8 h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (f16_t)acc[p][i];
                    *(f16x8*)(oh + o) = h;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d0 = acc[p][i] - b0[0], d1 = acc[p][4 + i] - b1[0];
                    s0 += d0; ss0 = fmaf(d0, d0, ss0); s1 += d1; ss1 = fmaf(d1, d1, ss1);
                }
                ++cnt;
            }
        }
        if (gn_partial) {
            // per-thread (n, mean, M2) for each 4-cout half, then fixed-order merges in LDS
            const float n = 4.0f * (float)cnt;
            float* r0 = red + ((cg * 2 + 0) * 16 + ps) * 3;
            float* r1 = red + ((cg * 2 + 1) * 16 + ps) * 3;
            const float m0 = cnt ? s0 / n : 0.f, m1 = cnt ? s1 / n : 0.f;
            r0[0] = n; r0[1] = b0[0] + m0; r0[2] = cnt ? fmaxf(ss0 - s0 * m0, 0.f) : 0.f;
            r1[0] = n; r1[1] = b1[0] + m1; r1[2] = cnt ? fmaxf(ss1 - s1 * m1, 0.f) : 0.f;
        }
    }
    if (gn_partial) {
        __syncthreads();
        const int G = Cout / gn_cpg, hpg = gn_cpg >> 2;        // 4-cout halves per group
        if (tid < G) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
            for (int hh = 0; hh < hpg; ++hh)
                for (int q = 0; q < 16; ++q) {
                    const float* r = red + ((tid * hpg + hh) * 16 + q) * 3;
                    vt_chan_merge(n, mean, m2, r[0], r[1], r[2]);
                }
            const int part = blockIdx.y * gridDim.x + blockIdx.x, nparts = gridDim.x * gridDim.y;
            float* o = gn_partial + (((long long)b * nparts + part) * G + tid) * 3;
            o[0] = n; o[1] = mean; o[2] = m2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv_in on the matrix cores (Cout == 128): the 27 taps x channels of a pixel are one K = 32 operand row, so the conv
// is one v_mfma_f32_16x16x32_bf16 K-step per 16x16 output tile -- three of them, because both operands are split into
// bf16 hi + lo parts (x.w ~ xh.wh + xl.wh + xh.wl, relative error ~2^-16 instead of the 2^-9 of single bf16 operands:
// the first layer's rounding was 5 % of the whole encoder's latent error).  K slots 27..29 carry the bias the same way
// (operand 1.0 x three bf16 pieces of the bias), so accumulators start at zero.  The kernel is bound by its output write.
// Workgroup = 4 waves, 8 rows x 64 pixels x 128 couts; LDS: fp32 halo 3 x 10 x 66 (reused for the statistics) + 4 KB of
// store staging per wave + the 8 KB of lo weights = 32 KB.  A lane builds its operand fragment (8 k-values of one pixel) straight from the halo:
// 8 ds_read_b32 at per-lane tap offsets.  The 16 KB of weights go from L2 to registers.  Weight rows use the interleaved
// cout map: lane (fq, fr) holds couts 64*h + 16*fq + 4*i + r of pixel fr (tile i of half h), i.e. 16 consecutive couts
// per half -> 32-B fp16 / 64-B fp32 pieces.
// wpk: [2 (hi, lo)][128 rows][32 k] bf16 in that row order, k = ci*9 + ky*3 + kx; hi rows carry the bias in k = 27..29.
constexpr int CM_ROWS = 8, CM_PIX = 64;
#ifndef CONV_IN_OCC
#define CONV_IN_OCC 3
#endif
#ifndef CONV_IN_UNROLL
#define CONV_IN_UNROLL 1
#endif
__global__ __launch_bounds__(256, CONV_IN_OCC) void conv_in_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wpk,
                                                           const float* __restrict__ bias, float* __restrict__ o32,
                                                           bf16_t* __restrict__ o16, f16_t* __restrict__ oh,
                                                           float* __restrict__ gn_partial, int H, int W) {
    constexpr int RW = CM_PIX + 2, RH = CM_ROWS + 2, C = 128;
    __shared__ __attribute__((aligned(16))) float sin[3 * RH * RW];
    __shared__ __attribute__((aligned(16))) char stgbuf[4 * 4096];
    __shared__ __attribute__((aligned(16))) char wlds[128 * 64];   // the lo weight rows (64 B each, 16-B chunk ^ ((row >> 2) & 3))
    float (*red)[32][3] = (float (*)[32][3])sin;       // statistics scratch: reuses the halo once every wave is done with it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, y0 = blockIdx.y * CM_ROWS, x0 = blockIdx.x * CM_PIX;
    for (int i = tid; i < 3 * RH * RW; i += 256) {
        const int c = i / (RH * RW), r = (i / RW) % RH, xx = i % RW;
        const int iy = y0 - 1 + r, ix = x0 - 1 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long long)b * 3 + c) * H + iy) * W + ix];
        sin[i] = v;
    }
    for (int q = tid; q < 512; q += 256) {
        const int row = q >> 2, ch = q & 3;
        *(bf16x8*)(wlds + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) = *(const bf16x8*)(wpk + C * 32 + row * 32 + ch * 8);
    }
    const int fr = lane & 15, fq = lane >> 4;
    // byte offsets of this lane's 8 k-values (k = 8*fq + r) relative to the pixel's window origin; k >= 27 reads tap 0
    int offb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = fq * 8 + r, kk = k < 27 ? k : 0;
        const int ci = kk / 9, ky = (kk - ci * 9) / 3, kx = kk - ci * 9 - ky * 3;
        offb[r] = ((ci * RH + ky) * RW + kx) * 4;
    }
    bf16x8 wh[8];                                       // hi weights (8 KB, L2-resident) live in registers, lo ones in LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) wh[i] = *(const bf16x8*)(wpk + (i * 16 + fr) * 32 + fq * 8);
    const char* const wlp = wlds + fr * 64 + ((fq ^ ((fr >> 2) & 3)) << 4);
    float piv[8];                                       // statistics pivot: the first bias of the lane's group
#pragma unroll
    for (int i = 0; i < 8; ++i) piv[i] = bias[64 * (i >> 2) + 16 * fq + 4 * (i & 3)];
    float gs[8], gss[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gs[i] = gss[i] = 0.f;
    int cnt = 0;
    __syncthreads();
    // wave w: tile rows 2w, 2w+1 (128 pixels = 8 pixel tiles).  In the accumulator layout neighbouring lanes are
    // neighbouring PIXELS (256 B apart in the 16-bit output), so 16-bit outputs are transposed through the wave's 4 KB of
    // LDS and leave as 1 KB contiguous per store (4 pixels x 128 channels).  Staged rows are 256 B (128 couts x 2 B) with
    // the 16-B chunk index XOR-ed with the pixel (accumulator-layout writes touch 16 rows per access).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    char* const stg = stgbuf + wave * 4096;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll CONV_IN_UNROLL
    for (int t = 0; t < 8; ++t) {
        const int p = wave * 128 + t * 16 + fr;                       // this lane's pixel in the tile
        const char* pb = (const char*)sin + ((p >> 6) * RW + (p & 63)) * 4;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = *(const float*)(pb + offb[r]);
        if (fq == 3) { v[3] = v[4] = v[5] = 1.f; v[6] = v[7] = 0.f; }  // k = 27..29: the bias slots; 30, 31 unused
        bf16x8 xh, xl;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            xh[r] = (bf16_t)v[r];
            xl[r] = (bf16_t)(v[r] - (float)xh[r]);
        }
        const int y = y0 + (p >> 6), xx = x0 + (p & 63);
        const bool ok = y < H && xx < W;
        const long long o = (((long long)b * H + y) * W + xx) * C + 16 * fq;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, zero4, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(wlp + i * 1024), xh, acc[i], 0, 0, 0);
        if (ok && o32) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(o32 + o + 64 * hh + 4 * i) = acc[hh * 4 + i];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            char* outp = pass == 0 ? (char*)oh : (char*)o16;
            if (!outp) continue;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    char* d = stg + fr * 256 + (((hh * 8 + fq * 2 + (i >> 1)) ^ fr) << 4);
                    if (pass == 0) {
                        f16x8 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[hh * 4 + i][r]; h[4 + r] = (f16_t)acc[hh * 4 + i + 1][r]; }
                        *(f16x8*)d = h;
                    } else {
                        bf16x8 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { h[r] = (bf16_t)acc[hh * 4 + i][r]; h[4 + r] = (bf16_t)acc[hh * 4 + i + 1][r]; }
                        *(bf16x8*)d = h;
                    }
                }
            asm volatile("" ::: "memory");                              // (LDS executes one wave's accesses in order)
            const int p0 = wave * 128 + t * 16;                        // the tile's first pixel: 16 consecutive x of one row
            const int ty = y0 + (p0 >> 6), tx = x0 + (p0 & 63);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = q * 4 + (lane >> 4), ch = lane & 15;
                const u32x4 v4 = *(const u32x4*)(stg + px * 256 + ((ch ^ px) << 4));
                if (ty < H && tx + px < W) {
#ifdef CONV_IN_NT
                    __builtin_nontemporal_store(v4, (u32x4*)(outp + ((((long long)b * H + ty) * W + tx + px) * C) * 2 + ch * 16));
#else
                    *(u32x4*)(outp + ((((long long)b * H + ty) * W + tx + px) * C) * 2 + ch * 16) = v4;
#endif
                }
            }
            asm volatile("" ::: "memory");
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][r] - piv[i]; gs[i] += d; gss[i] = fmaf(d, d, gss[i]); }
            ++cnt;
        }
    }
    if (gn_partial) {
        // 32 GroupNorm groups of 4 couts: group of (half h, fq, tile i) = 16*h + 4*fq + i.  Sums are relative to the
        // group's first bias (E[x] ~ 0 makes that a good pivot); merge the 16 pixel columns, then the 4 waves.
        __syncthreads();                                              // every wave has read its last halo values
        const float n = vt_row16_sum(4.0f * (float)cnt);              // 16-lane DPP row sums (vt_common.h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float s1 = vt_row16_sum(gs[i]), s2 = vt_row16_sum(gss[i]);
            if (fr == 0) {
                const int g = 16 * (i >> 2) + 4 * fq + (i & 3);
                const float ms = n > 0.f ? s1 / n : 0.f;
                red[wave][g][0] = n; red[wave][g][1] = piv[i] + ms; red[wave][g][2] = n > 0.f ? fmaxf(s2 - s1 * ms, 0.f) : 0.f;
            }
        }
        __syncthreads();
        if (tid < 32) {
            float nn = 0.f, mean = 0.f, m2 = 0.f;
            for (int w = 0; w < 4; ++w) vt_chan_merge(nn, mean, m2, red[w][tid][0], red[w][tid][1], red[w][tid][2]);
            const int part = blockIdx.y * gridDim.x + blockIdx.x, nparts = gridDim.x * gridDim.y;
            float* o = gn_partial + (((long long)b * nparts + part) * 32 + tid) * 3;
            o[0] = nn; o[1] = mean; o[2] = m2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax of the attention scores (fp32, or fp16 as the QK^T epilogue writes them) -> bf16 P, the operand of
// the P.V GEMM.  One workgroup per row.  Pad columns [n, ldp) are written as zeros so the GEMM's 8-element
// k-chunks never see garbage.
template <typename T> __device__ __forceinline__ float sm_ld(const T* p, long long i) { return (float)p[i]; }

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ s, bf16_t* __restrict__ p, int n,
                                                           int lds, int ldp) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const T* sr = s + row * lds;
    bf16_t* pr = p + row * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, sm_ld(sr, i));
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += __expf(sm_ld(sr, i) - m);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    for (int i = tid; i < ldp; i += 256) pr[i] = (bf16_t)(i < n ? __expf(sm_ld(sr, i) - m) * inv : 0.f);
}

// Single-read variant: the whole row (ldp <= NV*1024 values) lives in registers -- one HBM read, one bf16 write.
// Requires lds % 4 == 0, ldp % 4 == 0 and aligned bases.
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_rows_cached_kernel(const T* __restrict__ s, bf16_t* __restrict__ p,
                                                                  int n, int lds, int ldp) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const T* sr = s + row * lds;
    bf16_t* pr = p + row * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    f32x4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (k * 256 + tid) * 4;
        if (i + 3 < n) {
            if constexpr (sizeof(T) == 4) v[k] = *(const f32x4*)(sr + i);
            else { const f16x4 h = *(const f16x4*)(sr + i); v[k] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[k][r] = (i + r < n) ? sm_ld(sr, i + r) : -INFINITY;
        }
        m = fmaxf(m, fmaxf(fmaxf(v[k][0], v[k][1]), fmaxf(v[k][2], v[k][3])));
    }
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[k][r] = __expf(v[k][r] - m); sum += v[k][r]; }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (k * 256 + tid) * 4;
        if (i < ldp) {                      // ldp % 4 == 0: the 4-wide store stays inside the row; pad columns get exp(-inf) = 0
            bf16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (bf16_t)(v[k][r] * inv);
            *(bf16x4*)(pr + i) = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ToTensor + Normalize(0.5, 0.5) on the device: uint8 HWC RGB -> fp32 NCHW in [-1, 1], the tensor the
// reference builds on the CPU (modules.py:136-140).  Same fp32 operations as torch: x/255, then (x-0.5)/0.5.
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                            long long HW) {
    const int b = blockIdx.y;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const unsigned char* src = in + ((long long)b * HW + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = __fdiv_rn((float)src[c], 255.0f);
        out[((long long)b * 3 + c) * HW + p] = (v - 0.5f) / 0.5f;
    }
}

// ---------------------------------------------------------------------------------------------
// Pillow's 8-bit two-pass resample (libImaging/Resample.c) on the device: the resize the reference runs on the CPU before
// the encoder (modules.py:126-178).  The host builds Pillow's per-axis tables (first sample, count, 22-bit fixed-point
// coefficients); these kernels apply them exactly as ImagingResampleHorizontal_8bpc / Vertical_8bpc do: int32
// accumulation from a rounding half, arithmetic shift, clip to uint8, and the horizontal pass's uint8 feeds the vertical.
// tab: [n_out][2 + ksize] int32 = (first, count, coefficients...).  3 channels, HWC.
constexpr int RS_PRECISION_BITS = 32 - 8 - 2;
__device__ __forceinline__ unsigned char rs_clip8(int v) {
    v >>= RS_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// src rows [top, top+rows) x cols [left, ...) of a [src_h][src_w][3] image -> out [rows][out_w][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, int src_w, int left, int top,
                                                       const int* __restrict__ tab, int ksize, unsigned char* __restrict__ out,
                                                       int rows, int out_w) {
    const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xx >= out_w || y >= rows) return;
    const int* t = tab + (long long)xx * (2 + ksize);
    const int x0 = t[0], n = t[1];
    const unsigned char* p = src + ((long long)(top + y) * src_w + left + x0) * 3;
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
        const int c = t[2 + k];
        a0 += p[3 * k] * c; a1 += p[3 * k + 1] * c; a2 += p[3 * k + 2] * c;
    }
    unsigned char* o = out + ((long long)y * out_w + xx) * 3;
    o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
}
// in [in_rows][w][3] -> out [out_h][w][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ in, const int* __restrict__ tab, int ksize,
                                                       unsigned char* __restrict__ out, int w, int out_h) {
    const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    if (xx >= w || yy >= out_h) return;
    const int* t = tab + (long long)yy * (2 + ksize);
    const int y0 = t[0], n = t[1];
    const unsigned char* p = in + ((long long)y0 * w + xx) * 3;
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
        const int c = t[2 + k];
        const unsigned char* q = p + (long long)k * w * 3;
        a0 += q[0] * c; a1 += q[1] * c; a2 += q[2] * c;
    }
    unsigned char* o = out + ((long long)yy * w + xx) * 3;
    o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
}
// plain crop copy (a pass whose size does not change is skipped, as in Pillow)
__global__ __launch_bounds__(256) void crop_copy_kernel(const unsigned char* __restrict__ src, int src_w, int left, int top,
                                                        unsigned char* __restrict__ out, int rows, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (i >= w * 3 || y >= rows) return;
    out[(long long)y * w * 3 + i] = src[((long long)(top + y) * src_w + left) * 3 + i];
}

}  // namespace

hipError_t vt_launch_preprocess_u8(const unsigned char* in_hwc, float* out_nchw, int B, int H, int W, hipStream_t s) {
    if (!in_hwc || !out_nchw || B <= 0 || H <= 0 || W <= 0) return hipErrorInvalidValue;
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, s, in_hwc, out_nchw, HW);
    return hipGetLastError();
}

hipError_t vt_launch_resize_u8(const unsigned char* src, int src_h, int src_w, int left, int top, int crop_w, int crop_h,
                               unsigned char* dst, int dst_h, int dst_w, const int* tab_h, int ksize_h, const int* tab_v,
                               int ksize_v, unsigned char* tmp, hipStream_t s) {
    if (!src || !dst || left < 0 || top < 0 || crop_w <= 0 || crop_h <= 0 || left + crop_w > src_w || top + crop_h > src_h ||
        dst_h <= 0 || dst_w <= 0 || dst_h > 65535 || crop_h > 65535)
        return hipErrorInvalidValue;
    const bool need_h = dst_w != crop_w, need_v = dst_h != crop_h;
    if ((need_h && !tab_h) || (need_v && !tab_v) || (need_h && need_v && !tmp)) return hipErrorInvalidValue;
    if (!need_h && !need_v) {
        hipLaunchKernelGGL(crop_copy_kernel, dim3((crop_w * 3 + 255) / 256, crop_h), dim3(256), 0, s, src, src_w, left, top, dst, crop_h, crop_w);
        return hipGetLastError();
    }
    if (need_h) {
        unsigned char* o = need_v ? tmp : dst;
        hipLaunchKernelGGL(resize_h_kernel, dim3((dst_w + 255) / 256, crop_h), dim3(256), 0, s, src, src_w, left, top, tab_h, ksize_h, o, crop_h, dst_w);
        if (need_v) hipLaunchKernelGGL(resize_v_kernel, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, s, tmp, tab_v, ksize_v, dst, dst_w, dst_h);
    } else {
        // vertical only: the source rows are the crop itself; copy it out first so the pass reads a dense [crop_h][w][3] image
        if (!tmp) return hipErrorInvalidValue;
        hipLaunchKernelGGL(crop_copy_kernel, dim3((crop_w * 3 + 255) / 256, crop_h), dim3(256), 0, s, src, src_w, left, top, tmp, crop_h, crop_w);
        hipLaunchKernelGGL(resize_v_kernel, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, s, tmp, tab_v, ksize_v, dst, dst_w, dst_h);
    }
    return hipGetLastError();
}

hipError_t vt_launch_conv_in(const float* x, const float* wp, const float* bias, float* o32, bf16_t* o16, f16_t* oh,
                             float* gn_partial, int gn_cpg, int* gn_parts, int B, int H, int W, int Cout, hipStream_t s) {
    if (!x || !wp || !bias || (!o32 && !o16 && !oh) || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 8))
        return hipErrorInvalidValue;
    if (gn_partial && ((gn_cpg % 4) || gn_cpg <= 0 || (Cout % gn_cpg) || Cout / gn_cpg > 256)) return hipErrorInvalidValue;
    const size_t smem = (size_t)(27 * Cout + 3 * (CI_ROWS + 2) * (CI_PIX + 2) + (Cout / 4) * 16 * 3) * sizeof(float);
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid((W + CI_PIX - 1) / CI_PIX, (H + CI_ROWS - 1) / CI_ROWS, B);
    if (gn_parts) *gn_parts = grid.x * grid.y;
    hipLaunchKernelGGL(conv_in_kernel, grid, dim3(256), smem, s, x, wp, bias, o32, o16, oh, gn_partial, gn_cpg, H, W, Cout);
    return hipGetLastError();
}

hipError_t vt_launch_conv_in_mfma(const float* x, const bf16_t* wpk, const float* bias, float* o32, bf16_t* o16, f16_t* oh,
                                  float* gn_partial, int* gn_parts, int B, int H, int W, hipStream_t s) {
    if (!x || !wpk || !bias || (!o32 && !o16 && !oh) || B <= 0 || H <= 0 || W <= 0) return hipErrorInvalidValue;
    dim3 grid((W + CM_PIX - 1) / CM_PIX, (H + CM_ROWS - 1) / CM_ROWS, B);
    if (gn_parts) *gn_parts = grid.x * grid.y;
    hipLaunchKernelGGL(conv_in_mfma_kernel, grid, dim3(256), 0, s, x, wpk, bias, o32, o16, oh, gn_partial, H, W);
    return hipGetLastError();
}
int vt_conv_in_mfma_parts(int H, int W) { return ((W + CM_PIX - 1) / CM_PIX) * ((H + CM_ROWS - 1) / CM_ROWS); }

int vt_conv_in_parts(int H, int W) { return ((W + CI_PIX - 1) / CI_PIX) * ((H + CI_ROWS - 1) / CI_ROWS); }

// fp16 scores, 16-B loads and stores: thread t owns values [(k*256 + t)*8, +8) of the row, k < NV (n <= NV*2048).
// Requires lds % 8 == 0, ldp % 8 == 0 and 16-B aligned bases.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_f16x8_kernel(const f16_t* __restrict__ s, bf16_t* __restrict__ p,
                                                                 int n, int lds, int ldp) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const f16_t* sr = s + row * lds;
    bf16_t* pr = p + row * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float v[NV][8];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (k * 256 + tid) * 8;
        if (i + 7 < n) {
            const f16x8 h = *(const f16x8*)(sr + i);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[k][r] = (float)h[r];
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[k][r] = (i + r < n) ? (float)sr[i + r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) m = fmaxf(m, v[k][r]);
    }
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[k][r] = __expf(v[k][r] - m); sum += v[k][r]; }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = (k * 256 + tid) * 8;
        if (i < ldp) {
            bf16x8 h;
#pragma unroll
            for (int r = 0; r < 8; ++r) h[r] = (bf16_t)(v[k][r] * inv);
            *(bf16x8*)(pr + i) = h;
        }
    }
}

template <typename T>
static hipError_t softmax_dispatch(const T* scores, bf16_t* probs, long long rows, int n, int lds, int ldp, hipStream_t s) {
    const bool vec = (lds % 4 == 0) && (ldp % 4 == 0) && (((uintptr_t)scores) % 16 == 0) && (((uintptr_t)probs) % 8 == 0);
    const int need = (ldp + 1023) / 1024;           // NV covers ldp so the pad columns are written too
    const dim3 grid((unsigned)rows);
#define SMX(NV) hipLaunchKernelGGL((softmax_rows_cached_kernel<T, NV>), grid, dim3(256), 0, s, scores, probs, n, lds, ldp)
    if (vec && need <= 1) SMX(1);
    else if (vec && need <= 2) SMX(2);
    else if (vec && need <= 4) SMX(4);
    else if (vec && need <= 8) SMX(8);
    else if (vec && need <= 16) SMX(16);
    else hipLaunchKernelGGL((softmax_rows_kernel<T>), grid, dim3(256), 0, s, scores, probs, n, lds, ldp);
#undef SMX
    return hipGetLastError();
}

hipError_t vt_launch_softmax_rows(const void* scores, int scores_f16, bf16_t* probs, long long rows, int n, int lds, int ldp,
                                  hipStream_t s) {
    if (!scores || !probs || rows <= 0 || rows > 0x7fffffffLL || n <= 0 || lds < n || ldp < n) return hipErrorInvalidValue;
    if (scores_f16 && (lds % 8 == 0) && (ldp % 8 == 0) && (((uintptr_t)scores) % 16 == 0) && (((uintptr_t)probs) % 16 == 0) &&
        ldp <= 8 * 2048) {
        const int need = (ldp + 2047) / 2048;
        const dim3 grid((unsigned)rows);
#define SMH(NV) hipLaunchKernelGGL((softmax_rows_f16x8_kernel<NV>), grid, dim3(256), 0, s, (const f16_t*)scores, probs, n, lds, ldp)
        if (need <= 1) SMH(1); else if (need <= 2) SMH(2); else if (need <= 4) SMH(4); else SMH(8);
#undef SMH
        return hipGetLastError();
    }
    return scores_f16 ? softmax_dispatch((const f16_t*)scores, probs, rows, n, lds, ldp, s)
                      : softmax_dispatch((const float*)scores, probs, rows, n, lds, ldp, s);
}

// ---- attention without a softmax pass (capi.hip run_attention) -------------------------------------------------------------
// softmax(s)_ij = exp(s_ij - c_i) / sum_j exp(s_ij - c_i) for ANY per-row c_i, so the Q.K^T epilogue can emit the
// numerators directly once a c_i is known that keeps them inside the float range.  From the operands alone:
// s_ij <= |q_i| max_j|k_j| alpha =: u_i (Cauchy-Schwarz) and max_j s_ij >= s_ii =: l_i (self-attention: the diagonal exists).
// With c_i = (u_i + l_i) / 2 every numerator is <= exp((u_i - l_i) / 2) and the diagonal one is >= exp(-(u_i - l_i) / 2):
// while u_i - l_i <= 120 nothing overflows (16384 * e^60 * |v| << 3e38) and no row sum underflows.  Beyond that
// the launch group is flagged and c_i becomes the exact row maximum from an extra Q.K^T pass.
namespace {

// one wave per token: |q|^2, |k|^2, q.k of row i of the [rows][2C] q|k buffer
__global__ __launch_bounds__(256) void attn_row_norms_kernel(const bf16_t* __restrict__ qk, long long rows, int C,
                                                             float* __restrict__ qn, float* __restrict__ kn, float* __restrict__ sd) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* q = qk + row * 2 * C;
    float a = 0.f, b = 0.f, d = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const bf16x8 qv = *(const bf16x8*)(q + c), kv = *(const bf16x8*)(q + C + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float x = (float)qv[r], y = (float)kv[r];
            a += x * x; b += y * y; d += x * y;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); d += __shfl_xor(d, o); }
    if (lane == 0) { qn[row] = a; kn[row] = b; sd[row] = d; }
}

// the same with the fp8 attention's operands as a by-product: q8 | k8 = e4m3(scale q | scale k) (saturated at +-448; a clamp raises
// status bit 1), and the norms are those of the QUANTISED values -- the Cauchy-Schwarz bound of the exponent shift then holds for
// the operands the MFMA actually multiplies
__global__ __launch_bounds__(256) VT_NO_PACKED_F32 void attn_row_norms_fp8_kernel(const bf16_t* __restrict__ qk, long long rows, int C, float scale,
                                                                 unsigned char* __restrict__ qk8, float* __restrict__ qn,
                                                                 float* __restrict__ kn, float* __restrict__ sd, int* __restrict__ status) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* q = qk + row * 2 * C;
    unsigned char* o = qk8 + row * 2 * C;
    const float inv = 1.0f / scale;
    float a = 0.f, b = 0.f, d = 0.f, amax = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int c = lane * 8; c < C; c += 512) {
        const bf16x8 qv = *(const bf16x8*)(q + c), kv = *(const bf16x8*)(q + C + c);
        int qw[2] = {0, 0}, kw[2] = {0, 0};
        // element pair p of the eight -> 16-bit half (p & 1) of word (p >> 1); the half selector of the conversions is an immediate
        auto pair = [&](auto p_tag) {
            constexpr int P = decltype(p_tag)::value;
            constexpr bool HI = (P & 1) != 0;
            float x0 = (float)qv[2 * P] * scale, x1 = (float)qv[2 * P + 1] * scale, y0 = (float)kv[2 * P] * scale, y1 = (float)kv[2 * P + 1] * scale;
            amax = fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fmaxf(fabsf(y0), fabsf(y1)), amax));
            x0 = __builtin_amdgcn_fmed3f(x0, -448.f, 448.f); x1 = __builtin_amdgcn_fmed3f(x1, -448.f, 448.f);
            y0 = __builtin_amdgcn_fmed3f(y0, -448.f, 448.f); y1 = __builtin_amdgcn_fmed3f(y1, -448.f, 448.f);
            qw[P >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, qw[P >> 1], HI);
            kw[P >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(y0, y1, kw[P >> 1], HI);
            const f32x2 xq = __builtin_amdgcn_cvt_pk_f32_fp8(qw[P >> 1], HI), yq = __builtin_amdgcn_cvt_pk_f32_fp8(kw[P >> 1], HI);
            const float u0 = xq[0] * inv, u1 = xq[1] * inv, w0 = yq[0] * inv, w1 = yq[1] * inv;
            a += u0 * u0 + u1 * u1; b += w0 * w0 + w1 * w1; d += u0 * w0 + u1 * w1;
        };
        pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 1>{});
        pair(std::integral_constant<int, 2>{}); pair(std::integral_constant<int, 3>{});
        *(int2*)(o + c) = make_int2(qw[0], qw[1]);
        *(int2*)(o + C + c) = make_int2(kw[0], kw[1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); d += __shfl_xor(d, off); }
    if (lane == 0) { qn[row] = a; kn[row] = b; sd[row] = d; }
    if (status && amax > 448.f) atomicOr(status, 2);
}

// one block per image: c_i and the "bound too loose" flag of the group the image belongs to
__global__ __launch_bounds__(1024) void attn_shift_kernel(const float* __restrict__ qn, const float* __restrict__ kn,
                                                          const float* __restrict__ sd, int S, float alpha, float max_gap,
                                                          float* __restrict__ shift, int* __restrict__ flags, int group) {
    __shared__ float red[16];
    const long long base = (long long)blockIdx.x * S;
    float m = 0.f;
    for (int i = threadIdx.x; i < S; i += 1024) m = fmaxf(m, kn[base + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    // a hair above the exact norms: the MFMA's fp32 accumulation order is not this kernel's
    const float kmax = sqrtf(m) * alpha * 1.0001f;
    bool loose = false;
    for (int i = threadIdx.x; i < S; i += 1024) {
        const float u = sqrtf(qn[base + i]) * kmax, l = sd[base + i] * alpha;
        shift[base + i] = 0.5f * (u + l);
        loose |= !(u - l <= max_gap);              // NaN-safe: anything odd takes the exact path
    }
    if (__syncthreads_or(loose) && threadIdx.x == 0) atomicOr(flags + blockIdx.x / group, 1);
}

// row_part [batch][slots][row_bs] -> out [batch][row_bs]: the maximum (op 0) or 1 / sum (op 1) over the slots.
// 64 rows per block, the slots dealt to 4 waves and combined in a fixed order (deterministic sums).
__global__ __launch_bounds__(256) void attn_row_reduce_kernel(const float* __restrict__ part, int slots, long long row_bs, int S,
                                                              int op, float* __restrict__ out, const int* gate, int gate_expect) {
    if (gate && *gate != gate_expect) return;
    __shared__ float red[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float r = op == 0 ? -__builtin_inff() : 0.f;
    if (i < S) {
        const float* p = part + (long long)blockIdx.y * slots * row_bs + i;
        for (int s = g; s < slots; s += 4) {
            const float v = p[(long long)s * row_bs];
            r = op == 0 ? fmaxf(r, v) : r + v;
        }
    }
    red[g][threadIdx.x & 63] = r;
    __syncthreads();
    if (g == 0 && i < S) {
        const int l = threadIdx.x;
        r = op == 0 ? fmaxf(fmaxf(red[0][l], red[1][l]), fmaxf(red[2][l], red[3][l])) : (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        out[(long long)blockIdx.y * row_bs + i] = op == 0 ? r : 1.f / r;
    }
}

}  // namespace

hipError_t vt_launch_attn_row_norms(const bf16_t* qk, long long rows, int C, float* qn, float* kn, float* sd, hipStream_t s) {
    if (!qk || !qn || !kn || !sd || rows <= 0 || rows > 0x7fffffffLL || C <= 0 || (C % 8)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_row_norms_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, qk, rows, C, qn, kn, sd);
    return hipGetLastError();
}
hipError_t vt_launch_attn_row_norms_fp8(const bf16_t* qk, long long rows, int C, float scale, unsigned char* qk8, float* qn, float* kn, float* sd,
                                        int* status, hipStream_t s) {
    if (!qk || !qk8 || !qn || !kn || !sd || rows <= 0 || rows > 0x7fffffffLL || C <= 0 || (C % 8) || !(scale > 0.f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_row_norms_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, qk, rows, C, scale, qk8, qn, kn, sd, status);
    return hipGetLastError();
}
hipError_t vt_launch_attn_shift(const float* qn, const float* kn, const float* sd, int images, int S, float alpha, float max_gap,
                                float* shift, int* flags, int group, hipStream_t s) {
    if (!qn || !kn || !sd || !shift || !flags || images <= 0 || S <= 0 || group <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_shift_kernel, dim3(images), dim3(1024), 0, s, qn, kn, sd, S, alpha, max_gap, shift, flags, group);
    return hipGetLastError();
}
hipError_t vt_launch_attn_row_reduce(const float* part, int slots, long long row_bs, int S, int batch, int op, float* out,
                                     const int* gate, int gate_expect, hipStream_t s) {
    if (!part || !out || slots <= 0 || S <= 0 || batch <= 0 || row_bs < S) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_row_reduce_kernel, dim3((S + 63) / 64, batch), dim3(256), 0, s, part, slots, row_bs, S, op, out, gate,
                       gate_expect);
    return hipGetLastError();
}
