// Classification decoders (reference modules.py:303-475), fp32 end to end.
// ~50 MFLOP / image against ~4.9 TFLOP for the encoder: launch-latency / weight-read bound, so
// these are plain VALU kernels, one small launch per reference sub-module.
//   SpatialAttention.forward          modules.py:36-47   -> pool, gate, spmap, sgate
//   feature_compress                  modules.py:377-382 -> compress (conv3x3 + folded BN + ReLU + adaptive avg pool)
//   MultiHeadSelfAttention.forward    modules.py:66-91   -> self_attn
//   classifier                        modules.py:401-418 -> linear, ln_act
//   get_confidence                    modules.py:470-475 -> sort (descending logit, ascending index on ties)
#include "vt_common.h"
#include "vt_decoder.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// avg / max over H*W per (b, c)   (nn.AdaptiveAvgPool2d(1), nn.AdaptiveMaxPool2d(1))
__global__ __launch_bounds__(256) void dec_pool_kernel(const float* __restrict__ x, int HW, float* __restrict__ pool) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y, C = gridDim.x;
    const float* xp = x + ((long long)b * C + c) * HW;
    float s = 0.f, m = -INFINITY;
    for (int i = threadIdx.x; i < HW; i += 256) { const float v = xp[i]; s += v; m = fmaxf(m, v); }
    s = block_sum_256(s, red);
    m = block_max_256(m, red);
    if (threadIdx.x == 0) { pool[(b * C + c) * 2] = s / (float)HW; pool[(b * C + c) * 2 + 1] = m; }
}

// channel gate = sigmoid(mlp(avg) + mlp(max)), mlp = 1x1 conv C->C/r, ReLU, 1x1 conv C/r->C, no bias
__global__ void dec_gate_kernel(const float* __restrict__ pool, const float* __restrict__ w0,
                                const float* __restrict__ w2, int C, int R, float* __restrict__ gate) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    float out = 0.f;
    for (int which = 0; which < 2; ++which) {
        for (int r = 0; r < R; ++r) {
            float h = 0.f;
            for (int i = 0; i < C; ++i) h = fmaf(w0[r * C + i], pool[(b * C + i) * 2 + which], h);
            h = fmaxf(h, 0.f);
            out = fmaf(w2[c * R + r], h, out);
        }
    }
    gate[b * C + c] = vt_sigmoid_accurate(out);
}

// mean / max over channels of x*gate -> sp[b][2][HW]
__global__ __launch_bounds__(256) void dec_spmap_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                        int C, int HW, float* __restrict__ sp) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float s = 0.f, m = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float v = x[((long long)b * C + c) * HW + p] * gate[b * C + c];
        s += v; m = fmaxf(m, v);
    }
    sp[((long long)b * 2) * HW + p] = s / (float)C;
    sp[((long long)b * 2 + 1) * HW + p] = m;
}

// spatial gate = sigmoid(conv7x7(sp), pad 3, no bias)
__global__ __launch_bounds__(256) void dec_sgate_kernel(const float* __restrict__ sp, const float* __restrict__ w,
                                                        int H, int W, float* __restrict__ sg) {
    __shared__ float sw[98];
    if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float acc = 0.f;
    for (int c = 0; c < 2; ++c)
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = x + kx - 3;
                if (ix < 0 || ix >= W) continue;
                acc = fmaf(sw[(c * 7 + ky) * 7 + kx], sp[((long long)b * 2 + c) * HW + iy * W + ix], acc);
            }
        }
    sg[(long long)b * HW + p] = vt_sigmoid_accurate(acc);
}

// feature_compress: conv3x3(C->CO, pad 1, bias) -> BN(eval, folded) -> ReLU -> AdaptiveAvgPool(8,8).
// Input is x * gate[c] * sg[p] (SpatialAttention output) computed on the fly; gate/sg may be null.
// One workgroup per (pooled cell, image).
template <int CO>
__global__ __launch_bounds__(256) void dec_compress_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                           const float* __restrict__ sg, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift, int C, int H, int W,
                                                           float* __restrict__ pooled) {
    extern __shared__ float smw[];          // [CO][C][9]
    __shared__ float red[4];
    __shared__ float sgate[64];
    const int cell = blockIdx.x, b = blockIdx.y, HW = H * W;
    const int cy = cell >> 3, cx = cell & 7;
    for (int i = threadIdx.x; i < CO * C * 9; i += 256) smw[i] = w[i];
    if (threadIdx.x < C) sgate[threadIdx.x] = gate ? gate[b * C + threadIdx.x] : 1.f;
    __syncthreads();
    const int y0 = (cy * H) / 8, y1 = ((cy + 1) * H + 7) / 8;
    const int x0 = (cx * W) / 8, x1 = ((cx + 1) * W + 7) / 8;
    const int ch = y1 - y0, cw = x1 - x0, n = ch * cw;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int y = y0 + i / cw, xx = x0 + i % cw;
        float v[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) v[o] = bias[o];
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = xx + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const int q = iy * W + ix;
                const float sgv = sg ? sg[(long long)b * HW + q] : 1.f;
                for (int c = 0; c < C; ++c) {
                    // (x * channel_att) * spatial_att, same association as modules.py:41,47
                    const float in = (x[((long long)b * C + c) * HW + q] * sgate[c]) * sgv;
#pragma unroll
                    for (int o = 0; o < CO; ++o) v[o] = fmaf(smw[(o * C + c) * 9 + ky * 3 + kx], in, v[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] += fmaxf(fmaf(v[o], bn_scale[o], bn_shift[o]), 0.f);
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float s = block_sum_256(acc[o], red);
        if (threadIdx.x == 0) pooled[((long long)b * CO + o) * 64 + cell] = s / (float)n;
    }
}

// generic AdaptiveAvgPool2d [B,C,H,W] -> [B,C,OH,OW] (plain decoder: 4x4)
__global__ void dec_adaptive_pool_kernel(const float* __restrict__ x, int C, int H, int W, int OH, int OW,
                                         float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * OH * OW) return;
    const int c = i / (OH * OW), oy = (i / OW) % OH, ox = i % OW;
    const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
    const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int xx = x0; xx < x1; ++xx) s += x[(((long long)b * C + c) * H + y) * W + xx];
    out[(long long)b * C * OH * OW + i] = s / (float)((y1 - y0) * (x1 - x0));
}

// MultiHeadSelfAttention on 64 tokens x E dims (E <= 16), one token per lane; output C-major [b][e][token]
// in place over `t` (same layout as the pooled features).
// EC / HC > 0: E and heads as compile-time constants (the reference's 8 x 8: every array index folds, nothing lives in scratch
// memory -- same operations in the same order, 113 -> ~25 us); 0: run-time values.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"      // (the run-time instantiation cannot unroll its loops: expected)
template <int EC, int HC>
__global__ __launch_bounds__(64) void dec_self_attn_kernel(float* __restrict__ t, const DecSelfAttnW wts, int E_rt,
                                                           int heads_rt) {
    const int E = EC > 0 ? EC : E_rt, heads = HC > 0 ? HC : heads_rt;
    __shared__ float sk[64][16], sv[64][16];
    const int b = blockIdx.x, tok = threadIdx.x;
    float xin[16], xn[16], q[16];
    float mean = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { xin[e] = t[((long long)b * E + e) * 64 + tok]; mean += xin[e]; }
    mean /= (float)E;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float d = xin[e] - mean; var = fmaf(d, d, var); }
    var /= (float)E;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int e = 0; e < E; ++e) xn[e] = (xin[e] - mean) * rstd * wts.ln_w[e] + wts.ln_b[e];
#pragma unroll
    for (int o = 0; o < E; ++o) {
        float aq = wts.q_b[o], ak = wts.k_b[o], av = wts.v_b[o];
    #pragma unroll
    for (int e = 0; e < E; ++e) {
            aq = fmaf(wts.q_w[o * E + e], xn[e], aq);
            ak = fmaf(wts.k_w[o * E + e], xn[e], ak);
            av = fmaf(wts.v_w[o * E + e], xn[e], av);
        }
        q[o] = aq; sk[tok][o] = ak; sv[tok][o] = av;
    }
    __syncthreads();
    const int hd = E / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    float att[16];
#pragma unroll
    for (int h = 0; h < heads; ++h) {
        float m = -INFINITY;
        for (int j = 0; j < 64; ++j) {
            float s = 0.f;
#pragma unroll
        for (int d = 0; d < hd; ++d) s = fmaf(q[h * hd + d], sk[j][h * hd + d], s);
            m = fmaxf(m, s * scale);
        }
        float den = 0.f, o[16];
#pragma unroll
        for (int d = 0; d < hd; ++d) o[d] = 0.f;
        for (int j = 0; j < 64; ++j) {
            float s = 0.f;
#pragma unroll
        for (int d = 0; d < hd; ++d) s = fmaf(q[h * hd + d], sk[j][h * hd + d], s);
            const float p = expf(s * scale - m);
            den += p;
#pragma unroll
        for (int d = 0; d < hd; ++d) o[d] = fmaf(p, sv[j][h * hd + d], o[d]);
        }
#pragma unroll
        for (int d = 0; d < hd; ++d) att[h * hd + d] = o[d] / den;
    }
#pragma unroll
    for (int o = 0; o < E; ++o) {
        float a = wts.o_b[o];
    #pragma unroll
    for (int e = 0; e < E; ++e) a = fmaf(wts.o_w[o * E + e], att[e], a);
        t[((long long)b * E + o) * 64 + tok] = a + xin[o];     // residual is the un-normed input (modules.py:73,87)
    }
}

#pragma clang diagnostic pop

// y[b][o] = W[o][:] . x[b][:] + bias[o].  One wave per output neuron, weights read once per 8 images.
constexpr int LIN_BT = 8;
__global__ __launch_bounds__(256) void dec_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int B, int IN, int OUT) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= OUT) return;
    const float* wr = w + (long long)o * IN;
    for (int b0 = 0; b0 < B; b0 += LIN_BT) {
        float acc[LIN_BT];
#pragma unroll
        for (int i = 0; i < LIN_BT; ++i) acc[i] = 0.f;
        for (int k = lane; k < IN; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int i = 0; i < LIN_BT; ++i)
                if (b0 + i < B) acc[i] = fmaf(wv, x[(long long)(b0 + i) * IN + k], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < LIN_BT; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0 && b0 + i < B) y[(long long)(b0 + i) * OUT + o] = s + bias[o];
        }
    }
}

// in-place LayerNorm(eps 1e-5) + activation (0 = ReLU, 1 = LeakyReLU(0.2)), one workgroup per row
__global__ __launch_bounds__(256) void dec_ln_act_kernel(float* __restrict__ y, const float* __restrict__ g,
                                                         const float* __restrict__ bta, int N, int act) {
    __shared__ float red[4];
    float* r = y + (long long)blockIdx.x * N;
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) s += r[i];
    const float mean = block_sum_256(s, red) / (float)N;
    float v = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) { const float d = r[i] - mean; v = fmaf(d, d, v); }
    const float var = block_sum_256(v, red) / (float)N;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int i = threadIdx.x; i < N; i += 256) {
        float t = (r[i] - mean) * rstd * g[i] + bta[i];
        t = act == 0 ? fmaxf(t, 0.f) : (t > 0.f ? t : 0.2f * t);
        r[i] = t;
    }
}

// flat[b][:] += mean_j(att[b][j])   (cross-attention merge, modules.py:459)
__global__ __launch_bounds__(256) void dec_add_rowmean_kernel(float* __restrict__ flat, const float* __restrict__ att,
                                                              int NF, int NA) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < NA; i += 256) s += att[(long long)b * NA + i];
    const float m = block_sum_256(s, red) / (float)NA;
    for (int i = threadIdx.x; i < NF; i += 256) flat[(long long)b * NF + i] += m;
}

// CrossAttention core (modules.py:107-124): one query token per image against 64 key/value tokens.
// qv [B][E] (q_proj(query)), kv input tokens come from feat [B][CH][64] via k/v projections computed here.
__global__ __launch_bounds__(64) void dec_cross_attn_kernel(const float* __restrict__ qv, const float* __restrict__ feat,
                                                            const float* __restrict__ k_w, const float* __restrict__ k_b,
                                                            const float* __restrict__ v_w, const float* __restrict__ v_b,
                                                            int CH, int E, int heads, float* __restrict__ out) {
    // lane = key token j; loop over embedding dims grouped by head
    __shared__ float so[256];
    const int b = blockIdx.x, j = threadIdx.x;
    float f[16];
    for (int c = 0; c < CH; ++c) f[c] = feat[((long long)b * CH + c) * 64 + j];
    const int hd = E / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    for (int h = 0; h < heads; ++h) {
        float s = 0.f;
        for (int d = 0; d < hd; ++d) {
            const int e = h * hd + d;
            float kk = k_b[e];
            for (int c = 0; c < CH; ++c) kk = fmaf(k_w[e * CH + c], f[c], kk);
            s = fmaf(qv[(long long)b * E + e], kk, s);
        }
        s *= scale;
        const float m = wave_max(s);
        const float p = expf(s - m);
        const float den = wave_sum(p);
        for (int d = 0; d < hd; ++d) {
            const int e = h * hd + d;
            float vv = v_b[e];
            for (int c = 0; c < CH; ++c) vv = fmaf(v_w[e * CH + c], f[c], vv);
            const float o = wave_sum(p * vv) / den;
            if (j == 0) so[e] = o;
        }
    }
    __syncthreads();
    for (int e = j; e < E; e += 64) out[(long long)b * E + e] = so[e];
}

__global__ void dec_add_kernel(float* __restrict__ a, const float* __restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// ---- get_confidence: descending sort of (logit, tag index) per image ------------------------------------------------
// One 64-bit key per tag whose unsigned order IS the output order: high word = the logit mapped to a monotone unsigned
// (NaN -> 1: after every real value, -inf included), low word = ~index (equal logits: ascending index).  Keys are unique, so
// the order is strict and total.  The network is the single-direction ("flip") bitonic sort: stage k compares i with
// i ^ (2k' - 1) first and then i ^ j for j = k'/2 .. 1, every exchange in the same direction, so tags beyond N are virtual
// minimum keys that never move and N needs no power-of-two padding.  Up to SORT_CH tags the whole sort runs in LDS in one
// launch; beyond that each 16384-block is sorted in LDS, the j > SORT_CH/2 steps of the later stages run as global-memory
// passes over the key array (kept in the int64 index output) and each stage's tail runs in LDS again.
constexpr int SORT_CH = 16384;

__device__ __forceinline__ unsigned long long sort_key(float f, int i) {
    unsigned u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;                             // -0.0 ties with +0.0, as in a comparison sort
    u = (f != f) ? 1u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    return ((unsigned long long)u << 32) | (unsigned)(~i);
}
__device__ __forceinline__ float sort_key_logit(unsigned long long k) {
    const unsigned u = (unsigned)(k >> 32);
    if (u == 1u) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// MODE 0: build the keys of block c from the logits and sort the block (stages k = 2 .. SORT_CH).
// MODE 1: load the keys of block c and run the steps j = SORT_CH/2 .. 1 of a later stage.
// `final`: write sigmoid(conf) / int64 indices instead of keys.
template <int MODE>
__global__ __launch_bounds__(1024) void dec_sort_local_kernel(const float* __restrict__ logits, int N,
                                                              unsigned long long* __restrict__ keys,
                                                              float* __restrict__ conf, long long* __restrict__ idx, int final) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned long long* key = (unsigned long long*)sm;
    const int b = blockIdx.y, base = blockIdx.x * SORT_CH;
    const int n = min(SORT_CH, N - base);                    // real elements of this block
    int np = 2;
    while (np < n) np <<= 1;
    for (int i = threadIdx.x; i < np; i += 1024) {
        unsigned long long k = 0ull;
        if (i < n) k = MODE == 0 ? sort_key(logits[(long long)b * N + base + i], base + i) : keys[(long long)b * N + base + i];
        key[i] = k;
    }
    __syncthreads();
    auto step = [&](int j, bool flip, int kk) {
        for (int i = threadIdx.x; i < np; i += 1024) {
            const int l = flip ? (i ^ (kk - 1)) : (i ^ j);
            if (l > i && l < n) {
                const unsigned long long a = key[i], c = key[l];
                if (a < c) { key[i] = c; key[l] = a; }       // descending
            }
        }
        __syncthreads();
    };
    if (MODE == 0) {
        for (int k = 2; k <= np; k <<= 1) {
            step(0, true, k);
            for (int j = k >> 2; j > 0; j >>= 1) step(j, false, 0);
        }
    } else {
        for (int j = SORT_CH >> 1; j > 0; j >>= 1) step(j, false, 0);
    }
    for (int i = threadIdx.x; i < n; i += 1024) {
        const long long o = (long long)b * N + base + i;
        if (final) {
            conf[o] = vt_sigmoid_accurate(sort_key_logit(key[i]));
            idx[o] = (long long)(int)(~(unsigned)key[i]);
        } else {
            keys[o] = key[i];
        }
    }
}

// one global-memory step of stage k: flip (partner i ^ (k - 1)) or plain (partner i ^ j); pairs with a partner >= N stay
__global__ __launch_bounds__(256) void dec_sort_global_kernel(unsigned long long* __restrict__ keys, int N, int k, int j, int flip) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;            // pair number
    const int half = flip ? (k >> 1) : j;
    const int i = ((t / half) * 2) * half + (t % half);      // lower element of pair t
    const int l = flip ? (i ^ (k - 1)) : (i ^ j);
    if (l >= N || i >= N) return;
    unsigned long long* kb = keys + (long long)b * N;
    const unsigned long long a = kb[i], c = kb[l];
    if (a < c) { kb[i] = c; kb[l] = a; }
}

// ---- per-image summary of the sorted confidences (infer_full.py:106-125) on the device: the number of tags at or above the
// threshold, the first K (confidence, index) pairs, the maximum and the top-5 mean (the sum of the first five divided by 5,
// whatever N is) -- B x (2 K + 4) values cross PCIe instead of B x N x 12 bytes.
// stats[b] = { count >= threshold, max confidence, top-5 sum / 5, number of non-finite confidences }
__global__ __launch_bounds__(256) void dec_summary_kernel(const float* __restrict__ conf, const long long* __restrict__ idx, int N,
                                                          float threshold, int K, float* __restrict__ top_conf,
                                                          int* __restrict__ top_idx, float* __restrict__ stats) {
    __shared__ int s_cnt, s_bad;
    const int b = blockIdx.x;
    const float* c = conf + (long long)b * N;
    const long long* ix = idx + (long long)b * N;
    if (threadIdx.x == 0) { s_cnt = 0; s_bad = 0; }
    __syncthreads();
    int cnt = 0, bad = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = c[i];
        cnt += v >= threshold;                               // the reference's test, element by element (NaN fails it)
        bad += !(fabsf(v) <= 3.0e38f);
    }
    atomicAdd(&s_cnt, cnt);
    atomicAdd(&s_bad, bad);
    for (int i = threadIdx.x; i < K; i += 256) {
        top_conf[(long long)b * K + i] = i < N ? c[i] : 0.f;
        top_idx[(long long)b * K + i] = i < N ? (int)ix[i] : -1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s5 = 0.f;
        for (int i = 0; i < 5 && i < N; ++i) s5 += c[i];
        float mx = c[0];
        float* o = stats + (long long)b * 4;
        o[0] = (float)s_cnt; o[1] = mx; o[2] = s5 / 5.0f; o[3] = (float)s_bad;
    }
}

}  // namespace

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) return _e; } while (0)
#define CKL() CK(hipGetLastError())

hipError_t vt_decoder_forward(const DecoderWeights& w, const float* latent, int B, int H, int Wd, float* ws,
                              float* logits, hipStream_t s) {
    const int C = w.latent_channels, HW = H * Wd;
    if (B <= 0 || H <= 0 || Wd <= 0 || C <= 0 || C > 64) return hipErrorInvalidValue;
    if (w.plain) {
        float* pooled = ws;                       // [B][C*16]
        float* h0 = pooled + (size_t)B * C * 16;  // [B][512]
        float* h1 = h0 + (size_t)B * 512;         // [B][256]
        const int n = C * 16;
        hipLaunchKernelGGL(dec_adaptive_pool_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, latent, C, H, Wd, 4, 4, pooled); CKL();
        hipLaunchKernelGGL(dec_linear_kernel, dim3((512 + 3) / 4), dim3(256), 0, s, pooled, w.cls_w[0], w.cls_b[0], h0, B, n, 512); CKL();
        hipLaunchKernelGGL(dec_ln_act_kernel, dim3(B), dim3(256), 0, s, h0, w.cls_ln_w[0], w.cls_ln_b[0], 512, 1); CKL();
        hipLaunchKernelGGL(dec_linear_kernel, dim3((256 + 3) / 4), dim3(256), 0, s, h0, w.cls_w[1], w.cls_b[1], h1, B, 512, 256); CKL();
        hipLaunchKernelGGL(dec_ln_act_kernel, dim3(B), dim3(256), 0, s, h1, w.cls_ln_w[1], w.cls_ln_b[1], 256, 1); CKL();
        hipLaunchKernelGGL(dec_linear_kernel, dim3((w.num_classes + 3) / 4), dim3(256), 0, s, h1, w.cls_w[2], w.cls_b[2], logits, B, 256, w.num_classes); CKL();
        return hipSuccess;
    }
    const int CO = C / 2;
    if (CO != 8) return hipErrorInvalidValue;     // compress kernel is instantiated for the reference's 16 -> 8
    float* p = ws;
    float* pool = p; p += (size_t)B * C * 2;
    float* gate = p; p += (size_t)B * C;
    float* sp = p; p += (size_t)B * 2 * HW;
    float* sg = p; p += (size_t)B * HW;
    float* feat = p; p += (size_t)B * CO * 64;      // [B][CO][64]  == flattened [B][512]
    float* h0 = p; p += (size_t)B * 1024;
    float* h1 = p; p += (size_t)B * 512;
    float* h2 = p; p += (size_t)B * 256;
    float* cq = p; p += (size_t)B * 512;
    float* cqp = p; p += (size_t)B * 256;
    float* co = p; p += (size_t)B * 256;
    float* cat = p; p += (size_t)B * 512;
    const float* gptr = nullptr;
    const float* sgptr = nullptr;
    if (w.use_spatial) {
        hipLaunchKernelGGL(dec_pool_kernel, dim3(C, B), dim3(256), 0, s, latent, HW, pool); CKL();
        hipLaunchKernelGGL(dec_gate_kernel, dim3(B), dim3(64), 0, s, pool, w.ca_w0, w.ca_w2, C, w.ca_hidden, gate); CKL();
        hipLaunchKernelGGL(dec_spmap_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, latent, gate, C, HW, sp); CKL();
        hipLaunchKernelGGL(dec_sgate_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, sp, w.sa_w, H, Wd, sg); CKL();
        gptr = gate; sgptr = sg;
    }
    hipLaunchKernelGGL((dec_compress_kernel<8>), dim3(64, B), dim3(256), (size_t)CO * C * 9 * sizeof(float), s, latent, gptr, sgptr,
                       w.fc_w, w.fc_b, w.bn_scale, w.bn_shift, C, H, Wd, feat); CKL();
    if (w.use_self) {
        if (CO % w.heads) return hipErrorInvalidValue;
        if (CO == 8 && w.heads == 8) hipLaunchKernelGGL((dec_self_attn_kernel<8, 8>), dim3(B), dim3(64), 0, s, feat, w.sa, CO, w.heads);
        else hipLaunchKernelGGL((dec_self_attn_kernel<0, 0>), dim3(B), dim3(64), 0, s, feat, w.sa, CO, w.heads);
        CKL();
    }
    if (w.use_cross) {
        // query = query_generator(flat); attended = out_proj(attn(q_proj(query), k/v(spatial))) + query;
        // flat += mean(attended)   (modules.py:451-459)
        hipLaunchKernelGGL(dec_linear_kernel, dim3(512 / 4), dim3(256), 0, s, feat, w.qg_w, w.qg_b, cq, B, CO * 64, 512); CKL();
        hipLaunchKernelGGL(dec_linear_kernel, dim3(256 / 4), dim3(256), 0, s, cq, w.cx_q_w, w.cx_q_b, cqp, B, 512, 256); CKL();
        hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(B), dim3(64), 0, s, cqp, feat, w.cx_k_w, w.cx_k_b, w.cx_v_w, w.cx_v_b, CO, 256, w.heads, co); CKL();
        hipLaunchKernelGGL(dec_linear_kernel, dim3(512 / 4), dim3(256), 0, s, co, w.cx_o_w, w.cx_o_b, cat, B, 256, 512); CKL();
        hipLaunchKernelGGL(dec_add_kernel, dim3((B * 512 + 255) / 256), dim3(256), 0, s, cat, cq, (long long)B * 512); CKL();
        hipLaunchKernelGGL(dec_add_rowmean_kernel, dim3(B), dim3(256), 0, s, feat, cat, CO * 64, 512); CKL();
    }
    hipLaunchKernelGGL(dec_linear_kernel, dim3(1024 / 4), dim3(256), 0, s, feat, w.cls_w[0], w.cls_b[0], h0, B, CO * 64, 1024); CKL();
    hipLaunchKernelGGL(dec_ln_act_kernel, dim3(B), dim3(256), 0, s, h0, w.cls_ln_w[0], w.cls_ln_b[0], 1024, 0); CKL();
    hipLaunchKernelGGL(dec_linear_kernel, dim3(512 / 4), dim3(256), 0, s, h0, w.cls_w[1], w.cls_b[1], h1, B, 1024, 512); CKL();
    hipLaunchKernelGGL(dec_ln_act_kernel, dim3(B), dim3(256), 0, s, h1, w.cls_ln_w[1], w.cls_ln_b[1], 512, 0); CKL();
    hipLaunchKernelGGL(dec_linear_kernel, dim3(256 / 4), dim3(256), 0, s, h1, w.cls_w[2], w.cls_b[2], h2, B, 512, 256); CKL();
    hipLaunchKernelGGL(dec_ln_act_kernel, dim3(B), dim3(256), 0, s, h2, w.cls_ln_w[2], w.cls_ln_b[2], 256, 0); CKL();
    hipLaunchKernelGGL(dec_linear_kernel, dim3((w.num_classes + 3) / 4), dim3(256), 0, s, h2, w.cls_w[3], w.cls_b[3], logits, B, 256, w.num_classes); CKL();
    return hipSuccess;
}

size_t vt_decoder_workspace_floats(int B, int C, int H, int Wd) {
    const size_t HW = (size_t)H * Wd;
    return (size_t)B * (C * 2 + C + 2 * HW + HW + (C / 2) * 64 + 1024 + 512 + 256 + 512 + 256 + 256 + 512 + C * 16 + 512 + 256) + 64;
}

hipError_t vt_decoder_sort(const float* logits, int B, int N, float* conf, long long* idx, hipStream_t s) {
    if (N <= 0 || B <= 0) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    CK(vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)dec_sort_local_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_sort_local_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8);
        return e;
    }));
    const int chunks = (N + SORT_CH - 1) / SORT_CH;
    unsigned long long* keys = (unsigned long long*)idx;      // the int64 index output doubles as the key array
    int np = 2;
    while (np < (N < SORT_CH ? N : SORT_CH)) np <<= 1;
    const size_t smem = (size_t)np * 8;
    hipLaunchKernelGGL(dec_sort_local_kernel<0>, dim3(chunks, B), dim3(1024), smem, s, logits, N, keys, conf, idx, chunks == 1 ? 1 : 0); CKL();
    if (chunks == 1) return hipSuccess;
    long long top = SORT_CH;
    while (top < N) top <<= 1;
    const unsigned pair_blocks = (unsigned)((top / 2 + 255) / 256);
    for (long long k = 2LL * SORT_CH; k <= top; k <<= 1) {
        hipLaunchKernelGGL(dec_sort_global_kernel, dim3(pair_blocks, B), dim3(256), 0, s, keys, N, (int)k, 0, 1); CKL();
        for (long long j = k >> 2; j >= SORT_CH; j >>= 1) {
            hipLaunchKernelGGL(dec_sort_global_kernel, dim3(pair_blocks, B), dim3(256), 0, s, keys, N, (int)k, (int)j, 0); CKL();
        }
        hipLaunchKernelGGL(dec_sort_local_kernel<1>, dim3(chunks, B), dim3(1024), (size_t)SORT_CH * 8, s, logits, N, keys, conf, idx, k == top ? 1 : 0); CKL();
    }
    return hipSuccess;
}

hipError_t vt_decoder_summary(const float* conf, const long long* idx, int B, int N, float threshold, int K, float* top_conf,
                              int* top_idx, float* stats, hipStream_t s) {
    if (B <= 0 || N <= 0 || K <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_summary_kernel, dim3(B), dim3(256), 0, s, conf, idx, N, threshold, K, top_conf, top_idx, stats); CKL();
    return hipSuccess;
}
