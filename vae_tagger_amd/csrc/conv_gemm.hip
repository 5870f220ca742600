// Implicit-GEMM convolution / batched NT-GEMM on CDNA4 matrix cores (gfx950).
//
// One kernel serves every contraction on the encoder path:
//   3x3 stride-1 pad-1 convs, 3x3 stride-2 convs with the asymmetric (0,1,0,1) pad, 1x1 shortcut
//   convs, the attention projections, Q.K^T and P.V.  All are  out[p][c] = sum_k X[p,k] * W[c,k]
//   with X rows gathered on the fly (im2col never materialised):
//     k = (tap, ci);  X[p,(tap,ci)] = x[b, oy*stride-pad+ky, ox*stride-pad+kx, ci]  (0 outside)
//
// Layout: activations NHWC bf16 (channels contiguous => one 16-B lane load = 8 k-values of one
// pixel), weights [Cout][tap][Cin] bf16 (k contiguous per cout).  Both operands therefore have
// the "8 contiguous k per lane" shape v_mfma_f32_16x16x32_bf16 wants, for A and for B.
//
// MFMA orientation: weights are the A operand (rows = cout), pixels the B operand (cols = pixel),
// so a lane's 4 accumulator registers are 4 CONSECUTIVE couts of ONE pixel -> 8-B/16-B vector
// stores into NHWC rows, float4 residual loads, and GroupNorm partial sums that stay in-lane.
//
// Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4): one wave-instruction writes 8 rows
// x 128 B (BK = 64 bf16).  The DMA destination is lane-linear, so the bank-conflict swizzle
// (16-B chunk ^ (row & 7)) is applied to the per-lane SOURCE address and again on the ds_read
// side (guide rule 21).  Zero padding / tails: the lane's source points at a zero page.
// Two LDS stages; one barrier per K-step; the next K-step's DMA is issued right after the barrier
// and lands under the current step's MFMAs.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int ROWB = 128;         // bytes per LDS row = one K-step of a row: 64 bf16 or (F8) 128 e4m3 k-values

// OCC2: compiled for two workgroups per CU (<= 128 VGPRs; the launcher checks the LDS fits twice): the latency-bound
// short-K launches (1x1 shortcuts, attention projections, Q.K^T) overlap one workgroup's prologue / epilogue with the other's MFMAs.
// one output tile; `logical` = tile index in [0, ptiles * ctiles * batch)
// F8: both operands are OCP e4m3 bytes (a.X / a.W reinterpreted; every "element" offset is then a byte offset) and a K-step is
// ONE v_mfma_scale_f32_16x16x128_f8f6f4 per tile pair (both scales 2^0) instead of two v_mfma_f32_16x16x32_bf16: the same LDS and
// DMA bytes per K-step carry twice the K, which is what a fill-bound tile needs (the stride-2 convs of the fp8 mode, flag 11).
// F16: the 16-bit operands hold fp16 bits (v_mfma_f32_16x16x32_f16): conv_out in the fp16-operand mode of the convs (vt_set_flag 18).
template <int BP, int BC, int WP, int WC, bool F8 = false, bool F16 = false>
__device__ __forceinline__ void conv_gemm_tile(const ConvGemmArgs& a, char* smem, int logical) {
    static_assert(!(F8 && F16), "one operand type");
    static_assert(WP * WC == 8, "8 waves per workgroup");
    constexpr int ES = F8 ? 1 : 2;            // bytes per element
    constexpr int BK = ROWB / ES;             // k-values per K-step
    constexpr int CE = 16 / ES;               // elements per 16-B DMA chunk
    constexpr int TP = BP / WP / 16;          // 16-pixel MFMA tiles per wave
    constexpr int TC = BC / WC / 16;          // 16-cout MFMA tiles per wave
    constexpr int XI = BP / 8;                // DMA wave-instructions per X tile (8 rows each)
    constexpr int WI = BC / 8;
    constexpr int NXJ = (XI + 7) / 8;         // per wave
    constexpr int NWJ = (WI + 7) / 8;
    constexpr int STAGE_BYTES = (BP + BC) * ROWB;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WC, wc = wave % WC;

    // ---- tile coordinates (XCD-aware: cout tiles of one pixel tile + neighbours share an L2)
    const int HWo = a.Hout * a.Wout;
    const int ptiles = (HWo + BP - 1) / BP;
    const int ctiles = (a.Cout + BC - 1) / BC;
    const int per_img = ptiles * ctiles;
    const int b = logical / per_img;
    logical -= b * per_img;
    const int p0 = (logical / ctiles) * BP;
    const int c0 = (logical % ctiles) * BC;

    const char* Xb = (const char*)a.X + (long long)b * a.x_bs * ES;
    const char* Wb = (const char*)a.W + (long long)b * a.w_bs * ES;

    // ---- per-lane DMA bookkeeping.  Lane l of a wave-instruction writes LDS row (l>>3), physical
    // chunk (l&7); the logical chunk it must fetch is (l&7) ^ (row&7) = (l&7) ^ (l>>3).
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const int ntaps = a.ksize * a.ksize;
    const int ncb = (a.Cin + BK - 1) / BK;
    const bool chunk_tail_possible = (a.Cin % BK) != 0;

    int xoff[NXJ];          // element offset of (pixel's window origin, chunk) inside the image
    unsigned xmask[NXJ];    // bit t set <=> tap t of this row is inside the image
#pragma unroll
    for (int j = 0; j < NXJ; ++j) {
        const int r = (j * 8 + wave) * 8 + lrow;
        const int p = p0 + r;
        const bool rv = (r < BP) && (p < HWo);
        const int oy = p / a.Wout, ox = p - oy * a.Wout;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xoff[j] = (iy0 * a.Win + ix0) * a.ldx + lchunk * CE;
        unsigned m = 0;
        if (rv) {
            for (int t = 0; t < ntaps; ++t) {
                const int ky = t / a.ksize, kx = t - ky * a.ksize;
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) m |= 1u << t;
            }
        }
        xmask[j] = m;
    }
    int woff[NWJ];
    bool wvalid[NWJ];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
        const int rblk = j * 8 + wave;
        // 64-cout wave groups use the interleaved cout map of conv3x3_halo (LDS row 16*i + 4*q + r holds cout r + 4*i + 16*q):
        // MFMA tile i / register r of lane (fq, frow) is then cout 16*fq + 4*i + r -- 16 consecutive couts per lane
        const int rl = rblk * 8 + lrow;
        const int n = c0 + (TC == 4 ? (rl & ~63) + (rl & 3) + 4 * ((rl >> 4) & 3) + 16 * ((rl >> 2) & 3) : rl);
        wvalid[j] = (rblk < WI) && (n < a.Wrows);
        woff[j] = n * a.ldw + lchunk * CE;
    }

    auto stage = [&](int tap, int cb, int buf) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + BP * ROWB;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;      // tap < 9
        const int doff = (a.ksize == 1) ? 0 : (ky * a.Win + kx) * a.ldx;
        const int k0 = cb * BK;
        const bool cv = !chunk_tail_possible || (k0 + lchunk * CE < a.Cin);
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int rblk = j * 8 + wave;
            if (XI % 8 == 0 || rblk < XI) {
                const bool v = ((xmask[j] >> tap) & 1u) && cv;
                const void* src = v ? (const void*)(Xb + (long long)(xoff[j] + doff + k0) * ES) : a.zeros;
                if (a.x_stream) __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(xs + rblk * 1024), 16, 0, 2);   // nt
                else __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(xs + rblk * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NWJ; ++j) {
            const int rblk = j * 8 + wave;
            if (WI % 8 == 0 || rblk < WI) {
                const bool v = wvalid[j] && cv;
                const void* src = v ? (const void*)(Wb + (long long)(woff[j] + tap * a.Cin + k0) * ES) : a.zeros;
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(ws + rblk * 1024), 16, 0, 0);
            }
        }
    };

    // ---- fragment read offsets: lane reads row (tile_base + (lane&15)), logical chunk kk*4+(lane>>4),
    // stored at physical chunk ^ (row&7) = ^ (lane&7) (tile bases are multiples of 16).
    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int foff0 = frow * ROWB + (((0 + fq) ^ (lane & 7)) << 4);
    const int foff1 = frow * ROWB + (((4 + fq) ^ (lane & 7)) << 4);

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = ntaps * ncb;
    stage(0, 0, 0);
    int tap_n = 0, cb_n = 1;                 // coordinates of the NEXT K-step to stage
    if (cb_n == ncb) { cb_n = 0; tap_n = 1; }

    for (int t = 0; t < nk; ++t) {
        __syncthreads();                     // vmcnt(0) + barrier: tile t landed, buffer (t+1)&1 free
        if (t + 1 < nk) {
            stage(tap_n, cb_n, (t + 1) & 1);
            if (++cb_n == ncb) { cb_n = 0; ++tap_n; }
        }
        const char* xs = smem + (t & 1) * STAGE_BYTES + (wp * (BP / WP)) * ROWB;
        const char* ws = smem + (t & 1) * STAGE_BYTES + BP * ROWB + (wc * (BC / WC)) * ROWB;
        if constexpr (F8) {
            // lane (fq, frow): row frow, k = 32 fq .. + 31 = logical chunks 2 fq, 2 fq + 1 (physical ^ (row & 7))
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            typedef int i32x8 __attribute__((ext_vector_type(8)));
            const int fa = frow * ROWB + (((2 * fq) ^ (lane & 7)) << 4), fb = frow * ROWB + (((2 * fq + 1) ^ (lane & 7)) << 4);
            i32x8 wf[TC], xf[TP];
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const i32x4 lo = *(const i32x4*)(ws + i * 16 * ROWB + fa), hi = *(const i32x4*)(ws + i * 16 * ROWB + fb);
                wf[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const i32x4 lo = *(const i32x4*)(xs + j * 16 * ROWB + fa), hi = *(const i32x4*)(xs + j * 16 * ROWB + fb);
                xf[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[j], acc[i][j], 0, 0, 0, 127, 0, 127);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int fo = kk ? foff1 : foff0;
                bf16x8 wf[TC], xf[TP];
#pragma unroll
                for (int i = 0; i < TC; ++i) wf[i] = *(const bf16x8*)(ws + i * 16 * ROWB + fo);
#pragma unroll
                for (int j = 0; j < TP; ++j) xf[j] = *(const bf16x8*)(xs + j * 16 * ROWB + fo);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        if constexpr (F16) {
                            typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m, wf[i]), __builtin_bit_cast(f16x8m, xf[j]), acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                        }
                    }
            }
        }
    }

    // ---- epilogue: lane holds couts cg(i)..+3 (regs) of pixel p for every (i, j) tile; with the interleaved map (TC == 4)
    // the four tiles of a row are 16 consecutive couts -> two 16-B stores of 16-bit outputs per row instead of four 8-B ones.
    const long long ob = (long long)b * a.o_bs;
    const float* resb = a.res ? a.res + (long long)b * a.r_bs : nullptr;
    const f16_t* resh = a.res_f16 ? a.res_f16 + (long long)b * a.r_bs : nullptr;
    const bool wide16 = TC == 4 && (a.ldo % 8) == 0 && (a.o_bs % 8) == 0;
    // whole cout tile real + only 16-bit outputs: they leave through the LDS transpose below (full 128-B lines per store)
    const bool staged = wide16 && a.out_mode == 0 && (a.out_bf16 || a.out_f16) && c0 + BC <= a.Cout;
    unsigned valid = 0;
    const bool nomask = c0 + BC <= a.Wrows;                 // every column of this tile is a real one
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int p = p0 + wp * (BP / WP) + j * 16 + frow;
        if (p >= HWo) continue;
        valid |= 1u << j;
        const int cg0 = c0 + wc * (BC / WC) + (TC == 4 ? fq * 16 : fq * 4);
        const float rin = a.row_in ? a.row_in[(long long)b * a.row_bs + p] : 0.f;
        float racc = a.row_mode == 1 ? -__builtin_inff() : 0.f;
        const float alpha2 = a.alpha * 1.44269504f, rin2 = rin * 1.44269504f;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int cg = cg0 + (TC == 4 ? i * 4 : i * 16);
            if (cg >= a.Cout) continue;
            f32x4 v = acc[i][j] * a.alpha;
            if (a.col_scale) v *= *(const f32x4*)(a.col_scale + cg);       // e4m3 weights: per-cout scale / activation scale
            if (a.row_mode) {
                if (a.row_mode == 3) {
                    v *= rin;
                } else if (a.row_mode == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {      // exp(alpha*acc - rin) as one fma + v_exp_f32
                        const float e = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], alpha2, -rin2));
                        v[r] = (nomask || cg + r < a.Wrows) ? e : 0.f;
                        racc += v[r];                  // (the fp32 value: like a softmax pass that normalises before rounding)
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cg + r < a.Wrows) racc = fmaxf(racc, v[r]);
                    continue;
                }
            }
            if (a.bias_mode == 1) {
                const f32x4 bv = *(const f32x4*)(a.bias + cg);
                v += bv;
            } else if (a.bias_mode == 2) {
                const float bv = a.bias[p];
                v += f32x4{bv, bv, bv, bv};
            }
            if (a.out_mode == 1) {
                // latent: fp32 NCHW, first cout_keep channels, affine post-scale (DiffusersVAEWrapper.encode)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (cg + r < a.cout_keep)
                        a.out_f32[ob + (long long)(cg + r) * HWo + p] = v[r] * a.post_scale + a.post_shift;
                continue;
            }
            const long long o = (long long)p * a.ldo + cg;
            if (resb) {
                const f32x4 rv = *(const f32x4*)(resb + (long long)p * a.ldr + cg);
                v += rv;
            } else if (resh) {
                const f16x4 rh = *(const f16x4*)(resh + (long long)p * a.ldr + cg);
                v += f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
            }
            if (a.out_f32) *(f32x4*)(a.out_f32 + ob + o) = v;
            if (!staged && !(wide16 && cg0 + 16 <= a.Cout)) {
                if (a.out_bf16) {
                    bf16x4 h;
                    h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
                    *(bf16x4*)(a.out_bf16 + ob + o) = h;
                }
                if (a.out_f16) {
                    f16x4 h;
                    h[0] = (f16_t)v[0]; h[1] = (f16_t)v[1]; h[2] = (f16_t)v[2]; h[3] = (f16_t)v[3];
                    *(f16x4*)(a.out_f16 + ob + o) = h;
                }
            }
            acc[i][j] = v;
        }
        if (a.row_part) {
            // the four fq lanes of a row hold this wave's other columns of it
            const float o1 = __shfl_xor(racc, 16);
            racc = a.row_mode == 1 ? fmaxf(racc, o1) : racc + o1;
            const float o2 = __shfl_xor(racc, 32);
            racc = a.row_mode == 1 ? fmaxf(racc, o2) : racc + o2;
            const int slots = ctiles * WC;
            if (fq == 0) a.row_part[((long long)b * slots + (c0 / BC) * WC + wc) * a.row_bs + p] = racc;
        }
        if constexpr (TC == 4) {
            if (!staged && wide16 && cg0 + 16 <= a.Cout && a.out_mode == 0) {
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                const long long o = ob + (long long)p * a.ldo + cg0;
#pragma unroll
                for (int i = 0; i < TC; i += 2) {
                    if (a.out_bf16) {
                        bf16x8 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { h[r] = (bf16_t)acc[i][j][r]; h[4 + r] = (bf16_t)acc[i + 1][j][r]; }
                        *(bf16x8*)(a.out_bf16 + o + 4 * i) = h;
                    }
                    if (a.out_f16) {
                        f16x8 h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j][r]; h[4 + r] = (f16_t)acc[i + 1][j][r]; }
                        *(f16x8*)(a.out_f16 + o + 4 * i) = h;
                    }
                }
            }
        }
    }
    if constexpr (TC == 4) {
        if (staged) {
            // In the accumulator layout neighbouring lanes are neighbouring ROWS of the output (ldo elements apart): a direct
            // store touches 64 lines with 16 B each.  Epilogue-heavy launches (Q.K^T: 128 KB of scores per 8-K-step tile)
            // are bound by that, so 16-bit outputs are transposed through the (now idle) stage buffers: per wave
            // [row][64 couts] rows of 128 B + 16 B pad, re-read as 8 lanes x 16 B per row -> 8 full lines per store.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            constexpr int RS = 144, RP = TP * 16 > 64 ? 64 : TP * 16, JP = RP / 16;   // rows per pass (LDS budget), row tiles per pass
            static_assert(8 * RP * RS <= 2 * (BP + BC) * ROWB && (TP % JP) == 0, "staging fits the stage buffers");
            __syncthreads();                                 // every wave has consumed its last fragments
            char* R = smem + wave * (RP * RS);
            const int cl = lane & 7, pl = lane >> 3;
            const long long cb = ob + c0 + wc * (BC / WC) + cl * 8;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                char* outp = pass == 0 ? (char*)a.out_f16 : (char*)a.out_bf16;
                if (!outp) continue;
#pragma unroll
                for (int j0 = 0; j0 < TP; j0 += JP) {
#pragma unroll
                    for (int jj = 0; jj < JP; ++jj)
#pragma unroll
                        for (int i = 0; i < TC; i += 2) {
                            char* d = R + (jj * 16 + frow) * RS + fq * 32 + i * 8;
                            if (pass == 0) {
                                f16x8 h;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j0 + jj][r]; h[4 + r] = (f16_t)acc[i + 1][j0 + jj][r]; }
                                *(f16x8*)d = h;
                            } else {
                                bf16x8 h;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { h[r] = (bf16_t)acc[i][j0 + jj][r]; h[4 + r] = (bf16_t)acc[i + 1][j0 + jj][r]; }
                                *(bf16x8*)d = h;
                            }
                        }
                    asm volatile("" ::: "memory");           // (LDS executes one wave's accesses in order)
#pragma unroll
                    for (int k = 0; k < RP / 8; ++k) {
                        const int row = k * 8 + pl;
                        const int p = p0 + wp * (BP / WP) + j0 * 16 + row;
                        const u32x4 v = *(const u32x4*)(R + row * RS + cl * 16);
                        if (p < HWo) *(u32x4*)(outp + (cb + (long long)p * a.ldo) * 2) = v;
                    }
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    if (a.gn_partial) {
        // GroupNorm statistics of this tile's outputs (requires Cout % BC == 0: every cout column is real)
        __syncthreads();
        const int G = a.Cout / a.gn_cpg;
        float* out = a.gn_partial + (((long long)b * ptiles + (p0 / BP)) * G + c0 / a.gn_cpg) * 3;
        if constexpr (TC == 4) vt_gn_epilogue_partials_il<TC, TP>(acc, valid, a.gn_cpg, wp, WP, wc * (BC / WC), BC, (float*)smem, out);
        else vt_gn_epilogue_partials<TC, TP>(acc, valid, a.gn_cpg, wp, WP, wc * (BC / WC), BC, (float*)smem, out);
    }
}

template <int BP, int BC, int WP, int WC, bool OCC2 = false, bool F8 = false, bool F16 = false>
__global__ __launch_bounds__(512, OCC2 ? 4 : 2) void conv_gemm_kernel(const ConvGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_gemm_tile<BP, BC, WP, WC, F8, F16>(a, smem, vt_xcd_remap(blockIdx.x, gridDim.x));
}

// Gated launches (a.gate: usually a no-op decided on the device) run as a small resident grid that walks the tiles, so a
// launch that turns out to be a no-op costs 512 workgroup dispatches instead of one per tile (0.6 ms for Q.K^T at 16 x 1024^2).
template <int BP, int BC, int WP, int WC, bool OCC2 = false>
__global__ __launch_bounds__(512, OCC2 ? 4 : 2) void conv_gemm_gated_kernel(const ConvGemmArgs a, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (*a.gate != a.gate_expect) return;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        conv_gemm_tile<BP, BC, WP, WC>(a, smem, t);
        __syncthreads();                                   // the epilogue may still be reading the stage buffers
    }
}

// fp8 launches (a.f8) and fp16-operand launches (a.f16): plain grid only (no gated variant)
template <int BP, int BC, int WP, int WC, bool OCC2 = false, bool F16 = false>
hipError_t launch_cfg_f8(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (BP + BC) * ROWB;
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = conv_gemm_kernel<BP, BC, WP, WC, OCC2, !F16, F16>;
    hipError_t ea = vt_once_per_device(attr_done, [&] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); });
    if (ea != hipSuccess) return ea;
    const int HWo = a.Hout * a.Wout;
    const long long nblk = (long long)((HWo + BP - 1) / BP) * ((a.Cout + BC - 1) / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), smem, s, a);
    return hipGetLastError();
}

template <int BP, int BC, int WP, int WC, bool OCC2 = false>
hipError_t launch_cfg(const ConvGemmArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (BP + BC) * ROWB;
    static_assert(!OCC2 || smem <= 80 * 1024, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = conv_gemm_kernel<BP, BC, WP, WC, OCC2>;
    auto gated = conv_gemm_gated_kernel<BP, BC, WP, WC, OCC2>;
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gated, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        return e;
    });
    if (ea != hipSuccess) return ea;
    const int HWo = a.Hout * a.Wout;
    const int ptiles = (HWo + BP - 1) / BP, ctiles = (a.Cout + BC - 1) / BC;
    const long long nblk = (long long)ptiles * ctiles * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.gate) hipLaunchKernelGGL(gated, dim3((unsigned)(nblk < 512 ? nblk : 512)), dim3(512), smem, s, a, (int)nblk);
    else hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), smem, s, a);
    return hipGetLastError();
}

}  // namespace

// Host-side shape validation lives here so a bad call can never reach the DMA address math.
hipError_t vt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t s) {
    if (!a.X || !a.W || !a.zeros) return hipErrorInvalidValue;
    if (a.ksize != 1 && a.ksize != 3) return hipErrorInvalidValue;
    if (a.Cin <= 0 || a.Cout <= 0 || a.batch <= 0 || a.Wrows <= 0 || a.Wrows > a.Cout) return hipErrorInvalidValue;
    const int ce = a.f8 ? 16 : 8;                                          // elements per 16-B chunk
    if ((a.ldx % ce) || (a.ldw % ce)) return hipErrorInvalidValue;        // 16-B aligned rows for the DMA
    if (a.f8 && (a.gate || a.row_mode || a.x_stream || (a.Cin % 16) || (a.x_bs % 16) || (a.w_bs % 16))) return hipErrorInvalidValue;
    if (a.out_mode == 0 && ((a.ldo % 4) || (a.Cout % 4))) return hipErrorInvalidValue;
    if ((a.res || a.res_f16) && (a.ldr % 4)) return hipErrorInvalidValue;
    if (a.res && a.res_f16) return hipErrorInvalidValue;
    if (a.Cin % 8) return hipErrorInvalidValue;                           // k tail handled per 8-element chunk
    if (a.gn_partial) {
        if (a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
        const int bc = a.Cout <= 32 ? 32 : (a.Cout <= 128 ? 128 : 256);
        if (a.out_mode != 0 || (a.Cout % bc) || a.Cout <= 32) return hipErrorInvalidValue;
    }
    if (a.row_mode < 0 || a.row_mode > 3 || (a.row_mode != 0 && (a.out_mode != 0 || a.gn_partial || a.row_bs < (long long)a.Hout * a.Wout)))
        return hipErrorInvalidValue;
    if ((a.row_mode == 2 || a.row_mode == 3) != (a.row_in != nullptr)) return hipErrorInvalidValue;
    if ((a.row_mode == 1 || a.row_mode == 2) != (a.row_part != nullptr)) return hipErrorInvalidValue;
    if (a.row_mode == 1 && (a.out_f32 || a.out_bf16 || a.out_f16)) return hipErrorInvalidValue;
    // per-image offsets are 32-bit
    if ((long long)a.Hin * a.Win * a.ldx >= (1LL << 31)) return hipErrorInvalidValue;
    if ((long long)a.Wrows * a.ldw >= (1LL << 31)) return hipErrorInvalidValue;
    if (a.f16) {
        if (a.f8 || a.gate || a.row_mode || vt_conv_gemm_config(a) != 0) return hipErrorInvalidValue;      // only conv_out's tile is instantiated
        return launch_cfg_f8<128, 32, 8, 1, false, true>(a, s);
    }
    if (a.f8) {
        switch (vt_conv_gemm_config(a)) {
            case 9: return launch_cfg_f8<192, 128, 4, 2, true>(a, s);
            case 2: return launch_cfg_f8<256, 256, 2, 4>(a, s);
            default: return hipErrorInvalidValue;                          // only the tiles the stride-2 convs use are instantiated
        }
    }
    switch (vt_conv_gemm_config(a)) {
        case 0: return launch_cfg<128, 32, 8, 1>(a, s);
        case 1: return launch_cfg<256, 128, 4, 2>(a, s);
        case 9: return launch_cfg<192, 128, 4, 2, true>(a, s);
        default: return launch_cfg<256, 256, 2, 4>(a, s);
    }
}

// upper bound of the pixel tiles (= GroupNorm partials per image) any configuration uses for this Cout: buffer sizing
int vt_conv_gemm_ptiles(int HWo, int Cout) {
    const int bp = Cout <= 32 ? 128 : 192;
    return (HWo + bp - 1) / bp;
}
// the pixel tiles of THIS launch
int vt_conv_gemm_ptiles_of(const ConvGemmArgs& a) {
    const int cfg = vt_conv_gemm_config(a);
    const int bp = cfg == 0 ? 128 : (cfg == 9 ? 192 : 256);
    return (a.Hout * a.Wout + bp - 1) / bp;
}

// (row, column slot) partials per row a row_mode 1 / 2 launch writes
int vt_conv_gemm_col_slots(const ConvGemmArgs& a) {
    const int cfg = vt_conv_gemm_config(a);
    const int bc = cfg == 0 ? 32 : (cfg == 2 ? 256 : 128), wc = cfg == 0 ? 1 : (cfg == 2 ? 4 : 2);
    return (a.Cout + bc - 1) / bc * wc;
}

int vt_conv_gemm_config(const ConvGemmArgs& a) {
    if (a.Cout <= 32) return 0;
    // the 128-cout stride-2 conv (18 K-steps per tile) gains the same way from a second resident workgroup
    if (a.Cout == 128 && a.short_tiles && a.ksize == 3 && a.out_mode == 0) return 9;
    if (a.Cout <= 128) return 1;
    // 1x1 / GEMM launches with K <= 512 spend most of a 256x256 tile's life in its prologue and epilogue
    if (a.short_tiles && a.ksize == 1 && a.Cin <= 512 && (a.Cout % 128) == 0 && a.out_mode == 0) return 9;
    return 2;
}
const char* vt_conv_gemm_config_name(int cfg) {
    static const char* n[VT_NUM_PROF_SLOTS] = {"conv_gemm_kernel<128,32,8,1>", "conv_gemm_kernel<256,128,4,2>",
                                               "conv_gemm_kernel<256,256,2,4>", "conv3x3_halo_kernel<2,2,0,8,4>",
                                               "conv3x3_halo_kernel<4,2,0,4,4>", "conv3x3_halo_kernel<4,2,0,8,6>",
                                               "conv3x3_halo_kernel<2,4,0,8,6>", "conv3x3_halo_kernel<.,.,1,8,6>",
                                               "conv3x3_halo_kernel<.,.,2,8,6>", "conv_gemm_kernel<192,128,4,2,occ2>", "attn_qk_kernel", "conv3x3_halo_fp8_kernel", "attn_pv_kernel", "conv_gemm_fp8_kernel", "conv3x3_s2_halo_kernel", "attn_qk_fp8_kernel", "attn_pv_fp8_kernel", "conv3x3_s2_halo_fp8_kernel", "proj_fp8_kernel", "conv3x3_halo_fp8_kernel", "attn_qk_kernel<4> (projections)", "conv_out_halo_kernel", "gn_apply_kernel"};
    return (cfg >= 0 && cfg < VT_NUM_PROF_SLOTS) ? n[cfg] : "?";
}
