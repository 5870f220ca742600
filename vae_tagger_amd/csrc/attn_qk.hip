// Q.K^T of the VAE mid-block attention (1 head, d = 512) with the softmax numerators in the epilogue.
//
// The generic GEMM streams BOTH operands through LDS on every K-step and, at K = 512, spends most of a tile's life in
// its load -> compute -> epilogue chain (0.62 PF).  Here the roles are fixed instead:
//   * a workgroup owns 256 query rows for a whole sweep over the keys; each of its 8 waves keeps its 32 rows x 512 of Q in
//     REGISTERS (128 VGPRs: the MFMA B operands of all 16 k-steps), loaded once;
//   * keys stream through LDS in tiles of 64 rows x 512 (64 KB, two buffers, LDS-DMA, one barrier per tile); every wave
//     reads the whole tile as A operands: 16.8 MFLOP per 64 KB filled, 5x the generic tile's ratio;
//   * a wave owns complete rows of the score matrix, so row sums (or row maxima) accumulate in registers over the sweep:
//     no partial buffers, no reduction kernel;
//   * the epilogue of a key tile (one fma + v_exp_f32 per score, bf16 store of 32 B per lane and row) runs on the VALU
//     beside the other resident wave's MFMAs.
// mode 2: P[row][key] = exp(alpha * q.k - shift[row]) as bf16 (keys >= S: 0, up to ldp), rowval[row] = 1 / sum.
// mode 1: no P; rowval[row] = max_key alpha * q.k (the exact shift of run_attention's flagged path).
// mode 4 (round 4): the same skeleton as a linear layer of K = 512 (the attention's q | k and v^T projections): rows of one operand in
//         registers, the other operand's rows streamed as "keys", out = bf16(alpha * q.k + kbias[key] + qbias[row]) row-major.  The generic
//         GEMM ran these two launches at 0.5 PF (K = 512 is 8 of its K-steps: a tile lives in its prologue and epilogue).
// mode 5 (round 4): mode 4 with the attention's to_out epilogue: + residual stream, fp16 / fp32 stores, and GroupNorm (n, mean, M2) partials of the
//         result per (32-row slab of a wave, group of 16 output channels) for the norm that follows -- the last launch of the default path that
//         was still on the generic GEMM.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int QB = 256;        // query rows per workgroup (8 waves x 32)
constexpr int KT = 64;         // keys per LDS tile
constexpr int D = 512;         // head dim (= channels of the mid block)
constexpr int KROWB = D * 2;   // bytes per key row in LDS
constexpr int KBUF = KT * KROWB;

template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_qk_kernel(const AttnQkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 key tiles
    if (a.gate && *a.gate != a.gate_expect) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // an XCD gets a contiguous range of workgroups = the query tiles of one image (or a few): its L2 keeps that image's keys
    const int qtiles = (a.S + QB - 1) / QB;
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int per_img = qtiles * nsplit;
    const int b = logical / per_img, qs = logical - b * per_img;
    const int qt = qs / nsplit, ksp = qs - qt * nsplit;           // the splits of a query block are neighbours: they share its Q rows in L2
    const bf16_t* qb = a.q + (long long)b * a.qk_bs;
    constexpr bool LIN = MODE == 4 || MODE == 5;                  // linear-layer forms: separate operands, key tiles split evenly
    const bf16_t* kb = a.k + (long long)b * (LIN ? a.k_bs : a.qk_bs);
    const int nkeys = LIN ? a.nk : a.S;                           // rows of the streamed operand
    const int ldk = LIN ? a.ldk : a.ldq;
    const int row0 = qt * QB + wave * 32;

    // ---- this wave's Q slab: B operand of k-step ks for row tile j = q[row0 + 16 j + fr][32 ks + 8 fq .. +8]
    bf16x8 qf[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + j * 16 + fr;
        const bf16_t* src = row < a.S ? qb + (long long)row * a.ldq + fq * 8 : (const bf16_t*)a.zeros;
        const int step = row < a.S ? 32 : 0;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) qf[j][ks] = *(const bf16x8*)(src + ks * step);
    }
    float rv[2];                                        // per row tile: running sum (mode 2) / maximum (mode 1)
    float sh2[2];                                       // shift * log2(e)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + j * 16 + fr;
        rv[j] = MODE == 1 ? -__builtin_inff() : 0.f;
        if (LIN) sh2[j] = (a.qbias && row < a.S) ? a.qbias[row] : 0.f;                // (linear forms: the row's bias)
        else sh2[j] = (MODE >= 2 && row < a.S) ? a.rowin[(long long)b * a.row_bs + row] * 1.44269504f : 0.f;
    }
    const float alpha2 = a.alpha * 1.44269504f;

    // ---- key tile staging: one wave-instruction = one key row (1 KB); lane l writes physical 16-B chunk l and fetches
    // logical chunk l ^ (R & 15) (R = LDS row), so the 16 rows a fragment read touches hit 16 different chunk slots.
    // LDS row R = 16 i + 4 q + r holds key 32 (i >> 1) + 8 q + 4 (i & 1) + r of the tile: MFMA tiles 0, 1 of lane (fq, fr)
    // are keys 8 fq .. 8 fq + 7 and tiles 2, 3 keys 32 + 8 fq .. -- each 16-B store piece sits next to the other fq lanes'
    // pieces, so one store instruction writes a contiguous 64-B half line per row.
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int R = wave * 8 + jj;
            const int key = kt * KT + 32 * ((R >> 5) & 1) + 8 * ((R >> 2) & 3) + 4 * ((R >> 4) & 1) + (R & 3);
            const void* src = key < nkeys ? (const void*)(kb + (long long)key * ldk + ((lane ^ (R & 15)) << 3)) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(smem + buf * KBUF + R * KROWB), 16, 0, 0);
        }
    };
    // fragment of key tile i at k-step ks: LDS row 16 i + fr, logical chunk 4 ks + fq -> physical (4 ks + fq) ^ fr.
    // (4 ks + fq) ^ fr = ((ks & 3) ^ (fr >> 2)) << 2 | (fq ^ fr) & 3, plus (ks >> 2) << 4: four per-lane bases + immediates.
    int kbase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) kbase[m] = fr * KROWB + (((((m ^ (fr >> 2)) & 3) << 2) | ((fq ^ fr) & 3)) << 4);

    f32x4 acc[4][2];                                    // (every tile's first k-step starts from zero: no clearing)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // P pieces of a tile, packed by its epilogue and stored right behind it
    // (In the accumulator layout the four pieces of a row's 64-B half line sit in lanes 16 apart, and NEIGHBOURING lanes
    // hold different rows: stored as they stand, every lane's 16 B is its own memory request -- measured, the P write then
    // costs as much as the MFMAs.  The epilogue therefore moves piece (fr, fq) to lane 4 fr + fq with ds_bpermute, so each
    // quad of lanes writes 64 contiguous bytes.)
    bf16x8 hold[2][2];
    const int sr = lane >> 2, sp = lane & 3;                   // after the permute: this lane stores row sr, piece sp
    const int perm_addr = (sp * 16 + sr) << 2;                 // ... which it takes from lane 16 sp + sr
    auto store_held = [&](int kt_prev) __attribute__((always_inline)) {
        if constexpr (MODE == 5) return;                       // (stored by its epilogue, in accumulator layout)
        if constexpr (MODE == 3) {
            // fragment order (attn_pv.hip): piece (j, h) of this lane as it stands -- 1 KB contiguous per store instruction,
            // the wave's stream over the key tiles sequential in memory
            if (row0 < a.S) {
                bf16_t* dst = a.P + (long long)b * a.p_bs + (long long)(row0 >> 5) * vt_attn_pt_slab_stride(a.S) + (long long)kt_prev * 2048 + lane * 8;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
#ifdef PFRAG_PLAIN
                    for (int h = 0; h < 2; ++h) *(bf16x8*)(dst + (j * 2 + h) * 512) = hold[j][h];
#else
                    for (int h = 0; h < 2; ++h) __builtin_nontemporal_store(hold[j][h], (bf16x8*)(dst + (j * 2 + h) * 512));
#endif
            }
            return;
        }
        const int key0 = kt_prev * KT + 8 * sp;                // half h: keys key0 + 32 h .. + 7
        const bool full = kt_prev * KT + KT <= nkeys;
        if constexpr (MODE == 4) {                             // plain stores: the consumer (row norms / Q.K^T / P.V) reads the tensor next
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = row0 + j * 16 + sr;
                if (row < a.S) {
                    bf16_t* dst = a.P + (long long)b * a.p_bs + (long long)row * a.ldp + key0;
                    if (full || key0 < a.ldp) *(bf16x8*)dst = hold[j][0];
                    if (full || key0 + 32 < a.ldp) *(bf16x8*)(dst + 32) = hold[j][1];
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = row0 + j * 16 + sr;
            if (row < a.S) {
                bf16_t* dst = a.P + (long long)b * a.p_bs + (long long)row * a.ldp + key0;
                // streaming stores: 4+ GB of P per launch must not push the key tiles out of the XCD's L2
                if (full || key0 < a.ldp) __builtin_nontemporal_store(hold[j][0], (bf16x8*)dst);
                if (full || key0 + 32 < a.ldp) __builtin_nontemporal_store(hold[j][1], (bf16x8*)(dst + 32));
            }
        }
    };
    // scores of tile kt (in acc) -> running row value, packed P pieces
    auto epilogue_t = [&](int kt, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int key0 = kt * KT + 8 * fq;
        if constexpr (MODE == 5) {
            // lane (fq, fr): rows row0 + 16 j + fr, runs h = 0 / 1 = keys key0 + 32 h .. + 7 (tiles i = 2 h, 2 h + 1): 16-B fp16 (32-B fp32) pieces of
            // the residual and of the output; the GroupNorm group of run h is (kt * 64 + 32 h + 8 fq) / 16: shared by the lane pair fq ^ 1
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const int nrows = a.S - row0 < 0 ? 0 : (a.S - row0 > 32 ? 32 : a.S - row0);      // valid rows of this wave's slab (wave-uniform)
            const long long ob = (long long)b * a.p_bs;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kk = key0 + 32 * h;
                const bool kreal = FULL || kk + 7 < nkeys;       // (nk is a multiple of 16: a run is real or padding as a whole)
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                if (a.kbias && kreal) { b0 = *(const f32x4*)(a.kbias + kk); b1 = *(const f32x4*)(a.kbias + kk + 4); }
                float v[2][8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = row0 + j * 16 + fr;
                    const bool ok = row < a.S && kreal;
                    const long long o = ob + (long long)row * a.ldp + kk;
                    float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (ok && a.res_f16) {
                        const f16x8 rh = *(const f16x8*)(a.res_f16 + o);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rs[e] = (float)rh[e];
                    } else if (ok && a.res_f32) {
                        const f32x4 r0 = *(const f32x4*)(a.res_f32 + o), r1 = *(const f32x4*)(a.res_f32 + o + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { rs[e] = r0[e]; rs[4 + e] = r1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[j][e] = fmaf(acc[2 * h + (e >> 2)][j][e & 3], a.alpha, (e < 4 ? b0[e & 3] : b1[e & 3]) + sh2[j]) + rs[e];
                    if (ok && a.out_f16) {
                        f16x8 oh;
#pragma unroll
                        for (int e = 0; e < 8; ++e) oh[e] = (f16_t)v[j][e];
                        *(f16x8*)(a.out_f16 + o) = oh;
                    } else if (ok && a.out_f32) {
                        *(f32x4*)(a.out_f32 + o) = f32x4{v[j][0], v[j][1], v[j][2], v[j][3]};
                        *(f32x4*)(a.out_f32 + o + 4) = f32x4{v[j][4], v[j][5], v[j][6], v[j][7]};
                    }
                }
                if (a.gn_partial && kreal) {
                    // (n, mean, M2) of the slab's rows x the group's 16 keys, relative to a pivot (the group's first value of the slab's first row)
                    const float piv = __shfl(v[0][0], lane & 32, 64);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (row0 + j * 16 + fr < a.S) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - piv; s1 += d; s2 = fmaf(d, d, s2); }
                        }
                    }
                    s1 = vt_row16_sum(s1); s2 = vt_row16_sum(s2);
                    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                    if (fr == 0 && (fq & 1) == 0) {
                        const float n = (float)nrows * 16.f;
                        const float ms = n > 0.f ? s1 / n : 0.f;
                        float* d = a.gn_partial + (((long long)b * a.gn_parts + (row0 >> 5)) * (nkeys >> 4) + (kk >> 4)) * 3;
                        d[0] = n; d[1] = n > 0.f ? piv + ms : 0.f; d[2] = n > 0.f ? fmaxf(s2 - s1 * ms, 0.f) : 0.f;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool real = FULL || key0 + 32 * (i >> 1) + 4 * (i & 1) + r < nkeys;
                    if (MODE == 1) {
                        if (real) rv[j] = fmaxf(rv[j], acc[i][j][r] * a.alpha);
                    } else if (MODE == 4) {
                        const int kk = key0 + 32 * (i >> 1) + 4 * (i & 1) + r;
                        const float kbv = (a.kbias && real) ? a.kbias[kk] : 0.f;
                        hold[j][i >> 1][(i & 1) * 4 + r] = (bf16_t)(real ? fmaf(acc[i][j][r], a.alpha, kbv + sh2[j]) : 0.f);
                    } else {
                        float e = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], alpha2, -sh2[j]));
                        if (!real) e = 0.f;
                        rv[j] += e;
                        hold[j][i >> 1][(i & 1) * 4 + r] = (bf16_t)e;
                    }
                }
        }
        if (MODE == 2 || MODE == 4) {  /* row-major P: lane permute */
            typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    i32x4 v = __builtin_bit_cast(i32x4, hold[j][h]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) v[d] = __builtin_amdgcn_ds_bpermute(perm_addr, v[d]);
                    hold[j][h] = __builtin_bit_cast(bf16x8, v);
                }
        }
    };
    // MODE 3 (fragment-order P for attn_pv): the row sums leave as FOUR segment sums -- key tiles [i nkt / 4, (i + 1) nkt / 4) -- each
    // reduced over the row's four lanes and written when its last tile's epilogue has run; attn_pv adds them in the fixed order
    // ((s0 + s1) + s2) + s3 and inverts.  The association of the additions is then the same whether one workgroup sweeps all keys or
    // (small grids) 2 or 4 workgroups share a query block's sweep: an image's result does not depend on the batch it ran in, bit for bit.
    const int nkt_all = (nkeys + KT - 1) / KT;
    const int segb[5] = {0, nkt_all / 4, nkt_all / 2, (int)(3LL * nkt_all / 4), nkt_all};
    auto write_segment = [&](int seg, bool zero) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v = zero ? 0.f : rv[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int row = row0 + j * 16 + fr;
            if (fq == 0 && row < a.S) a.rowout[(long long)seg * a.split_stride + (long long)b * a.row_bs + row] = v;
            rv[j] = 0.f;
        }
    };
    auto epilogue = [&](int kt) __attribute__((always_inline)) {
        if (kt * KT + KT <= nkeys) epilogue_t(kt, std::true_type{});
        else epilogue_t(kt, std::false_type{});
        if (MODE == 3) {
#pragma unroll
            for (int seg = 0; seg < 4; ++seg)
                if (kt + 1 == segb[seg + 1] && segb[seg + 1] > segb[seg]) write_segment(seg, false);
        }
    };

    // Waves w and w + 4 share a SIMD.  The first four run [MFMAs of tile kt][DMA of kt + 1][epilogue + stores of kt] per barrier
    // interval, the other four [DMA of kt + 1][epilogue + stores of kt - 1][MFMAs of kt]: one wave's DMA issue (~100 cycles per
    // piece), exp / convert / store work runs beside its partner's MFMAs instead of both leaving the barrier into the matrix pipe
    // together and into the VALU together.
    // The wait in front of the barrier is COUNTED: vector-memory operations retire in issue order and a wave's P stores (4 per tile
    // in the fragment order) are always issued after the DMA pieces of the same interval, so s_waitcnt vmcnt(4) covers the key tile
    // and leaves the stores -- HBM writes -- a second interval to drain (vmcnt(0) parked the wave on them every tile).
    const bool late = (wave & 4) != 0;
    const int seg0 = ksp * (4 / nsplit), seg1 = (ksp + 1) * (4 / nsplit);      // nsplit = 1, 2 or 4: whole segments per workgroup
    // this workgroup's key tiles [kt0, nkt): whole segments in the fragment-order mode, an even share of the tiles in the linear mode
    const int kt0 = MODE == 3 ? segb[seg0] : LIN ? (int)((long long)ksp * nkt_all / nsplit) : 0;
    const int nkt = MODE == 3 ? segb[seg1] : LIN ? (int)((long long)(ksp + 1) * nkt_all / nsplit) : nkt_all;
    const bool counted = MODE == 3 && row0 < a.S;              // this wave issues exactly 4 stores per epilogue
    stage(kt0, kt0 & 1);
    for (int kt = kt0; kt < nkt; ++kt) {
        // stores issued during the previous interval: early waves after every tile, late waves from their second interval on
        if (counted && kt - kt0 > (late ? 1 : 0)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows of tile kt have landed ...
        __builtin_amdgcn_s_barrier();                          // ... everyone's; and everyone has READ tile kt - 1
        asm volatile("" ::: "memory");
        if (late) {
            if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
            if (kt > kt0) { epilogue(kt - 1); if (MODE >= 2) store_held(kt - 1); }
        }
        const char* ks_base = smem + (kt & 1) * KBUF;
        // key fragments through a ring of eight register sets, read AHEAD fragments before the two MFMAs that use them: left to the
        // compiler (246 VGPRs) every ds_read_b128 reused one register set directly in front of its MFMAs behind an
        // s_waitcnt lgkmcnt(0) -- one exposed LDS round trip per 32 matrix-pipe cycles (51 % MFMA busy, half the wave cycles parked)
        constexpr int AHEAD = 6;
        bf16x8 kf[8];
        auto frag = [&](int idx) __attribute__((always_inline)) {
            const int ks = idx >> 2, i = idx & 3;
            return *(const bf16x8*)(ks_base + kbase[ks & 3] + (ks >> 2) * 256 + i * 16 * KROWB);
        };
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) kf[p] = frag(p);
#pragma unroll
        for (int idx = 0; idx < 64; ++idx) {
            if (idx + AHEAD < 64) kf[(idx + AHEAD) & 7] = frag(idx + AHEAD);
            const int ks = idx >> 2, i = idx & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[idx & 7], qf[j][ks], ks == 0 ? zero4 : acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                 // keep the read-ahead distance: the scheduler would sink the reads to their uses
        }
        if (!late) {
            if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
            epilogue(kt);
            if (MODE >= 2) store_held(kt);
        }
    }
    if (nkt > kt0 && late) {
        epilogue(nkt - 1);
        if (MODE >= 2) store_held(nkt - 1);
    }
    if constexpr (LIN) return;
    if (MODE == 3) {
        // empty segments (fewer than four key tiles) still have a defined sum
#pragma unroll
        for (int seg = 0; seg < 4; ++seg)
            if (seg >= seg0 && seg < seg1 && segb[seg + 1] == segb[seg]) write_segment(seg, true);
        return;
    }
    // ---- the four fq lanes of a row hold its other keys
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v = rv[j];
        const float o1 = __shfl_xor(v, 16);
        v = MODE == 1 ? fmaxf(v, o1) : v + o1;
        const float o2 = __shfl_xor(v, 32);
        v = MODE == 1 ? fmaxf(v, o2) : v + o2;
        const int row = row0 + j * 16 + fr;
        if (fq == 0 && row < a.S) a.rowout[(long long)b * a.row_bs + row] = MODE == 1 ? v : 1.f / v;
    }
}

}  // namespace

bool vt_attn_qk_supported(int S, int C) { return C == D && S > 0; }

int vt_attn_linear_parts(int S) { return (S + QB - 1) / QB * 8; }      // 32-row slabs launched per image (mode 5's GroupNorm partials)

hipError_t vt_launch_attn_qk(const AttnQkArgs& a, hipStream_t s) {
    if (a.mode == 5) {
        if (!a.q || !a.k || !a.zeros || a.batch <= 0 || !vt_attn_qk_supported(a.S, a.C) || a.nk <= 0 || (a.nk % 16) || a.gate) return hipErrorInvalidValue;
        if ((a.out_f16 != nullptr) == (a.out_f32 != nullptr) || (a.res_f16 && a.res_f32)) return hipErrorInvalidValue;
        if ((a.ldq % 8) || (a.ldk % 8) || (a.qk_bs % 8) || (a.k_bs % 8) || (a.ldp % 8) || (a.p_bs % 8) || a.ldp < a.nk) return hipErrorInvalidValue;
        if ((long long)a.S * a.ldq >= (1LL << 31) || (long long)a.nk * a.ldk >= (1LL << 31)) return hipErrorInvalidValue;
        if (a.gn_partial && a.gn_parts != vt_attn_linear_parts(a.S)) return hipErrorInvalidValue;
        const int nsp = a.nsplit > 1 ? a.nsplit : 1;
        if (nsp > (a.nk + KT - 1) / KT) return hipErrorInvalidValue;
        const long long nblk5 = (long long)((a.S + QB - 1) / QB) * a.batch * nsp;
        if (nblk5 > 0x7fffffffLL) return hipErrorInvalidValue;
        static std::atomic<unsigned long long> attr5{0};
        hipError_t e5 = vt_once_per_device(attr5, [&] { return hipFuncSetAttribute((const void*)attn_qk_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF); });
        if (e5 != hipSuccess) return e5;
        hipLaunchKernelGGL(attn_qk_kernel<5>, dim3((unsigned)nblk5), dim3(512), 2 * KBUF, s, a);
        return hipGetLastError();
    }
    if (a.mode == 4) {
        if (!a.q || !a.k || !a.P || !a.zeros || a.batch <= 0 || !vt_attn_qk_supported(a.S, a.C) || a.nk <= 0 || a.gate) return hipErrorInvalidValue;
        if ((a.ldq % 8) || (a.ldk % 8) || (a.qk_bs % 8) || (a.k_bs % 8) || (a.ldp % 8) || (a.p_bs % 8) || a.ldp < (a.nk + 7) / 8 * 8) return hipErrorInvalidValue;
        if ((long long)a.S * a.ldq >= (1LL << 31) || (long long)a.nk * a.ldk >= (1LL << 31)) return hipErrorInvalidValue;
        const int nsp = a.nsplit > 1 ? a.nsplit : 1;
        if (nsp > (a.nk + KT - 1) / KT) return hipErrorInvalidValue;
        const long long nblk4 = (long long)((a.S + QB - 1) / QB) * a.batch * nsp;
        if (nblk4 > 0x7fffffffLL) return hipErrorInvalidValue;
        static std::atomic<unsigned long long> attr4{0};
        hipError_t e4 = vt_once_per_device(attr4, [&] { return hipFuncSetAttribute((const void*)attn_qk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF); });
        if (e4 != hipSuccess) return e4;
        hipLaunchKernelGGL(attn_qk_kernel<4>, dim3((unsigned)nblk4), dim3(512), 2 * KBUF, s, a);
        return hipGetLastError();
    }
    if (!a.q || !a.k || !a.rowout || !a.zeros || a.batch <= 0 || !vt_attn_qk_supported(a.S, a.C)) return hipErrorInvalidValue;
    if (a.mode != 1 && a.mode != 2) return hipErrorInvalidValue;
    if (a.mode == 2 && (!a.P || !a.rowin || (a.p_bs % 8))) return hipErrorInvalidValue;
    if (a.mode == 2 && !a.p_frag && ((a.ldp % 8) || a.ldp < a.S)) return hipErrorInvalidValue;
    if (a.mode == 2 && a.p_frag && a.p_bs < vt_attn_pt_elems(a.S)) return hipErrorInvalidValue;
    if ((a.ldq % 8) || (a.qk_bs % 8) || a.row_bs < a.S) return hipErrorInvalidValue;
    if ((long long)a.S * a.ldq >= (1LL << 31)) return hipErrorInvalidValue;
    if (a.nsplit > 1 && (a.mode != 2 || !a.p_frag || (a.nsplit != 2 && a.nsplit != 4))) return hipErrorInvalidValue;
    if (a.mode == 2 && a.p_frag && a.split_stride < (long long)a.batch * a.row_bs) return hipErrorInvalidValue;      // rowout = [4 segments][split_stride]
    const long long nblk = (long long)((a.S + QB - 1) / QB) * a.batch * (a.nsplit > 1 ? a.nsplit : 1);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)attn_qk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_qk_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_qk_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
        return e;
    });
    if (ea != hipSuccess) return ea;
    if (a.mode == 1) hipLaunchKernelGGL(attn_qk_kernel<1>, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    else if (a.p_frag) hipLaunchKernelGGL(attn_qk_kernel<3>, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    else hipLaunchKernelGGL(attn_qk_kernel<2>, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    return hipGetLastError();
}
