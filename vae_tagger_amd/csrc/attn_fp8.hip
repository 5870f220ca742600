// The mid-block attention's two S x S contractions on fp8 (OCP e4m3) operands -- the rest of BASELINE.json configs[4]
// (vt_set_flag(ctx, 11, 1); diffusers Attention behind /root/reference/diffusers_vae_loader.py:79, 1 head, d = 512).
// Same division of labour as attn_qk.hip / attn_pv.hip (Q slab in registers, keys streamed through LDS, a wave owns whole rows of
// the score matrix; P stored in MFMA fragment order and loaded straight into registers by P.V, only v^T through LDS), re-shaped for
// v_mfma_scale_f32_16x16x128_f8f6f4 (both scales 2^0): four times the K per instruction at twice the cycles.
//   * operands: q8 | k8 = e4m3(8 q | 8 k) (misc_kernels.hip), v8^T = e4m3(8 v^T); P8 = e4m3(256 exp(s - max_row s)): the exponent shift
//     is the EXACT row maximum of the e4m3 scores, from a first sweep of this kernel (MODE 1, no exp / convert / store);
//   * a lane of the 16x16x128 MFMA holds 32 consecutive k-bytes of its row (two ds_read_b128 / two 16-B global loads);
//   * Q.K^T key tile = 128 keys x 512 B = 64 KB (two buffers); LDS row R = 16 m + r of the tile holds key 32 (r >> 2) + 16 (m >> 2) +
//     4 (m & 3) + (r & 3) of the 128-key block: the accumulators of lane (q4, .) over the tile's eight MFMA row tiles are then 32
//     CONSECUTIVE keys 32 q4 .. 32 q4 + 31 -- exactly its B-operand fragment of P.V's k-step over that block, and v^T needs no
//     permutation at all;
//   * the exponent shift may be a SAMPLED row maximum (every 8th key tile: 0.2 ms instead of 1.5 ms per step); the numerator sweep raises a
//     flag when a numerator exceeds e4m3's 448 and the launches gated on that flag redo the group with the exact maximum;
//   * P8 of (32-query slab, 128-key block) = four 1-KB pieces (query tile qb, key half): 4 KB contiguous per wave and block, 2304 B
//     between slabs against HBM channel conflicts (as attn_qk.hip); P traffic halves against bf16 (4.3 -> 2.1 GB per 8 images);
//   * P.V: v8^T tile = 256 channels x 128 keys = 32 KB (the bf16 kernel's bytes, twice its keys); its two ds_read_b128 per fragment
//     take the chunk pair in an order that depends on the lane quarter (conflict-free under the 8-chunk XOR swizzle), and
//     Q.K^T stores P8's two key halves of a lane in that same order -- the MFMA only needs A and B to agree.
// Row sums: of the e4m3-ROUNDED numerators, by the matrix pipe: P.V multiplies P8 with one more row of v^T that holds 1.0 everywhere
// (an A operand of constants, nothing read), so that a row's weights sum to one exactly and the Q.K^T epilogue carries no sum.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int QB = 256;        // query rows per workgroup (8 waves x 32)
constexpr int KT = 128;        // keys per LDS tile = one k-step of P.V
constexpr int D = 512;         // head dim
constexpr int KROWB = D;       // bytes per key row in LDS (e4m3)
constexpr int KBUF = KT * KROWB;    // 64 KB

__device__ __forceinline__ int opaque8(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ i32x8 cat8(i32x4 lo, i32x4 hi) { return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// ---------------------------------------------------------------------------------------------------------------------------
// Q.K^T: P8 fragments + segment sums
// MODE 1: no P; rowout[row] = max_key alpha * q8.k8 (the exact exponent shift: e4m3's range, 2^-6 .. 448 for normal numbers, is too
// short for a shift bounded from operand norms -- a row whose maximum sits e^-5 below the bound would keep one significant bit);
// MODE 3: P8 = e4m3(pscale * exp(alpha * q8.k8 - rowin[row])) in fragment order + segment sums of the rounded values.
template <int MODE>
__global__ __launch_bounds__(512, 2) void attn_qk_fp8_kernel(const AttnQk8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 key tiles
    if (a.gate && *a.gate != a.gate_expect) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int qtiles = (a.S + QB - 1) / QB;
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int per_img = qtiles * nsplit;
    const int b = logical / per_img, qs = logical - b * per_img;
    const int qt = qs / nsplit, ksp = qs - qt * nsplit;
    const unsigned char* qb = a.qk8 + (long long)b * a.qk_bs;
    const unsigned char* kb = qb + D;
    const int row0 = qt * QB + wave * 32;

    // ---- this wave's Q slab: B operand of k-step ks for query tile j = q8[row0 + 16 j + fr][128 ks + 32 fq .. + 32]
    i32x8 qf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + j * 16 + fr;
        const unsigned char* src = row < a.S ? qb + (long long)row * a.ldq + fq * 32 : (const unsigned char*)a.zeros;
        const int step = row < a.S ? 128 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[j][ks] = cat8(*(const i32x4*)(src + ks * step), *(const i32x4*)(src + ks * step + (row < a.S ? 16 : 0)));
    }
    float rv[2], sh2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + j * 16 + fr;
        rv[j] = MODE == 1 ? -__builtin_inff() : 0.f;
        // (pscale = 2^k rides in the exponent: exp2(x - (shift log2 e - k)))
        sh2[j] = (MODE == 3 && row < a.S) ? a.rowin[(long long)b * a.row_bs + row] * 1.44269504f - a.pscale_log2 : 0.f;
    }
    const float alpha2 = a.alpha * 1.44269504f;

    // ---- key tile staging: one wave-instruction = 2 key rows (1 KB); lane l -> row (l >> 5), physical 16-B chunk (l & 31), which holds
    // logical chunk (l & 31) ^ (R & 15) (R = LDS row): the 16 rows a fragment read touches hit 16 different chunk slots.
    // With R = 2 (8 jj + wave) + (lane >> 5): row tile m = jj and r = 2 wave + (lane >> 5) for every jj, so a lane's eight source
    // addresses of a tile differ by wave-uniform offsets only (kept that way on purpose: eight hoisted 64-bit per-lane addresses spill).
    const int sr_ = 2 * wave + (lane >> 5);                               // r of this lane's rows
    const int skey = 32 * (sr_ >> 2) + (sr_ & 3);                         // ... their key inside the block, before the row tile's 16 (m >> 2) + 4 (m & 3)
    const int schunk = ((lane & 31) ^ sr_) << 4;
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        const int key_l = kt * KT + opaque8(skey);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int key = key_l + 16 * (jj >> 2) + 4 * (jj & 3);
            const void* src = key < a.S ? (const void*)(kb + (long long)key * a.ldq + schunk) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(smem + buf * KBUF + (jj * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    // fragment of row tile m at k-step ks: LDS row 16 m + fr, logical chunks 8 ks + 2 fq (+ 1) -> physical ^ fr (low four bits)
    int kbase[2][2];                                       // [ks & 1][half]: the XOR touches chunk bits 0..3, ks >> 1 is an immediate
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) kbase[p][h] = fr * KROWB + ((((8 * p + 2 * fq + h) ^ fr) & 15) << 4);

    // A 128-key tile is worked off in two HALVES of four row tiles (64 keys x 32 queries = 32 accumulator registers): with all eight row
    // tiles in flight the kernel sat at 255 VGPRs and any further state -- the overflow watch below -- spilled ~100 of them.
    f32x4 acc[4][2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    i32x4 hold[2][2];                                      // P8 pieces of the tile: [query tile][key half]
    unsigned clampw = 0;                                   // MODE 3: bit 7 of a byte set <=> some stored P8 byte is 448 (the shift may be a SAMPLED maximum)

    auto store_held = [&](int kt_prev) __attribute__((always_inline)) {
        if (row0 < a.S) {
            unsigned char* dst = a.P8 + (long long)b * a.p_bs + (long long)(row0 >> 5) * vt_attn_p8_slab_stride(a.S) + (long long)kt_prev * 4096 + lane * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)      // P.V reads the key half (h ^ (fq & 1)) first: see the header
                    __builtin_nontemporal_store(hold[j][h], (i32x4*)(dst + (j * 2 + (h ^ (fq & 1))) * 1024));
        }
    };
    // scores of half h of tile kt (in acc) -> running row value / P8 piece hold[.][h]
    auto epilogue_half = [&](int kt, int h /* compile-time at every call */, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int key0 = kt * KT + 32 * fq + 16 * h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (FULL || key0 + 4 * mm + r < a.S) rv[j] = fmaxf(rv[j], acc[mm][j][r] * a.alpha);
                    continue;
                }
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool real = FULL || key0 + 4 * mm + r < a.S;
                    float v = __builtin_amdgcn_exp2f(fmaf(acc[mm][j][r], alpha2, -sh2[j]));
                    if (!real) v = 0.f;
                    e[r] = fminf(v, 448.f);
                }
                // (no row sum here: this epilogue is what bounds the kernel -- ~20 VALU cycles per score against 16 matrix-pipe cycles -- and
                // P.V gets the sum of the ROUNDED numerators for two extra MFMAs per tile, from an all-ones row of v^T)
                int w = hold[j][h][mm];
                w = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w, true);
                // a byte of 0x7e = 448: a numerator reached the clamp (values of 432 .. 448 round there too: a rare false alarm costs one
                // exact redo).  All bytes are <= 0x7e, so +2 per byte carries nothing into its neighbour.
                clampw |= ((unsigned)w + 0x02020202u) & 0x80808080u;
                hold[j][h][mm] = w;
            }
        }
    };
    const int nkt_all = (a.S + KT - 1) / KT;
    const int segb[5] = {0, nkt_all / 4, nkt_all / 2, (int)(3LL * nkt_all / 4), nkt_all};
    auto epilogue = [&](int kt, int h) __attribute__((always_inline)) {
        if (kt * KT + KT <= a.S) epilogue_half(kt, h, std::true_type{});
        else epilogue_half(kt, h, std::false_type{});
    };

    // Waves w and w + 4 share a SIMD.  Per barrier interval the first four run [MFMAs h0][epilogue h0][MFMAs h1][DMA of the next tile]
    // [epilogue h1 + stores], the other four [DMA][epilogue h1 + stores of the PREVIOUS tile][MFMAs h0][epilogue h0][MFMAs h1] -- their
    // second half's accumulators wait across the barrier -- so that one wave's DMA issue, exp / convert / store work runs beside its
    // partner's MFMAs (attn_qk.hip), now in four alternating phases instead of two.
    // The wait in front of the barrier is COUNTED (vmcnt(4): the wave's four P8 stores are always younger than its DMA pieces).
    const bool late = (wave & 4) != 0;
    const int seg0 = ksp * (4 / nsplit), seg1 = (ksp + 1) * (4 / nsplit);
    const int kt0 = MODE == 3 ? segb[seg0] : 0, nkt = MODE == 3 ? segb[seg1] : nkt_all;
    const bool counted = MODE == 3 && row0 < a.S;          // this wave issues exactly 4 stores per tile
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) hold[j][h] = i32x4{0, 0, 0, 0};
    // MODE 1 may sweep every kstep-th key tile only (a sampled maximum: see run_attention); the two LDS buffers alternate per iteration
    const int kstep = (MODE == 1 && a.kstride > 1) ? a.kstride : 1;
    int it = 0;
    if (kt0 < nkt) stage(kt0, 0);
    for (int kt = kt0; kt < nkt; kt += kstep, ++it) {
        if (counted && it > (late ? 1 : 0)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (late) {
            if (kt + kstep < nkt) stage(kt + kstep, (it + 1) & 1);
            if (it > 0) { epilogue(kt - kstep, 1); if (MODE == 3) store_held(kt - kstep); }
        }
        const char* ks_base = smem + (it & 1) * KBUF;
        // key fragments (32 B per lane = two ds_read_b128) through a ring of register sets, read AHEAD positions before their MFMAs
        constexpr int AHEAD = 3, RING = 4;
        auto half = [&](int h /* compile-time */) __attribute__((always_inline)) {
            i32x8 kf[RING];
            auto frag = [&](int idx) __attribute__((always_inline)) -> i32x8 {
                const int ks = idx >> 2, m = 4 * h + (idx & 3);
                const char* p = ks_base + (ks >> 1) * 256 + m * 16 * KROWB;
                return cat8(*(const i32x4*)(p + kbase[ks & 1][0]), *(const i32x4*)(p + kbase[ks & 1][1]));
            };
#pragma unroll
            for (int p = 0; p < AHEAD; ++p) kf[p] = frag(p);
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                if (idx + AHEAD < 16) kf[(idx + AHEAD) % RING] = frag(idx + AHEAD);
                const int ks = idx >> 2, mm = idx & 3;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mm][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kf[idx % RING], qf[j][ks], ks == 0 ? zero4 : acc[mm][j], 0, 0, 0, 127, 0, 127);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        half(0);
        epilogue(kt, 0);
        half(1);
        if (!late) {
            if (kt + kstep < nkt) stage(kt + kstep, (it + 1) & 1);
            epilogue(kt, 1);
            if (MODE == 3) store_held(kt);
        }
    }
    if (nkt > kt0 && late) {
        const int kl = kt0 + ((nkt - 1 - kt0) / kstep) * kstep;       // the last tile swept
        epilogue(kl, 1);
        if (MODE == 3) store_held(kl);
    }
    if (MODE == 3 && a.flag && __any(clampw != 0u) && lane == 0) atomicOr(a.flag, 1);
    if constexpr (MODE == 3) return;
    // MODE 1: the four fq lanes of a row hold its other keys
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v = rv[j];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        const int row = row0 + j * 16 + fr;
        if (fq == 0 && row < a.S) a.rowout[(long long)b * a.row_bs + row] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Linear projections of the fp8 mode on the same skeleton (rows of one operand resident in registers as a 32-row slab per wave, the other
// operand's rows streamed through LDS as 128-row tiles under the key -> LDS-row map above, so that a lane's accumulators over a tile are 32
// CONSECUTIVE "keys"): out8[q][k] = e4m3(clamp(oscale * (alpha * q8[q] . k8[k] + kbias[k] + qbias[q]))), 16-B stores of 16 consecutive k.
//   q | k projection: q rows = tokens e4m3(8 x^) [S][512], k rows = [Wq; Wk] as e4m3(W / s) [1024][512]  ->  q8 | k8 [S][1024] directly
//                     (no bf16 q | k, no conversion pass);
//   v projection:     q rows = Wv as e4m3(W / s) [512][512] (shared by the batch), k rows = tokens  ->  v8^T [512][pitch] directly.
// K = 512 = four k-steps, so a tile is 64 MFMAs per wave; the epilogue is one fma + clamp + convert per value.
__global__ __launch_bounds__(512, 2) void proj_fp8_kernel(const ProjFp8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 key tiles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int qtiles = (a.nq + QB - 1) / QB;
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int per_img = qtiles * nsplit;
    const int b = logical / per_img, qs = logical - b * per_img;
    const int qt = qs / nsplit, ksp = qs - qt * nsplit;
    const unsigned char* qb = a.q8 + (long long)b * a.q_bs;
    const unsigned char* kb = a.k8 + (long long)b * a.k_bs;
    const int row0 = qt * QB + wave * 32;

    i32x8 qf[2][4];
    float qbias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row0 + j * 16 + fr;
        const unsigned char* src = row < a.nq ? qb + (long long)row * a.ldq + fq * 32 : (const unsigned char*)a.zeros;
        const int step = row < a.nq ? 128 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[j][ks] = cat8(*(const i32x4*)(src + ks * step), *(const i32x4*)(src + ks * step + (row < a.nq ? 16 : 0)));
        qbias[j] = (a.qbias && row < a.nq) ? a.qbias[row] * a.oscale : 0.f;
    }
    const float alpha = a.alpha * a.oscale;

    const int sr_ = 2 * wave + (lane >> 5);
    const int skey = 32 * (sr_ >> 2) + (sr_ & 3);
    const int schunk = ((lane & 31) ^ sr_) << 4;
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        const int key_l = kt * KT + opaque8(skey);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int key = key_l + 16 * (jj >> 2) + 4 * (jj & 3);
            const void* src = key < a.nk ? (const void*)(kb + (long long)key * a.ldk + schunk) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(smem + buf * KBUF + (jj * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    int kbase[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) kbase[p][h] = fr * KROWB + ((((8 * p + 2 * fq + h) ^ fr) & 15) << 4);

    f32x4 acc[4][2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    unsigned clampw = 0;
    const int nkt_all = (a.nk + KT - 1) / KT;
    const int kt0 = (int)((long long)ksp * nkt_all / nsplit), kt1 = (int)((long long)(ksp + 1) * nkt_all / nsplit);
    unsigned char* const ob = a.out8 + (long long)b * a.o_bs;

    // half h of tile kt (in acc) -> 16 bytes per query tile: keys kt * 128 + 32 fq + 16 h .. + 15 of rows row0 + 16 j + fr
    auto epilogue_half = [&](int kt, int h) __attribute__((always_inline)) {
        const int key0 = kt * KT + 32 * fq + 16 * h;
        i32x4 outw[2];
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const f32x4 kbv = (a.kbias && key0 + 4 * mm + 3 < a.nk) ? *(const f32x4*)(a.kbias + key0 + 4 * mm) * a.oscale : zero4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(acc[mm][j][r], alpha, kbv[r] + qbias[j]);
                    v = __builtin_amdgcn_fmed3f(v, -448.f, 448.f);
                    e[r] = (key0 + 4 * mm + r < a.nk) ? v : 0.f;
                }
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w, true);
                clampw |= (((unsigned)w & 0x7f7f7f7fu) + 0x02020202u) & 0x80808080u;      // a byte of +-448: the value met the clamp
                outw[j][mm] = w;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = row0 + j * 16 + fr;
            if (row < a.nq && key0 < a.kext) *(i32x4*)(ob + (long long)row * a.ldo + key0) = outw[j];
        }
    };

    int it = 0;
    if (kt0 < kt1) stage(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < kt1) stage(kt + 1, (it + 1) & 1);
        const char* ks_base = smem + (it & 1) * KBUF;
        constexpr int AHEAD = 3, RING = 4;
        auto half = [&](int h /* compile-time */) __attribute__((always_inline)) {
            i32x8 kf[RING];
            auto frag = [&](int idx) __attribute__((always_inline)) -> i32x8 {
                const int ks = idx >> 2, m = 4 * h + (idx & 3);
                const char* p = ks_base + (ks >> 1) * 256 + m * 16 * KROWB;
                return cat8(*(const i32x4*)(p + kbase[ks & 1][0]), *(const i32x4*)(p + kbase[ks & 1][1]));
            };
#pragma unroll
            for (int p = 0; p < AHEAD; ++p) kf[p] = frag(p);
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                if (idx + AHEAD < 16) kf[(idx + AHEAD) % RING] = frag(idx + AHEAD);
                const int ks = idx >> 2, mm = idx & 3;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mm][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kf[idx % RING], qf[j][ks], ks == 0 ? zero4 : acc[mm][j], 0, 0, 0, 127, 0, 127);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        half(0);
        epilogue_half(kt, 0);
        __builtin_amdgcn_sched_barrier(0);
        half(1);
        epilogue_half(kt, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a.status && __any(clampw != 0u) && lane == 0) atomicOr(a.status, 2);          // VT_STATUS_FP8_SATURATED
}

// ---------------------------------------------------------------------------------------------------------------------------
// P.V: o[q][c] = (1 / (8 sum_q)) sum_k P8[q][k] v8^T[c][k]
constexpr int ROWB = KT;       // bytes per LDS row of the v^T tile (one channel, 128 keys)

template <int CB>
__global__ __launch_bounds__(512, 2) void attn_pv_fp8_kernel(const AttnPv8Args a) {
    constexpr int VBUF = CB * ROWB;                    // 32 KB (16 KB)
    constexpr int NCT = CB / 16;
    constexpr int NCP = D / CB;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 v^T tiles
    if (a.gate && *a.gate != a.gate_expect) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int qblocks = (a.S + QB - 1) / QB;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = logical % NCP;
    const int rest = logical / NCP;
    const int b = rest / qblocks, qb = rest - b * qblocks;
    const int nkt = (a.S + KT - 1) / KT;
    const int nslab = (a.S + 31) / 32;
    const int slab = qb * 8 + wave;
    const bool slab_real = slab < nslab;
    const unsigned char* pt = a.P8 + (long long)b * a.p_bs + (long long)(slab_real ? slab : 0) * vt_attn_p8_slab_stride(a.S) + lane * 16;
    const unsigned char* vb = a.vt8 + (long long)b * a.vt_bs + (long long)cb * CB * a.ldv;

    // ---- v^T tile staging: one wave-instruction = 8 LDS rows x 128 B; lane l -> row (l >> 3), physical chunk (l & 7),
    // logical chunk = physical ^ (row & 7); LDS row R holds channel (R & ~63) + (R & 3) + 4 ((R >> 4) & 3) + 16 ((R >> 2) & 3)
    // (interleaved cout map: a lane ends up with 16 consecutive channels).  Keys >= ldv16 (the written extent) come from the zero page.
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < CB / 64; ++jj) {
            const int R = (jj * 8 + wave) * 8 + lrow;
            const int ch = (R & ~63) + (R & 3) + 4 * ((R >> 4) & 3) + 16 * ((R >> 2) & 3);
            const int key = kt * KT + lchunk * 16;
            const void* src = key < a.kext ? (const void*)(vb + (long long)ch * a.ldv + key) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(smem + buf * VBUF + (jj * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    // A fragment (channel tile ct): LDS row ct*16 + fr, logical chunks 2 fq and 2 fq + 1, read in the order (2 fq + (fq & 1)) first --
    // with both lane quarters of a ds_read_b128 group on "their" first chunk the 16 lanes collide; crossed they do not
    const int c_first = 2 * fq + (fq & 1), c_second = 2 * fq + 1 - (fq & 1);
    const int foff0 = fr * ROWB + ((c_first ^ (fr & 7)) << 4), foff1 = fr * ROWB + ((c_second ^ (fr & 7)) << 4);

    f32x4 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x8 pf[2], pn[2];
    // row sums of P8 by the matrix pipe: A = a tile of e4m3 1.0 (0x38), so accs[j][.] of lane (., fr) = sum_k P8[query 16 j + fr][k]
    const i32x8 ones8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    f32x4 accs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto load_p = [&](int kt, i32x8 (&d)[2]) __attribute__((always_inline)) {
        const unsigned char* s = pt + (long long)kt * 4096;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            d[j] = cat8(__builtin_nontemporal_load((const i32x4*)(s + (j * 2) * 1024)), __builtin_nontemporal_load((const i32x4*)(s + (j * 2 + 1) * 1024)));
    };
    stage(0, 0);
    load_p(0, pf);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                               // vmcnt(0) + barrier: tile kt (and pf) landed, buffer (kt+1)&1 free
        if (kt + 1 < nkt) {
            stage(kt + 1, (kt + 1) & 1);
            load_p(kt + 1, pn);
        }
        const char* vs = smem + (kt & 1) * VBUF;
        constexpr int AHEAD = 3, RING = 4;
        i32x8 af[RING];
        auto frag = [&](int ct) __attribute__((always_inline)) -> i32x8 {
            return cat8(*(const i32x4*)(vs + ct * 16 * ROWB + foff0), *(const i32x4*)(vs + ct * 16 * ROWB + foff1));
        };
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) af[p] = frag(p);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (ct + AHEAD < NCT) af[(ct + AHEAD) % RING] = frag(ct + AHEAD);
            acc[ct][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[ct % RING], pf[0], acc[ct][0], 0, 0, 0, 127, 0, 127);
            acc[ct][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[ct % RING], pf[1], acc[ct][1], 0, 0, 0, 127, 0, 127);
            __builtin_amdgcn_sched_barrier(0);
        }
        accs[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones8, pf[0], accs[0], 0, 0, 0, 127, 0, 127);
        accs[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones8, pf[1], accs[1], 0, 0, 0, 127, 0, 127);
        if (kt + 1 < nkt) { pf[0] = pn[0]; pf[1] = pn[1]; }
    }
    if (!slab_real) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = slab * 32 + j * 16 + fr;
        if (row >= a.S) continue;
        const float rs = a.out_scale / accs[j][0];
        bf16_t* o = a.o + (long long)b * a.o_bs + (long long)row * a.ldo + cb * CB + 16 * fq;
#pragma unroll
        for (int G = 0; G < CB / 64; ++G)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                bf16x8 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { hv[r] = (bf16_t)(acc[4 * G + i][j][r] * rs); hv[4 + r] = (bf16_t)(acc[4 * G + i + 1][j][r] * rs); }
                *(bf16x8*)(o + 64 * G + 4 * i) = hv;
            }
    }
}

// v^T bf16 [C][ldv] -> e4m3(scale v^T) [C][ld8]; keys >= S are written as zero up to kext (a multiple of 16)
__global__ __launch_bounds__(256) void attn_vt_to_fp8_kernel(const bf16_t* __restrict__ vt, long long vt_bs, int ldv, unsigned char* __restrict__ v8,
                                                             long long v8_bs, int ld8, int S, int kext, int C, float scale, int* __restrict__ status) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int k0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (k0 >= kext) return;
    const bf16_t* src = vt + (long long)b * vt_bs + (long long)ch * ldv + k0;
    i32x4 o = {0, 0, 0, 0};
    float big = 0.f;                                           // largest |scale v| converted: beyond 448 the e4m3 value is a clamp (status bit 1)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float f[8];
        if (k0 + 8 * hh + 8 <= ((S + 7) / 8) * 8) {
            const bf16x8 v = *(const bf16x8*)(src + 8 * hh);       // (v^T is written up to the next multiple of 8, zeros beyond S)
#pragma unroll
            for (int r = 0; r < 8; ++r) f[r] = k0 + 8 * hh + r < S ? (float)v[r] * scale : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) f[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { big = fmaxf(big, fabsf(f[r])); f[r] = __builtin_amdgcn_fmed3f(f[r], -448.f, 448.f); }
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        o[2 * hh] = w0; o[2 * hh + 1] = w1;
    }
    *(i32x4*)(v8 + (long long)b * v8_bs + (long long)ch * ld8 + k0) = o;
    if (status && !(big <= 448.f)) atomicOr(status, 2);        // (NaN counts: it was turned into +-448 by the clamp)
}

}  // namespace

bool vt_attn_fp8_supported(int S, int C) { return C == D && S > 0; }
long long vt_attn_p8_bytes(int S) { return (long long)((S + 31) / 32) * vt_attn_p8_slab_stride(S); }

hipError_t vt_launch_attn_qk_fp8(const AttnQk8Args& a, hipStream_t s) {
    if (!a.qk8 || (a.mode == 1 && !a.rowout) || !a.zeros || a.batch <= 0 || !vt_attn_fp8_supported(a.S, a.C) || (a.mode != 1 && a.mode != 3)) return hipErrorInvalidValue;
    if ((a.ldq % 16) || (a.qk_bs % 16) || a.row_bs < a.S) return hipErrorInvalidValue;
    if (a.mode == 3 && (!a.P8 || !a.rowin || (a.p_bs % 16) || a.p_bs < vt_attn_p8_bytes(a.S))) return hipErrorInvalidValue;
    if (a.mode == 1 && a.nsplit > 1) return hipErrorInvalidValue;
    if (a.kstride < 0 || (a.mode == 3 && a.kstride > 1)) return hipErrorInvalidValue;
    if ((long long)a.S * a.ldq >= (1LL << 31)) return hipErrorInvalidValue;
    if (a.nsplit > 1 && a.nsplit != 2 && a.nsplit != 4) return hipErrorInvalidValue;
    const long long nblk = (long long)((a.S + QB - 1) / QB) * a.batch * (a.nsplit > 1 ? a.nsplit : 1);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)attn_qk_fp8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
        return e != hipSuccess ? e : hipFuncSetAttribute((const void*)attn_qk_fp8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
    });
    if (ea != hipSuccess) return ea;
    if (a.mode == 1) hipLaunchKernelGGL(attn_qk_fp8_kernel<1>, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    else hipLaunchKernelGGL(attn_qk_fp8_kernel<3>, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    return hipGetLastError();
}

hipError_t vt_launch_proj_fp8(const ProjFp8Args& a, hipStream_t s) {
    if (!a.q8 || !a.k8 || !a.out8 || !a.zeros || a.batch <= 0 || a.nq <= 0 || a.nk <= 0 || a.C != D) return hipErrorInvalidValue;
    if ((a.ldq % 16) || (a.q_bs % 16) || (a.ldk % 16) || (a.k_bs % 16) || (a.ldo % 16) || (a.o_bs % 16) || (a.kext % 16)) return hipErrorInvalidValue;
    if (a.kext > a.ldo || a.kext < a.nk - 15 || ((uintptr_t)a.out8 % 16)) return hipErrorInvalidValue;      // whole 16-byte chunks inside a row's pitch
    if (a.kbias && (a.nk % 4)) return hipErrorInvalidValue;
    if ((long long)a.nq * a.ldq >= (1LL << 31) || (long long)a.nk * a.ldk >= (1LL << 31)) return hipErrorInvalidValue;
    const int nkt = (a.nk + KT - 1) / KT;
    if (a.nsplit < 0 || a.nsplit > nkt) return hipErrorInvalidValue;
    const long long nblk = (long long)((a.nq + QB - 1) / QB) * a.batch * (a.nsplit > 1 ? a.nsplit : 1);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        return hipFuncSetAttribute((const void*)proj_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF);
    });
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(proj_fp8_kernel, dim3((unsigned)nblk), dim3(512), 2 * KBUF, s, a);
    return hipGetLastError();
}

hipError_t vt_launch_attn_pv_fp8(const AttnPv8Args& a, hipStream_t s) {
    if (!a.P8 || !a.vt8 || !a.o || !a.zeros || a.batch <= 0 || !vt_attn_fp8_supported(a.S, a.C)) return hipErrorInvalidValue;
    if ((a.ldv % 16) || (a.vt_bs % 16) || (a.ldo % 8) || (a.o_bs % 8) || (a.p_bs % 16) || (a.kext % 16) || a.kext > a.ldv) return hipErrorInvalidValue;
    if ((long long)a.C * a.ldv >= (1LL << 31)) return hipErrorInvalidValue;
    const long long qblk = (long long)((a.S + QB - 1) / QB) * a.batch;
    const bool narrow = qblk * 2 < 192;
    const long long nblk = qblk * (narrow ? 4 : 2);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)attn_pv_fp8_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * ROWB);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_pv_fp8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * ROWB);
        return e;
    });
    if (ea != hipSuccess) return ea;
    if (narrow) hipLaunchKernelGGL(attn_pv_fp8_kernel<128>, dim3((unsigned)nblk), dim3(512), 2 * 128 * ROWB, s, a);
    else hipLaunchKernelGGL(attn_pv_fp8_kernel<256>, dim3((unsigned)nblk), dim3(512), 2 * 256 * ROWB, s, a);
    return hipGetLastError();
}

hipError_t vt_launch_attn_vt_to_fp8(const bf16_t* vt, long long vt_bs, int ldv, unsigned char* v8, long long v8_bs, int ld8, int S, int kext,
                                    int C, int batch, float scale, int* status, hipStream_t s) {
    if (!vt || !v8 || S <= 0 || C <= 0 || batch <= 0 || (kext % 16) || kext > ld8 || (ld8 % 16) || (ldv % 8)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_vt_to_fp8_kernel, dim3((unsigned)((kext / 16 + 255) / 256), (unsigned)C, (unsigned)batch), dim3(256), 0, s, vt, vt_bs, ldv, v8,
                       v8_bs, ld8, S, kext, C, scale, status);
    return hipGetLastError();
}
