// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: launchers call it once per device
// (bit = current device id; racing first calls at worst set the attribute twice).  `set` performs the hipFuncSetAttribute calls.
#include <atomic>
#include <type_traits>
template <typename F>
inline hipError_t vt_once_per_device(std::atomic<unsigned long long>& done, F set) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = set();
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

#define VT_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define VT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float vt_silu(float x) {
    // x * sigmoid(x): v_mul + v_exp_f32, v_add, v_rcp_f32 (1 ulp, far below the bf16 / e4m3 outputs), v_mul.  NOT __frcp_rn: that is
    // the correctly rounded reciprocal, a ten-instruction v_div_scale / v_div_fmas / v_div_fixup sequence that made the GroupNorm
    // pass VALU-bound instead of HBM-bound (round 2: 18.6 -> 8.5 VALU instructions per element).
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float vt_sigmoid_accurate(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (blocks b and b+8 share an L2).  Give each XCD a contiguous run of logical tiles so tiles that
// share operand panels hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ int vt_xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm partial statistics from an MFMA epilogue.
// A lane holds v[i][j] = 4 consecutive couts (cg .. cg+3, cg = cout_base + i*16 + (lane>>4)*4) of pixel
// column (lane & 15) for TP pixel rows j; `valid` has bit j set where that pixel exists.
// Produces one (n, mean, M2) triple per GroupNorm group covered by the workgroup and writes it to
// out[(group)*3].  Sums are taken relative to a per-(wave, i, fq) pivot (the wave's first pixel), so
// E[d^2]-E[d]^2 does not cancel when |mean| >> std; merges across lanes / waves use Chan's formula in a
// fixed order (deterministic).  cpg = channels per group, one of 4, 8, 16.  lds: >= WP*(BC/cpg)*3 floats,
// free for reuse (callers barrier before).
__device__ __forceinline__ void vt_chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
}

template <int TC, int TP>
__device__ __forceinline__ void vt_gn_epilogue_partials(const f32x4 (&v)[TC][TP], unsigned valid, int cpg, int wp,
                                                        int nwp, int wave_cout0 /* cout offset of this wave inside the block */,
                                                        int block_couts, float* lds, float* out /* block's first group */) {
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int gpb = block_couts / cpg;
    const float nl = 4.0f * (float)__popc(valid);
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const float piv = __shfl(v[i][0][0], lane & 48, 64);
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            if ((valid >> j) & 1u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = v[i][j][r] - piv; s += d; ss = fmaf(d, d, ss); }
            }
        }
        float n = nl;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); n += __shfl_xor(n, o, 64); }
        float mean = 0.f, m2 = 0.f;
        if (n > 0.f) { const float ms = s / n; mean = piv + ms; m2 = fmaxf(ss - s * ms, 0.f); }
        if (cpg >= 8) {
            const float nb = __shfl_xor(n, 16, 64), mb = __shfl_xor(mean, 16, 64), m2b = __shfl_xor(m2, 16, 64);
            if ((fq & 1) == 0) vt_chan_merge(n, mean, m2, nb, mb, m2b);
        }
        if (cpg >= 16) {
            const float nb = __shfl_xor(n, 32, 64), mb = __shfl_xor(mean, 32, 64), m2b = __shfl_xor(m2, 32, 64);
            if (fq == 0) vt_chan_merge(n, mean, m2, nb, mb, m2b);
        }
        const int per = cpg >> 2;                      // fq lanes per group
        if (fr == 0 && (fq % per) == 0) {
            const int lg = (wave_cout0 + i * 16 + fq * 4) / cpg;
            float* d = lds + (wp * gpb + lg) * 3;
            d[0] = n; d[1] = mean; d[2] = m2;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < gpb) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int w = 0; w < nwp; ++w) {
            const float* d = lds + (w * gpb + threadIdx.x) * 3;
            vt_chan_merge(n, mean, m2, d[0], d[1], d[2]);
        }
        float* o = out + threadIdx.x * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

// Sum over the 16 lanes of a DPP row (lanes sharing lane >> 4), result in every lane of the row.  Four v_add_f32 with DPP
// operands (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips:
// after the two quad steps every lane of a quad holds the quad's sum, so mirroring within 8 and then within 16 lanes pairs
// disjoint partial sums.
__device__ __forceinline__ float vt_row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}

// ---- packed fp32 and op_sel: a hazard found in round 4 ----
// In the halo conv kernels `v_pk_add_f32 d, a, v[p-1:p] op_sel:[0,1]` (the HIGH register of the source pair routed into the LOW lane) sometimes
// computed its low lane with 0.0 in place of that register -- in lanes 48..63, load dependent (never at batch 1, every launch at batch 4), whatever wrote
// the pair (ds_bpermute or v_readlane) and however long before: run to run, one element of one GroupNorm partial was summed against the wrong
// pivot (tests/diagnostics/halo_partials_lane_dump.py shows the lane and the element; -DGNIL_ASM_MODE=1..4 place the instruction by hand: modes 1-3
// differ in 300 of 300 launches, the mirrored op_sel_hi:[1,0] form in none).  The same code with the pivot in a materialised register pair is
// bit-stable over thousands of launches.  Rule, enforced by tests/test_isa_lint.py over every kernel of the BUILT library: no packed-fp32 instruction
// with a source op_sel.
//   - hand-written f32x2 code pins its broadcast operands with VT_PIN_PAIR (an empty asm that makes the pair a real one);
//   - kernels where the compiler forms packed ops from scalar code carry VT_NO_PACKED_F32.
// -DGNIL_OP_SEL_PIVOT rebuilds the round-3 code (tools/build_variant.sh opsel "-DGNIL_OP_SEL_PIVOT"), -DGNIL_DUMP the per-lane dump.
#if defined(__HIP_DEVICE_COMPILE__)
#define VT_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define VT_NO_PACKED_F32                         // (the host pass of hipcc does not know the feature name)
#endif
#define VT_PIN_PAIR(p2) asm volatile("" : "+v"(p2))
#ifdef GNIL_DUMP
static __device__ float* g_gnil_dump = nullptr;     // diagnostics build only: per-lane (s, ss, pivot, v[q][0][0]) before the lane reduction
#endif
// Variant for the "interleaved" cout map of conv3x3_halo: a lane's registers hold 16 CONSECUTIVE couts of its pixel,
// cout = wave_cout0 + 16*fq + 4*i + r (tile i, register r), so every GroupNorm group (4, 8 or 16 channels) lives in
// ONE lane; only the 16 pixel columns (fr) are merged across lanes.
template <int TC, int TP>
__device__ __forceinline__ void vt_gn_epilogue_partials_il(const f32x4 (&v)[TC][TP], unsigned valid, int cpg, int wp,
                                                           int nwp, int wave_cout0, int block_couts, float* lds,
                                                           float* out) {
    static_assert(TC == 4, "interleaved map covers 16 couts per lane");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int gpb = block_couts / cpg;
    const int tpg = cpg >> 2;                      // tiles (4-cout register groups) per GroupNorm group: 1, 2 or 4
    // values per group in this lane's 16-lane row: 4 channels x tpg tiles x (valid pixels of the row); the pixel count is
    // the same for every fq, so one ballot of the per-lane row masks gives it without shuffling floats
    float n = 0.f;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const unsigned long long m = __ballot((valid >> j) & 1u);
        n += (float)__popcll(m & 0xffffull);       // lanes 0..15 = the 16 pixel columns (identical for all four fq rows)
    }
    n *= 4.0f * (float)tpg;
    const bool full = __ballot(valid != (1u << TP) - 1u) == 0ull;       // every pixel of the wave's tile is inside the image
    // This phase is latency (its VALU issue hides behind the other resident workgroup's MFMAs, but a short main loop does not
    // cover a long epilogue): all groups' pivots are fetched first, their sums are independent chains, the lane reductions are
    // DPP, and ONE branch writes every partial.
    auto partials = [&](auto tpg_tag, auto full_tag) {
        constexpr int TPG = decltype(tpg_tag)::value;        // tiles per group, compile-time here: no control flow between the chains
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int NG = TC / TPG;
        float piv[NG], s[NG], ss[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
#if defined(GNIL_ASM_MODE) && GNIL_ASM_MODE == 3    // experiment: the pivot never passes through the LDS
            const int x0 = __builtin_bit_cast(int, v[q * TPG][0][0]);
            const int r0 = __builtin_amdgcn_readlane(x0, 0), r1 = __builtin_amdgcn_readlane(x0, 16), r2 = __builtin_amdgcn_readlane(x0, 32),
                      r3 = __builtin_amdgcn_readlane(x0, 48);
            piv[q] = __builtin_bit_cast(float, fq == 0 ? r0 : fq == 1 ? r1 : fq == 2 ? r2 : r3);
#else
            piv[q] = __shfl(v[q * TPG][0][0], lane & 48, 64);
#endif
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
#ifdef GNIL_ASM_MODE
            // experiments on the hazard of section 4.14 inside the kernel that shows it: the FIRST subtract of each group is a hand-placed packed add
            // whose pivot operand is read first by this instruction; everything after it uses the pinned pair.  Modes: 1 = op_sel:[0,1], pivot in the
            // high half; 2 = the same after s_waitcnt lgkmcnt(0) and 16 idle cycles; 3 = mode 1 with a v_readlane pivot; 4 = op_sel_hi:[1,0], pivot
            // in the low half (control); 5 = mode 1 with 1000.0 instead of 0.0 in the low half (what does a wrong lane read: the low register, or zero?).
            f32x2 pr = GNIL_ASM_MODE == 4 ? f32x2{piv[q], 0.f} : f32x2{GNIL_ASM_MODE == 5 ? 1000.f : 0.f, piv[q]};     // mode 5: a sentinel in the low half
            const f32x2 a01 = {v[q * TPG][0][0], v[q * TPG][0][1]};
            f32x2 d_first;
#if GNIL_ASM_MODE == 2
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(pr));
#endif
#if GNIL_ASM_MODE == 4
            asm volatile("v_pk_add_f32 %0, %2, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d_first), "+v"(pr) : "v"(a01));
            piv[q] = pr[0];
#else
            asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d_first), "+v"(pr) : "v"(a01));
            piv[q] = pr[1];
#endif
#endif
            f32x2 p2 = {piv[q], piv[q]};
#ifndef GNIL_OP_SEL_PIVOT
            VT_PIN_PAIR(p2);                           // without it the compiler routes piv[q] into the packed ops by op_sel: see above
#endif
            f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};    // two accumulators per sum: packed adds / fmas (v_pk_add_f32, v_pk_fma_f32)
#pragma unroll
            for (int i = q * TPG; i < (q + 1) * TPG; ++i) {
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if (FULL || ((valid >> j) & 1u)) {
                        f32x2 d0 = f32x2{v[i][j][0], v[i][j][1]} - p2;
                        const f32x2 d1 = f32x2{v[i][j][2], v[i][j][3]} - p2;
#ifdef GNIL_ASM_MODE
                        if (i == q * TPG && j == 0) d0 = d_first;
#endif
                        s2 += d0; q2 += d0 * d0;
                        s2 += d1; q2 += d1 * d1;
                    }
                }
            }
            s[q] = s2[0] + s2[1]; ss[q] = q2[0] + q2[1];
#ifdef GNIL_DUMP
            if (g_gnil_dump) {
                float* dd = g_gnil_dump + (((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4 + q) * 4;
                dd[0] = s[q]; dd[1] = ss[q]; dd[2] = piv[q]; dd[3] = v[q * TPG][0][0];
            }
#endif
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) { s[q] = vt_row16_sum(s[q]); ss[q] = vt_row16_sum(ss[q]); }
        if (fr == 0) {
            const float rn = n > 0.f ? 1.0f / n : 0.f;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const float ms = s[q] * rn;
                const int lg = (wave_cout0 + 16 * fq + 4 * q * TPG) / cpg;
                float* d = lds + (wp * gpb + lg) * 3;
                d[0] = n; d[1] = n > 0.f ? piv[q] + ms : 0.f; d[2] = n > 0.f ? fmaxf(ss[q] - s[q] * ms, 0.f) : 0.f;
            }
        }
    };
    auto partials_f = [&](auto tpg_tag) {
        if (full) partials(tpg_tag, std::true_type{}); else partials(tpg_tag, std::false_type{});
    };
    if (tpg == 1) partials_f(std::integral_constant<int, 1>{});
    else if (tpg == 2) partials_f(std::integral_constant<int, 2>{});
    else partials_f(std::integral_constant<int, 4>{});
    __syncthreads();
    if ((int)threadIdx.x < gpb) {
        float nn = 0.f, mean = 0.f, m2 = 0.f;
        for (int w = 0; w < nwp; ++w) {
            const float* d = lds + (w * gpb + threadIdx.x) * 3;
            vt_chan_merge(nn, mean, m2, d[0], d[1], d[2]);
        }
        float* o = out + threadIdx.x * 3;
        o[0] = nn; o[1] = mean; o[2] = m2;
    }
}

// K-step (inside a 32-channel chunk) at which conv3x3_halo processes filter tap (ky, kx) = (tap / 3, tap % 3): the taps
// run kx-major so that one column shift's halo rows serve all three ky from registers.
__host__ __device__ inline int vt_halo_step_of_tap(int tap /*0..8*/) { return (tap % 3) * 3 + tap / 3; }

// K-step (inside a 32-channel chunk) at which conv3x3_s2_halo processes filter tap (ky, kx) = (tap / 3, tap % 3): plane-major
// (plane = (ky & 1, kx & 1)), dx-major inside a plane -- see the table at the top of conv3x3_s2_halo.hip.
__host__ __device__ inline int vt_s2_step_of_tap(int tap /*0..8*/) { return (int)((0x351786240ULL >> (4 * tap)) & 15); }

// LDS row (inside a wave's 64-cout group) that must hold cout_local, so that MFMA tile i / A-row r' lands on
// cout_local = (r' & 3) + 4*i + 16*(r' >> 2).  Used by the host and device weight packers.
__host__ __device__ inline int vt_halo_row_of_cout(int cout_local /*0..63*/) {
    const int r = cout_local & 3, i = (cout_local >> 2) & 3, q = cout_local >> 4;
    return 16 * i + 4 * q + r;
}
