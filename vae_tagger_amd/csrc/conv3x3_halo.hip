// 3x3 / stride 1 / pad 1 convolution as a halo-tile implicit GEMM on CDNA4 matrix cores.
// 82.6 % of the encoder's FLOPs are these convs.
//
// Versus the generic implicit GEMM (conv_gemm.hip), which re-gathers the X tile for every tap:
//   * the workgroup's pixel tile is a 2-D patch (ROWS x 16 pixels); its (ROWS+2) x 18 halo is staged
//     into LDS ONCE per 32-channel chunk and all 9 taps read it at shifted row addresses
//     -> global->LDS traffic per K-step drops from (BP+BC)*128 B to ~BC*64 B + halo/9;
//   * K-step = (32-channel chunk, tap): one v_mfma_f32_16x16x32_bf16 per 16x16 output tile;
//   * weights arrive through an NW-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4) that is
//     kept NW-1 K-steps ahead; waves wait with a COUNTED s_waitcnt vmcnt(N) (never 0 in steady state)
//     and meet at ONE raw s_barrier per K-step, so DMA stays in flight across barriers;
//   * LDS rows are 64 B; bank-conflict-free for every tap shift with physical chunk =
//     logical chunk ^ (((row >> 2) & 1) << 1), applied to the DMA's per-lane SOURCE address (the DMA
//     destination is lane-linear) and again on the ds_read_b128 side.
//
// Weights are pre-packed on the host as Wp[chunk32][step][Cout][32] bf16 (step = kx*3 + ky: dx-major) so that every
// K-step's tile is one contiguous BC*64-byte block.  MFMA orientation as in conv_gemm.hip: weights = A operand (rows =
// cout), pixels = B operand; the packed cout rows are interleaved so that a lane holds 16 consecutive couts of one pixel.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

#ifdef GNIL_DUMP        // diagnostics build only (tests/diagnostics/halo_partials_lane_dump.py)
extern "C" int vt_debug_gnil_dump(float* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gnil_dump), &buf, sizeof(buf)); }
#endif
#ifdef HALO_STAMP
// Diagnostic build only (tools/stamp_halo.py): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of one workgroup.
__device__ unsigned long long* g_halo_stamps = nullptr;
extern "C" int vt_debug_halo_stamps(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_halo_stamps), &buf, sizeof(buf));
}
// ... or only the nth launch (counted from this call) whose (H, Cin) match: the launcher switches the device pointer on the stream
static unsigned long long* s_stamp_ptrs[2] = {nullptr, nullptr};      // [0] = off, [1] = the buffer
static int s_stamp_H = 0, s_stamp_Cin = 0, s_stamp_nth = -1, s_stamp_seen = 0;
static bool s_stamp_on = false;
extern "C" int vt_debug_halo_stamps_nth(unsigned long long* buf, int H, int Cin, int nth) {
    s_stamp_ptrs[1] = buf; s_stamp_H = H; s_stamp_Cin = Cin; s_stamp_nth = buf ? nth : -1; s_stamp_seen = 0; s_stamp_on = false;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_halo_stamps), &s_stamp_ptrs[0], sizeof(void*));
}
#ifndef STAMP_TID
#define STAMP_TID 0
#endif
#define STAMP(slot)                                                                                            \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (g_halo_stamps && threadIdx.x == STAMP_TID) g_halo_stamps[(long long)blockIdx.x * 16 + (slot)] = t_; \
    } while (0)
#define STAMP_CLK(slot)                                                                                        \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (g_halo_stamps && threadIdx.x == STAMP_TID) g_halo_stamps[(long long)blockIdx.x * 16 + (slot)] = t_; \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_CLK(slot) do {} while (0)
#endif

namespace {

// F16: the same instruction stream with v_mfma_f32_16x16x32_f16 on fp16 operand bits (vt_set_flag 18: 11 significand bits instead of
// bf16's 8 at the same 2 B per element; -DHALO_F16 forces it for the operand-precision experiment of tools/bench_f16_operands.py)
template <bool F16>
__device__ __forceinline__ f32x4 halo_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef HALO_F16
    constexpr bool H = true;
#else
    constexpr bool H = F16;
#endif
#ifdef GNIL_NO_MFMA          // experiment: the main loop without its matrix instructions (results are wrong by design)
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#endif
    if constexpr (H) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
}

constexpr int HB = 64;        // bytes per LDS row (32 bf16)
constexpr int TW = 16;        // tile width in pixels (one MFMA column block)
constexpr int HWID = TW + 2;  // halo width
constexpr int NW_DEFAULT = 6;  // weight ring depth of the 1-workgroup-per-CU variants

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
        C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// s_waitcnt vmcnt(n) that also names the registers an inline-asm load wrote: nothing that consumes them
// can be scheduled above the wait (hipcc neither counts asm loads nor knows when their data lands).
template <typename V>
__device__ __forceinline__ void wait_vmcnt_tied(int n, V& r0, V& r1) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" : "+v"(r0), "+v"(r1)::"memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1)::"memory"); break;
    }
}


// Identity the optimiser cannot see through: stops loop-invariant address arithmetic built on `v` from being
// hoisted into (and spilled from) long-lived registers -- it is recomputed where it is used instead.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <typename V>
__device__ __forceinline__ void wait_vmcnt_tied1(int n, V& r0) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" : "+v"(r0)::"memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0)::"memory"); break;
    }
}

// XT selects how the X halo reaches LDS:
//   0  bf16 activations, copied as they are by LDS-DMA;
//   1  fp32 tensor (the residual stream), 2  bf16 tensor (a conv1 output): loaded to registers by
//      hand-counted inline-asm global loads, GroupNorm'ed with the per-(image, channel) (scale, shift) the
//      previous layer's epilogue statistics produced, SiLU'ed, rounded to bf16 and written to LDS -- the
//      standalone GroupNorm+SiLU pass (one read + one write of the whole tensor) disappears.
// TPW = tile rows per wave: 8 (wave tile 128 px x 64 couts, <= 256 VGPRs) or 4 (64 px x 64 couts, <= 128 VGPRs).
// NW = weight ring depth.  Workgroup shapes (waves = WP x WC):
//   <2,2,0,8,4>  4 waves x 256 VGPRs, 16x16 px x 128 couts, 80 KB LDS: TWO workgroups per CU -- the default for every
//                plain-input layer: a tile spends ~12 us outside its main loop (index math, first-DMA latency, epilogue
//                VALU + stores, statistics; tools/stamp_halo.py) and the other workgroup's MFMAs run under them;
//   <4,2,0,4,4>  8 waves x 128 VGPRs, same tile, two workgroups per CU (the earlier form of the same idea);
//   <2,4,.,8,6> / <4,2,.,8,6>  8 waves x 256 VGPRs, 256 / 128 couts, one workgroup per CU (and the only shapes of XT != 0).
template <int WP, int WC, int XT, int TPW, int NW, bool F16 = false>
__global__ __launch_bounds__(64 * WP * WC, TPW == 4 ? 4 : TPW == 16 ? 1 : 2)
void conv3x3_halo_kernel(const Conv3x3Args a) {
    static_assert(!F16 || XT == 0, "fp16 operands: plain-input tiles only");
    constexpr int NWV = WP * WC;                 // waves per workgroup
    constexpr int NT = 64 * NWV;
    static_assert((NWV == 8 && (TPW == 8 || (TPW == 4 && XT == 0))) || (NWV == 4 && (TPW == 8 || TPW == 16) && XT == 0),
                  "8 waves x 8 rows, 8 waves x 4 rows (two workgroups per CU), 4 waves x 8 rows (two workgroups per CU), or -- the round-4 "
                  "experiment -- 4 waves x 16 rows: ONE wave per SIMD with up to 512 registers (accumulators in AGPRs)");
    constexpr int ROWS = WP * TPW;               // tile rows (each wave: TPW rows x 16 px)
    constexpr int BC = WC * 64;                  // couts per workgroup (each wave: 64)
    constexpr int TP = TPW, TC = 4;
    constexpr int HROWS = (ROWS + 2) * HWID;     // halo pixels
    constexpr int NLD = NWV;                     // every wave issues its share of the DMA pieces
    constexpr int NXW = ((HROWS + 15) / 16 + NLD - 1) / NLD;   // X staging wave-instructions per issuing wave (16 rows each)
    constexpr int XBUF = NXW * NLD * 16 * HB;    // bytes per X halo buffer
    constexpr int WPCS = BC / 16;                // W DMA pieces per K-step
    constexpr int WPW = (WPCS + NLD - 1) / NLD;  // ... per issuing wave (waves >= WPCS issue none when WPCS < NLD)
    constexpr int WBUF = BC * HB;                // bytes per W stage
    constexpr int LX = XT == 1 ? 2 : 1;          // register loads per staged row (8 channels)
    constexpr int DLY = XT == 1 ? 2 : 3;         // K-steps between a row's load and its normalise+write
    static_assert(WPW >= 1, "BC >= 128");
    static_assert(XT == 0 || NXW + DLY <= 9, "staging must finish inside the chunk");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbase = smem;                    // 2 halo buffers
    char* const wbase = smem + 2 * XBUF;         // NW weight stages
    float* const ssl = (float*)(wbase + NW * WBUF);   // XT != 0: (scale, shift) per input channel

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WC, wc = wave % WC;
    const bool early = wave < 4;                 // staging phase of this wave (see the K-loop)
    // W DMA pieces THIS wave issues per K-step (the vmcnt arithmetic below is per wave)
    const int wpw = (WPCS % NLD == 0) ? WPW : (wave < WPCS % NLD ? WPW : WPW - 1);

    STAMP(0);
    // ---- tile coordinates.  The divisors are launch constants: the host passes 2^40 / d + 1 multipliers (exact for
    // n * d < 2^40), so the five runtime integer divisions (~50 instructions each) in front of the first DMA become multiplies.
    const int tiles_x = a.tiles_x, ctiles = a.ctiles, per_img = a.per_img;
    auto fdiv = [](int n, unsigned long long m, int d) -> int {
        return m ? (int)(((unsigned long long)(unsigned)n * m) >> 40) : n / d;
    };
    int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int b = fdiv(logical, a.m_per_img, per_img);
    logical -= b * per_img;
    const int tile = fdiv(logical, a.m_ctiles, ctiles);
    const int ct = logical - tile * ctiles;
    const int tyi = fdiv(tile, a.m_tiles_x, tiles_x);
    const int ty0 = tyi * ROWS, tx0 = (tile - tyi * tiles_x) * TW;
    const int c0 = ct * BC;

    const long long img = (long long)b * a.H * a.W * a.Cin;
    const bf16_t* Xb = a.X ? a.X + img : nullptr;
    const float* Xf = a.Xf32 ? a.Xf32 + img : nullptr;
    const int nchunk = a.Cin >> 5;
    const int nk = nchunk * 9;

    // ---- staging bookkeeping.  One wave-instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical
    // chunk (l & 3); logical chunk = physical ^ swz(row), swz(row) = ((row >> 2) & 1) << 1 = ((l >> 4) & 1) << 1.
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    // XT != 0 keeps one source offset per staged row (also needed when the row is written); XT == 0 derives the
    // DMA addresses on the fly from the lane's first halo row (fewer live registers in the MFMA loop).
    constexpr int NXS = XT == 0 ? 1 : NXW;
    int xsrc[NXS];          // element offset of (halo pixel, logical chunk) in the image, or -1 if outside
    if constexpr (XT != 0) {
#pragma unroll
        for (int j = 0; j < NXW; ++j) {
            const int hr = (j * NLD + wave) * 16 + drow;
            const int hy = hr / HWID, hx = hr - hy * HWID;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            const bool v = hr < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            xsrc[j] = v ? (iy * a.W + ix) * a.Cin + dchunk * 8 : -1;
        }
    } else {
        xsrc[0] = 0;
    }
    const int hr0 = wave * 16 + drow;            // this lane's halo row in piece j = 0 (XT == 0 path)
    const int hy0 = hr0 / HWID, hx0 = hr0 - (hr0 / HWID) * HWID;
    // W: Wp[chunk][tap][Cout][32]; this lane's row inside the K-step tile, piece j = 0
    const int wsrc0 = (c0 + wave * 16 + drow) * 32 + dchunk * 8;
    const int wstep = a.Cout * 32;               // elements between consecutive K-steps

    auto issue_x_dma = [&](int chunk) {          // XT == 0
#ifdef EXP_NO_DMA                // timing experiment only (wrong results): the first two halos and the first NW weight tiles serve every K-step
        if (chunk >= 2) return;
#endif
        char* dst = xbase + (chunk & 1) * XBUF;
        int hy = opaque(hy0), hx = hx0, hr = hr0;
#pragma nounroll
        for (int j = 0; j < NXW; ++j) {                              // rolled: few live temporaries beside the accumulators
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            const bool v = hr < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const void* src = v ? (const void*)(Xb + ((iy * a.W + ix) * a.Cin + dchunk * 8 + chunk * 32)) : a.zeros;
#ifndef GNIL_NO_DMA          // experiment: the main loop without its LDS-DMA (results are wrong by design)
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(dst + (j * NLD + wave) * 1024), 16, 0, 0);
#else
            asm volatile("" :: "v"(src));
#endif
            hr += NLD * 16;                                         // next piece: NLD*16 halo rows further
            hx += (NLD * 16) % HWID; hy += (NLD * 16) / HWID;
            if (hx >= HWID) { hx -= HWID; ++hy; }
        }
    };
    // K-step t of the kernel -> weight tile in the (kx-major) packed order.  Every variant walks the taps kx-major except
    // the 128-VGPR one, which walks them ky-major (fewer live fragment addresses) and permutes its weight fetches instead.
    constexpr bool KYMAJOR = XT == 0 && TPW == 4;
    auto issue_w = [&](int t, int tap_of_t /* t % 9, a compile-time constant at every call site */) {
#ifdef EXP_NO_DMA
        if (t >= NW) return;
#endif
        char* dst = wbase + (t % NW) * WBUF;
        const int src_t = KYMAJOR ? t - tap_of_t + vt_halo_step_of_tap(tap_of_t) : t;
#ifdef W_WINDOW_EXPERIMENT       // timing experiment only (wrong results): every K-step's weights from a window of W_WINDOW_EXPERIMENT tiles
        const bf16_t* wt = a.Wp + (long long)(src_t % W_WINDOW_EXPERIMENT) * wstep + opaque(wsrc0);
#else
        const bf16_t* wt = a.Wp + (long long)src_t * wstep + opaque(wsrc0);
#endif
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            if (WPCS % NLD == 0 || j * NLD + wave < WPCS)
#ifndef GNIL_NO_DMA
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NLD * 16 * 32), VT_LDS_PTR(dst + (j * NLD + wave) * 1024), 16, 0, 0);
#else
                asm volatile("" :: "v"(wt));
#endif
    };
    // register-staged row: 8 channels of one halo pixel (XT 1: 2 x 16 B of fp32, XT 2: 16 B of bf16)
    auto load_row = [&](int j, int chunk, f32x4& r0, f32x4& r1) {
        const int xs_j = opaque(xsrc[j]);
        const int off = xs_j >= 0 ? xs_j + chunk * 32 : 0;
        if constexpr (XT == 1) {
            const float* src = xs_j >= 0 ? Xf + off : (const float*)a.zeros;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r0) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r1) : "v"(src) : "memory");
        } else {
            const bf16_t* src = xs_j >= 0 ? Xb + off : (const bf16_t*)a.zeros;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r0) : "v"(src) : "memory");
        }
    };
    // y = silu(x * scale + shift) -> bf16, written where the DMA would have put the raw row
    auto write_row = [&](int j, int chunk, const f32x4& r0, const f32x4& r1) {
        const float* sp = ssl + (chunk * 32 + opaque(dchunk) * 8) * 2;
        char* dst = xbase + (chunk & 1) * XBUF + (j * NLD + wave) * 1024 + opaque(lane) * 16;
        const bool pad = xsrc[j] < 0;                              // conv zero padding applies AFTER norm + SiLU
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float v[4];
            if constexpr (XT == 1) {
                const f32x4 r = hh ? r1 : r0;
                v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
            } else {
                const bf16x8 h = __builtin_bit_cast(bf16x8, r0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (float)h[hh * 4 + i];
            }
            const f32x4 q0 = *(const f32x4*)(sp + hh * 8), q1 = *(const f32x4*)(sp + hh * 8 + 4);   // (scale, shift) x 4 ch
            bf16x4 o;
            o[0] = (bf16_t)vt_silu(fmaf(v[0], q0[0], q0[1]));
            o[1] = (bf16_t)vt_silu(fmaf(v[1], q0[2], q0[3]));
            o[2] = (bf16_t)vt_silu(fmaf(v[2], q1[0], q1[1]));
            o[3] = (bf16_t)vt_silu(fmaf(v[3], q1[2], q1[3]));
            if (pad) o = bf16x4{0, 0, 0, 0};
            *(bf16x4*)(dst + hh * 8) = o;
        }
    };

    // ---- prologue: X(0), then W(0..NW-2).  Plain-input tiles issue their DMA before anything else is set up: the
    // fragment addresses and the bias loads that initialise the accumulators run under the DMA latency (~2 us per tile).
    // SPF (software-pipelined fragments, XT == 0): the MFMA operands of step t+1 are read from LDS DURING step t
    // (each X fragment is refilled as soon as its last MFMA has issued; the W fragments, live until the last MFMA,
    // are refilled at the end of the step and fly during the barrier wait), so after a barrier the matrix pipe
    // starts at once instead of waiting for 12 ds_read_b128.  That needs W(t+1) landed at the barrier of step t.
    constexpr bool SPF = XT == 0;
    // W(t + LEAD) is issued during step t into the stage of W(t-1), whose fragments were read during step t-2: TWO
    // barriers lie between a stage's last ds_read and the DMA that overwrites it.  (With one barrier -- LEAD = NW --
    // a ds_read issued before the barrier but still queued in the LDS pipe lost against the returning DMA about once
    // per 10^5 tiles at two workgroups per CU: s_barrier does not wait for lgkmcnt.)
    constexpr int LEAD = NW - 1;
    constexpr int WOUT = SPF ? LEAD - 2 : NW - 2;  // W tiles issued after the one a barrier needs
    const int fr = lane & 15, fq = lane >> 4;
    // (the bias loads are the oldest vector-memory operations of the wave: every counted wait below covers them)
    f32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bv[i] = a.bias ? *(const f32x4*)(a.bias + c0 + wc * 64 + 16 * fq + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");
    if constexpr (XT == 0) {
        issue_x_dma(0);
#pragma unroll
        for (int t = 0; t < LEAD; ++t)
            if (t < nk) issue_w(t, t % 9);
        asm volatile("" ::: "memory");
    }
    // ---- fragment addressing
    // W fragment i: row wc*64 + i*16 + fr, swz = ((fr >> 2) & 1) << 1
    const int wfoff = (wc * 64 + fr) * HB + ((fq ^ (((fr >> 2) & 1) << 1)) << 4);
    // X fragment j at tap (dy,dx): halo row hr = R + fr, R = wp*8*18 + (j+dy)*18 + dx (wave-uniform).
    // The swizzle bit of row hr is bit 2 of (R + fr), a function of (R & 7, fr) only, and R & 7 =
    // ((j+dy)*18 + dx) & 7 is a compile-time constant (wp*144 is a multiple of 8).  So 8 per-lane base
    // addresses (one per value of R & 7) plus ds_read immediate offsets cover all 72 (tap, j) reads.
    int xsel[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        xsel[k] = (wp * TPW * HWID + fr) * HB + ((fq ^ ((((k + fr) >> 2) & 1) << 1)) << 4);
    auto xaddr = [&](int k) -> int { return xsel[k]; };        // k is a compile-time constant after unrolling
    // accumulators start at the bias (cout map: tile i / register r of lane (fq, fr) = cout c0 + wc*64 + 16*fq + 4*i + r):
    // the epilogue neither adds it nor holds it in registers (its VALU time is exposed: 2 waves per SIMD, nothing to hide under)
    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = bv[i];
    }

    if constexpr (XT != 0) {
        const float* ssg = a.scale_shift + (long long)b * a.Cin * 2;
        for (int i = threadIdx.x; i < a.Cin * 2; i += NT) ssl[i] = ssg[i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NXW; ++j) {
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
            load_row(j, 0, r0, r1);
            if constexpr (XT == 1) wait_vmcnt_tied(0, r0, r1);
            else wait_vmcnt_tied1(0, r0);
            write_row(j, 0, r0, r1);
        }
#pragma unroll
        for (int t = 0; t < LEAD; ++t)
            if (t < nk) issue_w(t, t % 9);
    }

    // SPF operands.  K-steps run dx-major (step p of a chunk: dx = p / 3, dy = p % 3; the weights are packed in that
    // order), so the TP+2 halo rows a wave needs for one dx serve all three dy taps from REGISTERS: xr[r] = halo row r
    // of the wave at column shift dx, and the MFMA for output row j at tap dy reads xr[j + dy].  LDS->register traffic
    // for X falls from 9*TP to 3*(TP+2) fragments per chunk (72 -> 30 at TP = 8).  Row r is refilled with the NEXT
    // group's row r as soon as its last reader has issued: row 0 after (dy 0, j 0), row 1 after (dy 1, j 0), row r >= 2
    // after (dy 2, j = r-2).  The 128-VGPR variant (TPW = 4) has no room for the two extra fragments: it keeps one
    // fragment per output row and re-reads it for every tap.
    constexpr bool DYR = TPW >= 8;
    bf16x8 wfc[TC], xr[DYR ? TP + 2 : TP];
    STAMP(1);
    if constexpr (SPF) {
        int ahead0 = nk - 1;
        if (ahead0 > LEAD - 1) ahead0 = LEAD - 1;
        wait_vmcnt(ahead0 * wpw);                // W(0) and X(0) landed (this wave's pieces) ...
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();            // ... and everybody else's
        asm volatile("" ::: "memory");
        const char* ws0 = wbase + opaque(wfoff);
#pragma unroll
        for (int i = 0; i < TC; ++i) wfc[i] = *(const bf16x8*)(ws0 + i * 16 * HB);
#pragma unroll
        for (int r = 0; r < (DYR ? TP + 2 : TP); ++r) {
            const int rel = r * HWID;            // step 0: dx = dy = 0
            xr[r] = *(const bf16x8*)(xbase + xaddr(rel & 7) + rel * HB);
        }
    }
    STAMP(2);
    STAMP_CLK(11);

    // One chunk = 9 K-steps (taps).  LAST = the final chunk: no next halo, weight ring drains, so the
    // wait count is computed at run time; every other chunk's counts fold to immediates after unrolling.
    // VM-op issue order per wave and step: [wait][barrier] W(t+LEAD) [then the next chunk's halo: XT == 0 at tap 0
    // by DMA; XT != 0 row `tap` at the end of the step].  All VM ops retire in order, so "at most N outstanding"
    // with N = number of ops ISSUED after op X  <=>  X has landed.
    auto do_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* xs = xbase + (chunk & 1) * XBUF;
        const int tbase = chunk * 9;
        f32x4 rq[DLY][XT == 1 ? 2 : 1];           // staged rows in flight (static indices after unrolling)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t = tbase + tap;
            // the operand needed at this barrier: W(t) -- or W(t+1) with SPF, whose fragments are read during
            // this step.  Ops issued after it: the next NW-2 weight tiles (+ the next halo inside its window).
            if constexpr (!LAST) {
                int n = WOUT * wpw;
                if (XT == 0) { if (tap >= 1 && tap <= LEAD - 1) n += NXW; }               // next halo, DMA'd at tap 0
                else { for (int r = tap - (NW - 1); r <= tap - 1; ++r) if (r >= 0 && r < NXW) n += LX; }
                wait_vmcnt(n);
            } else {
                int ahead = 8 - tap - (SPF ? 1 : 0);       // nk - 1 - t (- 1): the last chunk ends the K-steps (nk = 9 * nchunk)
                if (ahead > WOUT) ahead = WOUT;
                if (ahead < 0) ahead = 0;
                wait_vmcnt(ahead * wpw);
            }
            if (XT != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's staged rows are written
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();        // all waves' pieces of the tile (and X(chunk)) are in LDS;
            asm volatile("" ::: "memory");       // everyone is done reading the stage that is refilled next
            if constexpr (!SPF) {
                if (!LAST || tap + LEAD < 9) issue_w(t + LEAD, (tap + LEAD) % 9);
            }

            if constexpr (SPF && !DYR) {
                // MFMAs of step t on registers filled during step t-1; meanwhile fetch step t+1's fragments
                const bool has_next = !LAST || tap < 8;
                const int tap_n = (tap + 1) % 9;
                const int dy_n = tap_n / 3, dx_n = tap_n % 3;            // ky-major walk (KYMAJOR)
                const char* xs_n = (tap == 8) ? xbase + ((chunk + 1) & 1) * XBUF : xs;
                const char* ws_n = wbase + ((t + 1) % NW) * WBUF + opaque(wfoff);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
#pragma unroll
                    for (int i = 0; i < TC; ++i)
                        acc[i][j] = halo_mfma<F16>(wfc[i], xr[j], acc[i][j]);
                    if (has_next) {
                        const int rel = (j + dy_n) * HWID + dx_n;
                        xr[j] = *(const bf16x8*)(xs_n + xaddr(rel & 7) + rel * HB);     // its last reader has issued
                    }
                    if (j == TP / 2 - 1) {
                        if (!LAST || tap + LEAD < 9) issue_w(t + LEAD, (tap + LEAD) % 9);
                        if constexpr (!LAST) { if (tap == 0) issue_x_dma(chunk + 1); }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if (has_next) {
#pragma unroll
                    for (int i = 0; i < TC; ++i) wfc[i] = *(const bf16x8*)(ws_n + i * 16 * HB);
                }
                continue;
            }
            if constexpr (SPF) {
                // MFMAs of step t on registers filled earlier; meanwhile fetch the next group's halo rows / next step's W
                const int dx = tap / 3, dy = tap % 3;
                const bool has_next = !LAST || tap < 8;                 // a next K-step exists (W fragments)
                const bool next_group = !LAST || dx < 2;                   // a next dx group exists (X fragments)
                const int dx_n = (dx + 1) % 3;
                const char* xs_n = (dx == 2) ? xbase + ((chunk + 1) & 1) * XBUF : xs;
                const char* ws_n = wbase + ((t + 1) % NW) * WBUF + opaque(wfoff);
                auto refill = [&](int r) {                                 // DYR: halo row r of the next dx group
#ifndef EXP_NO_FRAG_READS        // timing experiment only (wrong results): the step-0 fragments serve every K-step
                    const int rel = r * HWID + dx_n;
                    xr[r] = *(const bf16x8*)(xs_n + xaddr(rel & 7) + rel * HB);
#endif
                };
                // An LDS-DMA piece costs its wave ~100 issue cycles.  Issued right after the barrier by all 8 waves
                // at once, that kept the matrix pipe idle (~190 cycles per K-step, in-kernel stamps); instead each
                // wave slips its pieces into the MIDDLE of its MFMA sequence.  The older wave of a SIMD pair wins MFMA
                // arbitration and reaches that point ~200 cycles before its partner, so one wave's DMA issue runs
                // beside the other's MFMAs.
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
#pragma unroll
                    for (int i = 0; i < TC; ++i)
                        acc[i][j] = halo_mfma<F16>(wfc[i], xr[j + dy], acc[i][j]);
                    if (next_group) {                                      // rows whose last reader has just issued
                        if (dy == 0 && j == 0) refill(0);
                        if (dy == 1 && j == 0) refill(1);
                        if (dy == 2) refill(j + 2);
                    }
                    if (j == TP / 2 - 1) {
                        if (!LAST || tap + LEAD < 9) issue_w(t + LEAD, (tap + LEAD) % 9);
                        if constexpr (!LAST) { if (tap == 0) issue_x_dma(chunk + 1); }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
#ifndef EXP_NO_FRAG_READS
                if (has_next) {
                    // W fragments are live until the last MFMA: refill them now; the reads fly during the barrier wait
#pragma unroll
                    for (int i = 0; i < TC; ++i) wfc[i] = *(const bf16x8*)(ws_n + i * 16 * HB);
                }
#endif
                continue;
            }
            const char* ws = wbase + (t % NW) * WBUF + opaque(wfoff);     // stage bases beyond 64 KB cannot be ds_read immediates
            const int dx = tap / 3, dy = tap % 3;          // dx-major K-step order (see the weight packers)
            bf16x8 wf[TC];
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *(const bf16x8*)(ws + i * 16 * HB);

            // XT != 0: the row loaded DLY steps ago is normalised + SiLU'ed + written to LDS during THIS step.
            // Both waves of a SIMD leave the barrier together, so if both did this VALU work at the same point the
            // matrix pipe would idle meanwhile.  Waves w and w+4 share a SIMD: the first half of the workgroup
            // stages BEFORE its MFMAs, the second half AFTER, so each wave's VALU work runs beside its partner's MFMAs.
            constexpr bool STAGE = !LAST && XT != 0;
            const int r = tap - DLY;
            auto stage_row = [&]() {
                int n = DLY * wpw;                // W issued in steps r+1 .. tap (this step's W is already out)
                for (int q = r + 1; q < tap; ++q) if (q < NXW) n += LX;          // rows r+1 .. tap-1
                if constexpr (XT == 1) wait_vmcnt_tied(n, rq[r % DLY][0], rq[r % DLY][1]);
                else wait_vmcnt_tied1(n, rq[r % DLY][0]);
                write_row(r, chunk + 1, rq[r % DLY][0], rq[r % DLY][XT == 1 ? 1 : 0]);
            };
            if constexpr (STAGE) {
                if (r >= 0 && r < NXW) {
                    if (early) stage_row();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int jh = 0; jh < TP; jh += 4) {                          // two halves: 16 live X-fragment registers
                bf16x8 xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rel = (jh + j + dy) * HWID + dx;            // compile-time after unrolling
                    xf[j] = *(const bf16x8*)(xs + xaddr(rel & 7) + rel * HB);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TC; ++i)
                        acc[i][jh + j] = halo_mfma<F16>(wf[i], xf[j], acc[i][jh + j]);
            }
            __builtin_amdgcn_s_setprio(0);
            if constexpr (STAGE) {
                if (r >= 0 && r < NXW) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!early) stage_row();
                }
                // load row `tap` of the next chunk into the slot just freed.  VM issue order stays
                // W(t+NW-1) -> row `tap` -> W(t+NW).
                if (tap < NXW) load_row(tap, chunk + 1, rq[tap % DLY][0], rq[tap % DLY][XT == 1 ? 1 : 0]);
            }
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) do_chunk(chunk, std::false_type{});
    do_chunk(nchunk - 1, std::true_type{});
    if constexpr (XT == 0 && TPW >= 8) {                 // (the 128-VGPR tile has no registers to spare: the launcher avoids it)
        if (a.scX) {
            // ---- fused 1x1 shortcut: acc += scW . scX at the centre tap, scCin / 32 plain (un-pipelined) K-steps.  A block
            // whose channel count changes used to pay a separate GEMM launch writing shortcut(x) and this epilogue reading
            // it back as the residual; here the ~2 us of exposed DMA latency per step run beside the other workgroup's MFMAs.
            const bf16_t* Xs = a.scX + (long long)b * a.H * a.W * a.scCin;
            const int scn = a.scCin >> 5;
#pragma nounroll
            for (int c = 0; c < scn; ++c) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads have returned ...
                __builtin_amdgcn_s_barrier();                        // ... and everybody else's: the buffers are free
                asm volatile("" ::: "memory");
                {
                    int hy = opaque(hy0), hx = hx0, hr = hr0;
#pragma nounroll
                    for (int j = 0; j < NXW; ++j) {
                        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
                        const bool v = hr < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                        const void* src = v ? (const void*)(Xs + ((iy * a.W + ix) * a.scCin + dchunk * 8 + c * 32)) : a.zeros;
                        __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(xbase + (j * NLD + wave) * 1024), 16, 0, 0);
                        hr += NLD * 16;
                        hx += (NLD * 16) % HWID; hy += (NLD * 16) / HWID;
                        if (hx >= HWID) { hx -= HWID; ++hy; }
                    }
                    const bf16_t* wt = a.scW + (long long)c * wstep + opaque(wsrc0);
#pragma unroll
                    for (int j = 0; j < WPW; ++j)
                        if (WPCS % NLD == 0 || j * NLD + wave < WPCS)
                            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NLD * 16 * 32), VT_LDS_PTR(wbase + (j * NLD + wave) * 1024), 16, 0, 0);
                }
                wait_vmcnt(0);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char* ws = wbase + opaque(wfoff);
                bf16x8 wf[TC];
#pragma unroll
                for (int i = 0; i < TC; ++i) wf[i] = *(const bf16x8*)(ws + i * 16 * HB);
#pragma unroll
                for (int jh = 0; jh < TP; jh += 4) {
                    bf16x8 xf[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rel = (jh + j + 1) * HWID + 1;              // centre tap
                        xf[j] = *(const bf16x8*)(xbase + xaddr(rel & 7) + rel * HB);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < TC; ++i)
                            acc[i][jh + j] = halo_mfma<F16>(wf[i], xf[j], acc[i][jh + j]);
                }
            }
        }
    }
    STAMP_CLK(12);
    STAMP(3);
#ifdef HALO_STAMP
    asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[TC - 1][TP - 1]));      // the last MFMA results have landed
    STAMP(8);
#endif

    // ---- epilogue.  Weight rows are packed so that MFMA tile i / accumulator register r of lane (fq, fr) is
    // cout cw + 16*fq + 4*i + r: a lane owns 16 CONSECUTIVE couts of pixel (y, x = tx0 + fr) -> 16-B stores, and the
    // four fq lanes of a pixel cover one full 128-B (bf16) / 256-B (fp32) line per wave.
    const int HWp = a.H * a.W;
    const long long ob = (long long)b * HWp * a.Cout;
    const int x = tx0 + fr;
    const int cw = c0 + wc * 64 + 16 * fq;               // first of this lane's 16 couts
    // fp16 residual: fetched with ONE LDS-DMA burst per wave (its own TPW rows x 16 px x 64 couts, into the staging buffers the
    // main loop no longer needs) instead of TPW dependent global-load round trips inside the store loop (conv2 layers: -3 % on
    // the fp8 kernel, where the same change was measured first).  Piece p of a wave = pixels 8 p .. 8 p + 7 of its region x
    // 128 B; lane l -> pixel (l >> 3), physical 16-B chunk (l & 7), logical chunk = physical ^ (pixel & 7).
    constexpr int RPCS = TPW * 2;                        // 1-KB pieces per wave
    constexpr bool RES_DMA = NWV * RPCS * 1024 <= 2 * XBUF + NW * WBUF;
    char* const rbuf = smem + wave * (RPCS * 1024);
    if (RES_DMA && a.res_f16) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                 // every wave has read its last fragments: the buffers are free
#pragma unroll 4
        for (int pc = 0; pc < RPCS; ++pc) {
            const int pp = pc * 8 + (lane >> 3);
            const int yy = ty0 + wp * TPW + (pp >> 4), xx = tx0 + (pp & 15);
            const int lc = (lane & 7) ^ (pp & 7);
            const void* src = (yy < a.H && xx < a.W)
                ? (const void*)(a.res_f16 + ob + ((long long)yy * a.W + xx) * a.Cout + c0 + wc * 64 + lc * 8) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(rbuf + pc * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's residual tile is in LDS (only this wave reads it)
    }
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int y = ty0 + wp * TPW + j;
        if (y >= a.H || x >= a.W) continue;
        valid |= 1u << j;
        const long long o = ob + ((long long)y * a.W + x) * a.Cout + cw;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            f32x4 v = acc[i][j];
            if (a.res) v += *(const f32x4*)(a.res + o + 4 * i);
            if (a.out_f32) *(f32x4*)(a.out_f32 + o + 4 * i) = v;
            acc[i][j] = v;
        }
        if (a.res_f16) {
            // fp16 residual stream: the lane's 16 couts are 32 contiguous bytes
            const int pp = j * 16 + fr;
#pragma unroll
            for (int i = 0; i < TC; i += 2) {
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                const f16x8 rh = RES_DMA ? *(const f16x8*)(rbuf + pp * 128 + (((2 * fq + (i >> 1)) ^ (pp & 7)) << 4))
                                         : *(const f16x8*)(a.res_f16 + o + 4 * i);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][j][r] += (float)rh[r]; acc[i + 1][j][r] += (float)rh[4 + r]; }
            }
        }
        if (a.out_f16) {
#pragma unroll
            for (int i = 0; i < TC; i += 2) {
                typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                f16x8 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j][r]; h[4 + r] = (f16_t)acc[i + 1][j][r]; }
                *(f16x8*)(a.out_f16 + o + 4 * i) = h;
            }
        }
        if (a.out_bf16) {
            // a.out16_planar: the 16-bit copy is laid out [Cout/32][H][W][32] (chunk-planar) for the stride-2 phase-plane kernel that reads it
            const long long o16 = a.out16_planar ? (((long long)(b * (a.Cout >> 5) + (cw >> 5)) * a.H + y) * a.W + x) * 32 + (cw & 31) : o;
#pragma unroll
            for (int i = 0; i < TC; i += 2) {
                if (a.out16_f16) {                             // the operand copy carries fp16 bits when its consumer runs on fp16 operands
                    typedef _Float16 f16x8o __attribute__((ext_vector_type(8)));
                    f16x8o h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j][r]; h[4 + r] = (f16_t)acc[i + 1][j][r]; }
                    *(f16x8o*)((f16_t*)a.out_bf16 + o16 + 4 * i) = h;
                } else {
                bf16x8 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) { h[r] = (bf16_t)acc[i][j][r]; h[4 + r] = (bf16_t)acc[i + 1][j][r]; }
#ifdef EPI_NOSTORE
                asm volatile("" :: "v"(h));
#else
                *(bf16x8*)(a.out_bf16 + o16 + 4 * i) = h;
#endif
                }
            }
        }
    }
    STAMP(4);
    if (a.gn_partial) {
        // GroupNorm statistics of this tile's outputs for the NEXT layer's norm (replaces a full read pass)
        __syncthreads();                                   // every wave is done with the staging LDS
        STAMP(10);
        const int G = a.Cout / a.gn_cpg;
        float* out = a.gn_partial + (((long long)b * a.ptiles + tile) * G + c0 / a.gn_cpg) * 3;
        vt_gn_epilogue_partials_il<TC, TP>(acc, valid, a.gn_cpg, wp, WP, wc * 64, BC, (float*)smem, out);
    }
#ifdef HALO_STAMP
    STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    if (g_halo_stamps && threadIdx.x == STAMP_TID) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_halo_stamps[(long long)blockIdx.x * 16 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

template <int WP, int WC, int XT, int TPW, int NW = NW_DEFAULT, bool F16 = false>
hipError_t launch(const Conv3x3Args& a, hipStream_t s) {
    constexpr int ROWS = WP * TPW, BC = WC * 64, NWV = WP * WC;
    constexpr int HROWS = (ROWS + 2) * HWID;
    constexpr int NXW = ((HROWS + 15) / 16 + NWV - 1) / NWV;
    const int smem = 2 * NXW * NWV * 16 * HB + NW * BC * HB + (XT ? a.Cin * 8 : 0);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    auto kern = conv3x3_halo_kernel<WP, WC, XT, TPW, NW, F16>;
    hipError_t ea = vt_once_per_device(attr_done, [&] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (ea != hipSuccess) return ea;
    const long long tiles = (long long)((a.W + TW - 1) / TW) * ((a.H + ROWS - 1) / ROWS);
    const long long nblk = tiles * (a.Cout / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    Conv3x3Args k = a;
    k.tiles_x = (a.W + TW - 1) / TW; k.ctiles = a.Cout / BC; k.per_img = (int)(tiles * k.ctiles); k.ptiles = (int)tiles;
    auto magic = [&](long long d) -> unsigned long long {           // n / d = (n * m) >> 40 for every n < nblk, when n * d < 2^40
        return (nblk * d < (1LL << 40) && nblk < (1LL << 23)) ? ((1ULL << 40) / (unsigned long long)d + 1ULL) : 0ULL;
    };
    k.m_per_img = magic(k.per_img); k.m_ctiles = magic(k.ctiles); k.m_tiles_x = magic(k.tiles_x);
#ifdef HALO_STAMP
    if (s_stamp_nth >= 0) {
        const bool on = a.H == s_stamp_H && a.Cin == s_stamp_Cin && s_stamp_seen++ == s_stamp_nth;
        if (on != s_stamp_on) {
            (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_halo_stamps), &s_stamp_ptrs[on ? 1 : 0], sizeof(void*), 0, hipMemcpyHostToDevice, s);
            s_stamp_on = on;
        }
    }
#endif
#ifdef GNIL_ONE_WG           // experiment: more LDS than half a CU has, so that workgroups never share a CU
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NWV), smem > 90 * 1024 ? smem : 90 * 1024, s, k);
#else
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NWV), smem, s, k);
#endif
    return hipGetLastError();
}

// [Cout][9][Cin] (the generic kernel's layout) -> Wp[Cin/32][9][Cout][32]; device-side, for the op-level entry
__global__ void repack_ohwi_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int Cin, int Cout) {
    const long long n = (long long)Cout * 9 * Cin;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 9);
    const int co = (int)(i / ((long long)Cin * 9));
    const int row = (co & ~63) + vt_halo_row_of_cout(co & 63);      // interleaved cout map (see the kernel epilogue)
    wp[(((long long)(ci >> 5) * 9 + vt_halo_step_of_tap(tap)) * Cout + row) * 32 + (ci & 31)] = w[i];
}

}  // namespace

hipError_t vt_launch_repack_ohwi_to_halo(const bf16_t* w, bf16_t* wp, int Cin, int Cout, hipStream_t s) {
    const long long n = (long long)Cout * 9 * Cin;
    hipLaunchKernelGGL(repack_ohwi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cin, Cout);
    return hipGetLastError();
}

// Two-workgroups-per-CU tiles for plain-input layers (Conv3x3Args::occ2 = vt_set_flag(ctx, 3, mode)): while one workgroup runs
// its prologue / epilogue (~12 us of VALU, latency and stores per tile) the other one has the matrix pipes.
//   0 off; 1 = 128-cout layers on 8 waves x 128 VGPRs (16x16 px x 128 couts, wave tile 64 px x 64 couts);
//   2 = 128-cout layers on 4 waves x 256 VGPRs (same tile, wave tile 128 px x 64 couts, halo rows reused across ky);
//   3 = mode 2 for EVERY plain-input layer (256-cout layers run two 128-cout tiles per pixel tile).
// ONE function decides the kernel variant; the launcher, the profile slot and the GroupNorm-partials count all use it.
namespace {
enum HaloVariant { HV_2208_4 = 0, HV_4204_4, HV_4208_6, HV_2408_6, HV_XT1_128, HV_XT1_256, HV_XT2_128, HV_XT2_256, HV_2216_6 };
HaloVariant halo_variant(int Cout, int xt, int occ2, bool has_sc) {
    const bool big = (Cout % 256) == 0;                     // 16x16 px x 256 couts, else 32x16 px x 128 couts
    if (xt == 1) return big ? HV_XT1_256 : HV_XT1_128;
    if (xt == 2) return big ? HV_XT2_256 : HV_XT2_128;
    if (occ2 == 4) return HV_2216_6;                        // round-4 experiment: one wave per SIMD, 32 x 16 px x 128 couts per workgroup
    if (occ2 == 3 || (occ2 == 2 && !big)) return HV_2208_4;                     // 4 waves, 2 workgroups / CU
    if (occ2 && !big && !has_sc) return HV_4204_4;          // (the 128-VGPR tile has no registers for the fused shortcut)
    return big ? HV_2408_6 : HV_4208_6;
}
int halo_variant_rows(HaloVariant v) {
    switch (v) {
        case HV_4208_6: case HV_XT1_128: case HV_XT2_128: case HV_2216_6: return 32;
        default: return 16;
    }
}
}  // namespace

int vt_conv3x3_halo_tiles(int H, int W, int Cout, int xt, int occ2, int has_sc) {
    const int rows = halo_variant_rows(halo_variant(Cout, xt, occ2, has_sc != 0));
    return ((W + TW - 1) / TW) * ((H + rows - 1) / rows);
}
int vt_conv3x3_halo_tiles_max(int H, int W) { return ((W + TW - 1) / TW) * ((H + 15) / 16); }

bool vt_conv3x3_halo_supported(int Cin, int Cout) { return Cin >= 32 && (Cin % 32) == 0 && (Cout % 128) == 0; }
// fp16-operand form (Conv3x3Args::f16): compiled for the default two-workgroups-per-CU tile on plain input
bool vt_conv3x3_halo_f16_supported(int Cout, int occ2, int has_sc) { return halo_variant(Cout, 0, occ2, has_sc != 0) == HV_2208_4; }

static int halo_xt(const Conv3x3Args& a) { return a.scale_shift ? (a.Xf32 ? 1 : 2) : 0; }

int vt_conv3x3_halo_config(const Conv3x3Args& a) {      // profile slots 3..8
    switch (halo_variant(a.Cout, halo_xt(a), a.occ2, a.scX != nullptr)) {
        case HV_2208_4: return 3;
        case HV_4204_4: return 4;
        case HV_4208_6: return 5;
        case HV_2408_6: return 6;
        case HV_2216_6: return 5;                           // (shares the one-workgroup-per-CU 128-cout slot)
        case HV_XT1_128: case HV_XT1_256: return 7;
        default: return 8;
    }
}

hipError_t vt_launch_conv3x3_halo(const Conv3x3Args& a, hipStream_t s) {
    if (!a.Wp || !a.zeros || (!a.out_f32 && !a.out_bf16 && !a.out_f16)) return hipErrorInvalidValue;
    if ((a.res && a.res_f16) || (a.res_f16 && a.out_f32)) return hipErrorInvalidValue;
    if (!vt_conv3x3_halo_supported(a.Cin, a.Cout) || a.batch <= 0 || a.H <= 0 || a.W <= 0) return hipErrorInvalidValue;
    if (a.gn_partial && a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;        // 32-bit per-image offsets
    if ((a.scX != nullptr) != (a.scW != nullptr)) return hipErrorInvalidValue;
    if (a.scX && (a.scCin <= 0 || (a.scCin % 32) || a.scale_shift || (long long)a.H * a.W * a.scCin >= (1LL << 31))) return hipErrorInvalidValue;
    if ((long long)(a.Cin / 32) * 9 * a.Cout * 32 >= (1LL << 31)) return hipErrorInvalidValue;
    if (a.occ2 < 0 || a.occ2 > 4) return hipErrorInvalidValue;
    // input mode: raw bf16 (X), or GroupNorm+SiLU fused into the staging of an fp32 (Xf32) / bf16 (X) tensor
    const int xt = halo_xt(a);
    if (xt == 1 ? (a.X != nullptr) : (a.X == nullptr || a.Xf32 != nullptr)) return hipErrorInvalidValue;
    if (a.f16 && (xt != 0 || halo_variant(a.Cout, xt, a.occ2, a.scX != nullptr) != HV_2208_4)) return hipErrorInvalidValue;   // fp16 operands: the default tile only
    if (a.f16) return launch<2, 2, 0, 8, 4, true>(a, s);
    switch (halo_variant(a.Cout, xt, a.occ2, a.scX != nullptr)) {
        case HV_2208_4: return launch<2, 2, 0, 8, 4>(a, s);
        case HV_4204_4: return launch<4, 2, 0, 4, 4>(a, s);
        case HV_4208_6: return launch<4, 2, 0, 8>(a, s);
        case HV_2408_6: return launch<2, 4, 0, 8>(a, s);
        case HV_2216_6: return launch<2, 2, 0, 16, 6>(a, s);
        case HV_XT1_128: return launch<4, 2, 1, 8>(a, s);
        case HV_XT1_256: return launch<2, 4, 1, 8>(a, s);
        case HV_XT2_128: return launch<4, 2, 2, 8>(a, s);
        default: return launch<2, 4, 2, 8>(a, s);
    }
}
