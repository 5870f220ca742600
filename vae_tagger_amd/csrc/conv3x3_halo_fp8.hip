// 3x3 / stride 1 / pad 1 convolution on fp8 (OCP e4m3) operands -- BASELINE.json configs[4], opt-in (vt_set_flag(ctx, 11, 1)).
//
// Same halo-tile implicit GEMM as conv3x3_halo.hip (X halo staged once per channel chunk and read at shifted rows by all nine
// taps, weights through an LDS ring filled by LDS-DMA under counted vmcnt waits, one raw s_barrier per K-step, fragments of
// step t+1 read during step t, halo rows reused across ky from registers), re-shaped for
// v_mfma_scale_f32_32x32x64_f8f6f4 (both scales 2^0): twice the bf16 FLOPs per matrix-pipe cycle.
//   * an LDS row is still 64 B, now 64 fp8 channels: a K-step = (64-channel chunk, tap) carries twice the K of the bf16
//     kernel at the same LDS / DMA bytes, so every data rate per unit time stays where the bf16 kernel has it;
//   * workgroup = 4 waves (2 pixel-row groups x 2 cout groups) x <= 256 VGPRs, 80 KB LDS: two workgroups per CU;
//     tile = 8 rows x 32 pixels x 128 couts, wave tile = 4 rows x 32 pixels x 64 couts = 4 x 2 MFMAs of 32x32x64;
//   * operand fragment = 32 B per lane: lane (g = lane >> 5, i = lane & 31) holds channels 32 g .. 32 g + 31 of row i (weights:
//     cout row, activations: pixel), read as two ds_read_b128; rows 64 B apart are conflict-free with the 16-B chunk
//     swizzle  physical = logical ^ ((row ^ (row >> 2)) & 3), applied to the DMA's per-lane source address and to the reads;
//   * cout rows are packed so that accumulator register r of lane (g, x) is cout base + 16 g + r: 16 consecutive couts of one
//     pixel per lane (32-B fp16 / 64-B fp32 runs per store pair, GroupNorm groups stay inside a lane).
// Operands: X = e4m3(act_scale * silu(GroupNorm(.))) written by the GroupNorm-apply pass (groupnorm.hip, fp8 output);
// W = e4m3(w / wscale[cout]) packed by the host; the epilogue computes acc * (wscale[cout] / act_scale) + bias.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

#ifdef HALO_STAMP
// Diagnostic build only (tools/stamp_halo_fp8.py): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of one workgroup and
// one s_memtime (shader clock) pair around the main loop; written only for launches that match the filter, into a buffer nothing
// else reads.  In the product build no stamp executes.
__device__ unsigned long long* g_halo8_stamps = nullptr;
__device__ int g_halo8_filter[3] = {0, 0, 0};         // H, Cin, 1 + (fp16 residual present); 0 = any
extern "C" int vt_debug_halo_fp8_stamps(unsigned long long* buf, int H, int Cin, int res) {
    const int f[3] = {H, Cin, res};
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_halo8_filter), f, sizeof(f));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_halo8_stamps), &buf, sizeof(buf));
}
#define STAMP_ON (g_halo8_stamps && threadIdx.x == 0 && (!g_halo8_filter[0] || g_halo8_filter[0] == a.H) && \
                  (!g_halo8_filter[1] || g_halo8_filter[1] == a.Cin) && (!g_halo8_filter[2] || g_halo8_filter[2] == 1 + (a.res_f16 != nullptr)))
#define STAMP(slot)                                                                                            \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (STAMP_ON) g_halo8_stamps[(long long)blockIdx.x * 16 + (slot)] = t_;                                \
    } while (0)
#define STAMP_CLK(slot)                                                                                        \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (STAMP_ON) g_halo8_stamps[(long long)blockIdx.x * 16 + (slot)] = t_;                                \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_CLK(slot) do {} while (0)
#endif

namespace {

constexpr int HB = 64;                       // bytes per LDS row
constexpr int WC = 2, TP = 4;                // cout wave groups (x 64 couts); tile rows per wave
constexpr int BC = WC * 64;                  // 128 couts per workgroup
constexpr int WPCS = BC / 16;                // W pieces per K-step: 8
constexpr int WBUF = BC * HB;                // 8 KB per stage
constexpr int NW = 4;                        // weight ring depth
constexpr int LEAD = NW - 1;                 // W(t + LEAD) is issued during step t (two barriers after the stage's last read)
constexpr int WOUT = LEAD - 2;               // W tiles issued after the one a barrier needs (fragments of t+1 are read during t)

// Tile shape = (WP x TP) rows x (WX x 32) pixels x 128 couts on WP x WX x 2 waves; a wave's tile is always 4 rows x 32 px x 64 couts.
//   <2, 1>: 8 x 32 px, 4 waves, 80 KB LDS, two workgroups per CU (the default);
//   <4, 1>: 16 x 32 px, 8 waves, one workgroup per CU: one staged weight tile serves twice the pixels, halo over-read 1.19x instead of 1.33x;
//   <2, 2>: 8 x 64 px, 8 waves, one workgroup per CU (over-read 1.29x).  vt_set_flag(ctx, 16, shape); round-4 A/B: profiles/r04/halo_fp8_tile_shapes_ab.log
template <int WP_, int WX_>
struct Shape {
    static constexpr int WP = WP_, WX = WX_;
    static constexpr int TWX = 32 * WX;                  // tile width in pixels (the MFMA's N = 32 per wave)
    static constexpr int HWID = TWX + 2;                 // halo width
    static constexpr int ROWS = WP * TP;                 // tile rows
    static constexpr int NWV = WP * WX * WC, NT = 64 * NWV;
    static constexpr int HROWS = (ROWS + 2) * HWID;      // halo pixels
    static constexpr int NXW = ((HROWS + 15) / 16 + NWV - 1) / NWV;      // X pieces (16 rows each) per wave
    static constexpr int XBUF = NXW * NWV * 16 * HB;     // bytes per halo buffer
    static constexpr int WPW = WPCS / NWV;               // W pieces per wave and K-step
    static constexpr int SMEM_LOOP = 2 * XBUF + NW * WBUF;
    static constexpr int SMEM_RES = NWV * 16384 + WBUF;  // epilogue: a 16-KB fp16 residual tile per wave + the GroupNorm scratch behind them
    static constexpr int SMEM = SMEM_LOOP > SMEM_RES ? SMEM_LOOP : SMEM_RES;
    static constexpr int WG_PER_CU = SMEM <= 80 * 1024 ? 2 : 1;
    static_assert(WPCS % NWV == 0 && SMEM <= 160 * 1024, "tile shape");
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 2)) & 3; }

// 32-byte fragment of LDS row `row`: logical chunks 2 g, 2 g + 1
__device__ __forceinline__ i32x8 read_frag(const char* base, int row, int g) {
    const int a = row * HB + (((2 * g) ^ swz(row)) << 4);
    const i32x4 lo = *(const i32x4*)(base + a), hi = *(const i32x4*)(base + (a ^ 16));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int WP, int WX>
__global__ __launch_bounds__(64 * WP * WX * WC, 2) void conv3x3_halo_fp8_kernel(const Conv3x3Fp8Args a) {
    typedef Shape<WP, WX> SH;
    constexpr int TWX = SH::TWX, HWID = SH::HWID, ROWS = SH::ROWS, NWV = SH::NWV, HROWS = SH::HROWS, NXW = SH::NXW, XBUF = SH::XBUF,
                  WPW = SH::WPW, SMEM = SH::SMEM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbase = smem;                    // 2 halo buffers
    char* const wbase = smem + 2 * XBUF;         // NW weight stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave % WC, wx = (wave / WC) % WX, wp = wave / (WC * WX);
    const int g = lane >> 5, li = lane & 31;
    STAMP(0);

    // ---- tile coordinates (launch-constant divisors arrive as 2^40 / d + 1 multipliers)
    auto fdiv = [](int n, unsigned long long m, int d) -> int {
        return m ? (int)(((unsigned long long)(unsigned)n * m) >> 40) : n / d;
    };
    int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int b = fdiv(logical, a.m_per_img, a.per_img);
    logical -= b * a.per_img;
    const int tile = fdiv(logical, a.m_ctiles, a.ctiles);
    const int ct = logical - tile * a.ctiles;
    const int tyi = fdiv(tile, a.m_tiles_x, a.tiles_x);
    const int ty0 = tyi * ROWS, tx0 = (tile - tyi * a.tiles_x) * TWX;
    const int c0 = ct * BC;
    const unsigned char* Xb = a.X + (long long)b * a.H * a.W * a.Cin;
    const int nchunk = a.Cin >> 6;
    const int nk = nchunk * 9;

    // ---- DMA bookkeeping: one wave-instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical chunk (l & 3),
    // logical chunk = physical ^ swz(row) (swz of a piece's row depends on the row inside the piece only: pieces start at multiples of 16)
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ swz(drow);
    const int hr0 = wave * 16 + drow;
    const int hy0 = hr0 / HWID, hx0 = hr0 - (hr0 / HWID) * HWID;
    const int wsrc0 = (c0 + wave * 16 + drow) * HB + dchunk * 16;       // byte offset inside a K-step's [Cout][64] tile
    const int wstep = a.Cout * HB;                                     // bytes between consecutive K-steps

    auto issue_x_dma = [&](int chunk) {
#ifdef EXP_NO_DMA                    // timing experiment only (wrong results): the first two halos and the first NW weight tiles serve every K-step
        if (chunk >= 2) return;
#endif
        char* dst = xbase + (chunk & 1) * XBUF;
        int hy = opaque(hy0), hx = hx0, hr = hr0;
#pragma nounroll
        for (int j = 0; j < NXW; ++j) {
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            const bool v = hr < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const void* src = v ? (const void*)(Xb + ((iy * a.W + ix) * a.Cin + chunk * 64 + dchunk * 16)) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(dst + (j * NWV + wave) * 1024), 16, 0, 0);
            hr += NWV * 16;
            hx += (NWV * 16) % HWID; hy += (NWV * 16) / HWID;
            if (hx >= HWID) { hx -= HWID; ++hy; }
        }
    };
    auto issue_w = [&](int t) {
#ifdef EXP_NO_DMA
        if (t >= NW) return;
#endif
        char* dst = wbase + (t % NW) * WBUF;
#ifdef W_WINDOW_EXPERIMENT       // timing experiment only (wrong results): every K-step's weights from a window of W_WINDOW_EXPERIMENT tiles
        const unsigned char* wt = a.Wp + (long long)(t % W_WINDOW_EXPERIMENT) * wstep + opaque(wsrc0);
#else
        const unsigned char* wt = a.Wp + (long long)t * wstep + opaque(wsrc0);
#endif
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NWV * 16 * HB), VT_LDS_PTR(dst + (j * NWV + wave) * 1024), 16, 0, 0);
    };

    // ---- prologue: DMA first; the epilogue constants load under its latency
    issue_x_dma(0);
#pragma unroll
    for (int t = 0; t < LEAD; ++t)
        if (t < nk) issue_w(t);
    asm volatile("" ::: "memory");
    STAMP(1);

    f32x16 acc[2][TP];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][j][r] = 0.f;

    // fragment rows: W stage row wc*64 + 32 h + li; X halo row (wp*TP + r) * HWID + dx + li
    const int wrow0 = wc * 64 + li;
    const int xrow0 = wp * TP * HWID + wx * 32 + li;
    i32x8 wfc[2], xr[TP + 2];
    {
        int ahead0 = nk - 1;
        if (ahead0 > LEAD - 1) ahead0 = LEAD - 1;
        wait_vmcnt(ahead0 * WPW);                // W(0) and X(0) landed (this wave's pieces) ...
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();            // ... and everybody else's
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) wfc[h] = read_frag(wbase, opaque(wrow0) + 32 * h, g);
#pragma unroll
        for (int r = 0; r < TP + 2; ++r) xr[r] = read_frag(xbase, opaque(xrow0) + r * HWID, g);
    }
    STAMP(2);
    STAMP_CLK(11);

    // One chunk = 9 K-steps (taps, kx-major: step p -> dx = p / 3, dy = p % 3).  VM-op issue order per wave and step:
    // [wait][barrier] ... W(t+LEAD) [+ the next chunk's halo at tap 0], both in the middle of the MFMA sequence.
    auto do_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* xs = xbase + (chunk & 1) * XBUF;
        const int tbase = chunk * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t = tbase + tap;
            if constexpr (!LAST) {
                int n = WOUT * WPW;
                if (tap >= 1 && tap <= LEAD - 1) n += NXW;               // the next halo, DMA'd at tap 0
                wait_vmcnt(n);
            } else {
                // the last chunk ends the kernel's K-steps (nk = 9 * nchunk): t + 1 = nk - (8 - tap), all compile-time
                int ahead = 7 - tap;
                if (ahead > WOUT) ahead = WOUT;
                if (ahead < 0) ahead = 0;
                wait_vmcnt(ahead * WPW);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int dx = tap / 3, dy = tap % 3;
            const bool has_next = !LAST || tap < 8;
            const bool next_group = !LAST || dx < 2;
            const int dx_n = (dx + 1) % 3;
            const char* xs_n = (dx == 2) ? xbase + ((chunk + 1) & 1) * XBUF : xs;
            const char* ws_n = wbase + ((t + 1) % NW) * WBUF;
#ifndef EXP_NO_FRAG_READS            // timing experiment only (wrong results): the step-0 fragments serve every K-step
            auto refill = [&](int r) { xr[r] = read_frag(xs_n, opaque(xrow0) + r * HWID + dx_n, g); };
#else
            auto refill = [&](int) {};
#endif
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wfc[h], xr[j + dy], acc[h][j], 0, 0, 0, 127, 0, 127);
                if (next_group) {                                        // rows whose last reader has just issued
                    if (dy == 0 && j == 0) refill(0);
                    if (dy == 1 && j == 0) refill(1);
                    if (dy == 2) refill(j + 2);
                }
                if (j == TP / 2 - 1) {
                    if (!LAST || tap + LEAD < 9) issue_w(t + LEAD);
                    if constexpr (!LAST) { if (tap == 0) issue_x_dma(chunk + 1); }
                }
            }
            __builtin_amdgcn_s_setprio(0);
#ifndef EXP_NO_FRAG_READS
            if (has_next) {
#pragma unroll
                for (int h = 0; h < 2; ++h) wfc[h] = read_frag(ws_n, opaque(wrow0) + 32 * h, g);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);           // a K-step's MFMAs stay inside it (sunk past later barriers they cost spills)
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) do_chunk(chunk, std::false_type{});
    do_chunk(nchunk - 1, std::true_type{});
    STAMP_CLK(12);
    STAMP(3);
    if (a.scX) {
        // ---- fused 1x1 shortcut (a resnet block's conv_shortcut): acc += scW . scX at the tile's own pixels, as scCin / 32 plain
        // K-steps of bf16 32x32x16 MFMAs on the SAME accumulators (the fp8 and bf16 32x32 forms share the accumulator layout).
        // scW is pre-divided by mult[cout] on the host, so the epilogue's acc * mult + bias scales it back.
        const bf16_t* Xs = a.scX + (long long)b * a.H * a.W * a.scCin;
        const int scn = a.scCin >> 5;
#pragma nounroll
        for (int c = 0; c < scn; ++c) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // the staging buffers are free
            asm volatile("" ::: "memory");
            static_assert(ROWS * TWX / 16 == 4 * NWV, "shortcut staging: 4 pieces per wave");
#pragma unroll
            for (int j = 0; j < 4; ++j) {                        // the tile's ROWS x TWX pixels, 16 per piece, 4 pieces per wave
                const int pp = (j * NWV + wave) * 16 + drow;
                const int iy = ty0 + pp / TWX, ix = tx0 + pp % TWX;
                const void* src = (iy < a.H && ix < a.W) ? (const void*)(Xs + ((long long)(iy * a.W + ix) * a.scCin + c * 32 + dchunk * 8)) : a.zeros;
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(xbase + (j * NWV + wave) * 1024), 16, 0, 0);
            }
            const bf16_t* wt = a.scW + ((long long)c * a.Cout + c0 + wave * 16 + drow) * 32 + dchunk * 8;
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NWV * 16 * 32), VT_LDS_PTR(wbase + (j * NWV + wave) * 1024), 16, 0, 0);
            wait_vmcnt(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                     // k = 16 ks + 8 g .. + 7: logical 16-B chunk 2 ks + g
                bf16x8 wf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = wrow0 + 32 * h;
                    wf[h] = *(const bf16x8*)(wbase + row * HB + (((2 * ks + g) ^ swz(row)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int row = (wp * TP + j) * TWX + wx * 32 + li;
                    const bf16x8 xf = *(const bf16x8*)(xbase + row * HB + (((2 * ks + g) ^ swz(row)) << 4));
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc[h][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[h], xf, acc[h][j], 0, 0, 0);
                }
            }
        }
    }
    // pin the accumulators here: left alone, the compiler sinks each chain's last MFMAs into the epilogue's conditional
    // blocks (behind the first stores), keeps the operand fragments alive for them and spills ~100 registers
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < TP; ++j) asm volatile("" : "+v"(acc[h][j]));
    asm volatile("" ::: "memory");

    // ---- epilogue: register r of lane (g, x = li) in acc[h][j] is cout cw(h) + r of pixel (ty0 + wp*TP + j, tx0 + li)
    const int HWp = a.H * a.W;
    const long long ob = (long long)b * HWp * a.Cout;
    const int x = tx0 + wx * 32 + li;
    // fp16 residual: fetched with ONE LDS-DMA burst per wave (its own 4 rows x 32 px x 64 couts = 16 KB, into the staging
    // buffers the main loop no longer needs) instead of eight dependent global-load round trips inside the store loop.
    // Piece p of a wave = pixels 8 p .. 8 p + 7 of its region x 128 B; lane l -> pixel (l >> 3), physical 16-B chunk (l & 7),
    // logical chunk = physical ^ (pixel & 7) (rows 128 B apart: 2-way instead of 8-way bank conflicts on the reads).
    char* const rbuf = smem + wave * 16384;
    if (a.res_f16) {
        __syncthreads();                                 // every wave has read its last fragments: the buffers are free
#pragma unroll 4
        for (int pc = 0; pc < 16; ++pc) {
            const int pp = pc * 8 + (lane >> 3);
            const int yy = ty0 + wp * TP + (pp >> 5), xx = tx0 + wx * 32 + (pp & 31);
            const int lc = (lane & 7) ^ (pp & 7);
            const void* src = (yy < a.H && xx < a.W)
                ? (const void*)(a.res_f16 + ob + ((long long)yy * a.W + xx) * a.Cout + c0 + wc * 64 + lc * 8) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(rbuf + pc * 1024), 16, 0, 0);
        }
    }
    unsigned valid = 0;
    float amax8 = 0.f;                                  // largest |value| handed to the e4m3 conversion
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cw = c0 + wc * 64 + 32 * h + 16 * g;
        f32x4 mul[4], bia[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mul[i] = *(const f32x4*)(a.mult + cw + 4 * i);
            bia[i] = a.bias ? *(const f32x4*)(a.bias + cw + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (h == 0 && a.res_f16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's residual tile is in LDS
#ifdef HALO_STAMP
        if (h == 0) STAMP(9);
#endif
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int y = ty0 + wp * TP + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][j][r] = fmaf(acc[h][j][r], mul[r >> 2][r & 3], bia[r >> 2][r & 3]);
            if (y >= a.H || x >= a.W) continue;
            valid |= 1u << j;
            const long long o = ob + ((long long)y * a.W + x) * a.Cout + cw;
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            if (a.res) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 rv = *(const f32x4*)(a.res + o + 4 * i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[h][j][4 * i + q] += rv[q];
                }
            }
            if (a.res_f16) {
                const int pp = j * 32 + li;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f16x8 rh = *(const f16x8*)(rbuf + pp * 128 + (((4 * h + 2 * g + i) ^ (pp & 7)) << 4));
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[h][j][8 * i + q] += (float)rh[q];
                }
            }
            if (a.out_f32) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(f32x4*)(a.out_f32 + o + 4 * i) = f32x4{acc[h][j][4 * i], acc[h][j][4 * i + 1], acc[h][j][4 * i + 2], acc[h][j][4 * i + 3]};
            }
            if (a.out_f16) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f16x8 hh;
#pragma unroll
                    for (int q = 0; q < 8; ++q) hh[q] = (f16_t)acc[h][j][8 * i + q];
                    *(f16x8*)(a.out_f16 + o + 8 * i) = hh;
                }
            }
            if (a.out_e4m3) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                i32x4 o8 = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] = acc[h][j][4 * i + q] * a.out_e4m3_scale;
                    amax8 = fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fmaxf(fabsf(t[2]), fabsf(t[3])), amax8));
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] = __builtin_amdgcn_fmed3f(t[q], -448.f, 448.f);
                    o8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], o8[i], false);
                    o8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], o8[i], true);
                }
                // a.out8_planar: [Cout/64][H][W][64] (chunk-planar) for the fp8 stride-2 phase-plane kernel (see conv3x3_s2_halo.hip)
                const long long o8a = a.out8_planar ? (((long long)(b * (a.Cout >> 6) + (cw >> 6)) * a.H + y) * a.W + x) * 64 + (cw & 63) : o;
                *(i32x4*)(a.out_e4m3 + o8a) = o8;
            }
            if (a.out_bf16) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bf16x8 hh;
#pragma unroll
                    for (int q = 0; q < 8; ++q) hh[q] = (bf16_t)acc[h][j][8 * i + q];
                    *(bf16x8*)(a.out_bf16 + o + 8 * i) = hh;
                }
            }
        }
    }
    if (a.out_e4m3 && a.status && amax8 > 448.f) atomicOr(a.status, 2);      // e4m3 saturates silently: sticky bit 1 of vt_status
    STAMP(4);
    if (a.gn_partial) {
        // GroupNorm (n, mean, M2) of this tile's outputs for the next norm.  A group (cpg = 4, 8 or 16 consecutive couts) lives in
        // one lane; sums are taken relative to a per-(wave half, group) pivot, reduced over the wave's 32 pixel columns, then the
        // two pixel-row waves are merged with Chan's formula in a fixed order (deterministic).
        // This phase is latency, not work (its VALU issue is hidden behind the other resident workgroup's MFMAs, but a 128-channel
        // layer's main loop is shorter than its epilogue): every group's pivot, sums and lane reductions are independent chains
        // issued together, cross-lane steps are DPP / readlane (no LDS round trips), one branch writes all partials.
        float* lds = (float*)(smem + SMEM - WBUF);         // the last weight stage: clear of the residual staging (waves x 16 KB from 0)
        if (!a.res_f16 && !a.scX) {                        // those paths have passed a barrier since the last K-step's fragment reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const int cpg = a.gn_cpg;
        const int gpb = BC / cpg;
        float npix = 0.f;
#pragma unroll
        for (int j = 0; j < TP; ++j) npix += (float)__popcll(__ballot((valid >> j) & 1u) & 0xffffffffull);   // lanes 0..31 = the 32 columns
        const float n = npix * (float)cpg;
        const bool full = __ballot(valid != (1u << TP) - 1u) == 0ull;      // the whole wave tile is inside the image
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto stats = [&](auto cpg_tag, auto full_tag) {
            constexpr int CPG = decltype(cpg_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int NQ = 16 / CPG;
            float piv[2][NQ], s[2][NQ], ss[2][NQ];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {             // pivot = the group's first value at the half-wave's first pixel (finite even outside the image)
                    const int v = __builtin_bit_cast(int, acc[h][0][q * CPG]);
                    const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 0));
                    const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 32));
                    piv[h][q] = g ? p1 : p0;
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x2 p2 = {piv[h][q], piv[h][q]};
                    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        if (FULL || ((valid >> j) & 1u)) {
#pragma unroll
                            for (int r = q * CPG; r < (q + 1) * CPG; r += 2) {
                                const f32x2 d = f32x2{acc[h][j][r], acc[h][j][r + 1]} - p2;
                                s2 += d; q2 += d * d;
                            }
                        }
                    }
                    s[h][q] = s2[0] + s2[1]; ss[h][q] = q2[0] + q2[1];
                }
            // 32-column sums: inside each 16-lane row, then row 0 -> row 1 and row 2 -> row 3 (row_bcast15): lanes 16..31 / 48..63 hold them
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float a0 = vt_row16_sum(s[h][q]), b0 = vt_row16_sum(ss[h][q]);
                    s[h][q] = a0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x142, 0xA, 0xF, false));
                    ss[h][q] = b0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b0), 0x142, 0xA, 0xF, false));
                }
            if (li == 16) {
                const float rn = n > 0.f ? 1.0f / n : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float ms = s[h][q] * rn;
                        const int lg = (wc * 64 + 32 * h + 16 * g + q * CPG) / CPG;
                        float* d = lds + ((wp * WX + wx) * gpb + lg) * 3;
                        d[0] = n; d[1] = n > 0.f ? piv[h][q] + ms : 0.f; d[2] = n > 0.f ? fmaxf(ss[h][q] - s[h][q] * ms, 0.f) : 0.f;
                    }
            }
        };
        auto stats_c = [&](auto cpg_tag) {
            if (full) stats(cpg_tag, std::true_type{}); else stats(cpg_tag, std::false_type{});
        };
        if (cpg == 4) stats_c(std::integral_constant<int, 4>{});
        else if (cpg == 8) stats_c(std::integral_constant<int, 8>{});
        else stats_c(std::integral_constant<int, 16>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if ((int)threadIdx.x < gpb) {
            float nn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int pw = 0; pw < WP * WX; ++pw) {               // the pixel-wave groups, in a fixed order
                const float* d = lds + (pw * gpb + threadIdx.x) * 3;
                vt_chan_merge(nn, mean, m2, d[0], d[1], d[2]);
            }
            const int G = a.Cout / cpg;
            float* o = a.gn_partial + (((long long)b * a.ptiles + tile) * G + c0 / cpg + threadIdx.x) * 3;
            o[0] = nn; o[1] = mean; o[2] = m2;
        }
    }
#ifdef HALO_STAMP
    STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
    if (STAMP_ON) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_halo8_stamps[(long long)blockIdx.x * 16 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

}  // namespace

bool vt_conv3x3_halo_fp8_supported(int Cin, int Cout) { return Cin >= 64 && (Cin % 64) == 0 && (Cout % 128) == 0; }
int vt_conv3x3_halo_fp8_tiles_shape(int H, int W, int shape) {
    const int rows = shape == 1 ? Shape<4, 1>::ROWS : Shape<2, 1>::ROWS, twx = shape == 2 ? Shape<2, 2>::TWX : Shape<2, 1>::TWX;
    return ((W + twx - 1) / twx) * ((H + rows - 1) / rows);
}
int vt_conv3x3_halo_fp8_tiles(int H, int W) { return vt_conv3x3_halo_fp8_tiles_shape(H, W, 0); }      // (the default shape has the most tiles)

// LDS row (inside a 32-cout MFMA block) that must hold cout_local, so that accumulator register r of lane group g is cout 16 g + r
int vt_halo_fp8_row_of_cout(int cout_local /*0..31*/) {
    const int gg = cout_local >> 4, q = (cout_local >> 2) & 3, t = cout_local & 3;
    return 8 * q + 4 * gg + t;
}

namespace {
template <int WP, int WX>
hipError_t launch_shape(const Conv3x3Fp8Args& a, hipStream_t s) {
    typedef Shape<WP, WX> SH;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] { return hipFuncSetAttribute((const void*)conv3x3_halo_fp8_kernel<WP, WX>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::SMEM); });
    if (ea != hipSuccess) return ea;
    const long long tiles = (long long)((a.W + SH::TWX - 1) / SH::TWX) * ((a.H + SH::ROWS - 1) / SH::ROWS);
    const long long nblk = tiles * (a.Cout / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    Conv3x3Fp8Args k = a;
    k.tiles_x = (a.W + SH::TWX - 1) / SH::TWX; k.ctiles = a.Cout / BC; k.per_img = (int)(tiles * k.ctiles); k.ptiles = (int)tiles;
    auto magic = [&](long long d) -> unsigned long long {
        return (nblk * d < (1LL << 40) && nblk < (1LL << 23)) ? ((1ULL << 40) / (unsigned long long)d + 1ULL) : 0ULL;
    };
    k.m_per_img = magic(k.per_img); k.m_ctiles = magic(k.ctiles); k.m_tiles_x = magic(k.tiles_x);
    hipLaunchKernelGGL((conv3x3_halo_fp8_kernel<WP, WX>), dim3((unsigned)nblk), dim3(SH::NT), SH::SMEM, s, k);
    return hipGetLastError();
}
}  // namespace

hipError_t vt_launch_conv3x3_halo_fp8(const Conv3x3Fp8Args& a, hipStream_t s) {
    if (!a.X || !a.Wp || !a.mult || !a.zeros || (!a.out_f32 && !a.out_bf16 && !a.out_f16 && !a.out_e4m3)) return hipErrorInvalidValue;
    if ((a.res && a.res_f16) || (a.res_f16 && a.out_f32)) return hipErrorInvalidValue;
    if (!vt_conv3x3_halo_fp8_supported(a.Cin, a.Cout) || a.batch <= 0 || a.H <= 0 || a.W <= 0) return hipErrorInvalidValue;
    if (a.gn_partial && a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;        // 32-bit per-image offsets
    if ((long long)(a.Cin / 64) * 9 * a.Cout * 64 >= (1LL << 31)) return hipErrorInvalidValue;
    if ((a.scX != nullptr) != (a.scW != nullptr)) return hipErrorInvalidValue;
    if (a.scX && (a.scCin <= 0 || (a.scCin % 32) || a.res || a.res_f16 || (long long)a.H * a.W * a.scCin >= (1LL << 31))) return hipErrorInvalidValue;
    if (a.shape == 1) return launch_shape<4, 1>(a, s);
    if (a.shape == 2) return launch_shape<2, 2>(a, s);
    if (a.shape != 0) return hipErrorInvalidValue;
    return launch_shape<2, 1>(a, s);
}
