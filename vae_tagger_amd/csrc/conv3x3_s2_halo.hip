// 3x3 / stride 2 convolution with the asymmetric (0,1,0,1) padding of diffusers' Downsample2D (restated at
// oracle/encoder_ref.py:104-106) as a halo-tile implicit GEMM -- the stride-1 kernel's design (conv3x3_halo.hip: X staged once per
// channel chunk and read at shifted rows by several taps, weights through an LDS ring filled by LDS-DMA under counted vmcnt waits,
// one raw s_barrier per K-step, the fragments of step t+1 read during step t, halo rows reused across ky from registers) applied
// to the four PHASE PLANES of the input:
//     out(y, x) = sum_{ky,kx} w[ky][kx] . in(2y + ky, 2x + kx),   plane (py, px)(i, j) = in(2i + py, 2j + px)
//     tap (ky, kx) reads plane (ky & 1, kx & 1) at (y + (ky >> 1), x + (kx >> 1)): a stride-1 access with shift 0 or 1.
// Plane (0,0) serves four taps from a 17 x 17 halo, (0,1) two from 17 x 16, (1,0) two from 16 x 17, (1,1) one from 16 x 16.
// Nothing is rearranged in memory: the LDS-DMA's per-lane source address gathers a plane's pixels straight from the NHWC
// tensor (64 contiguous bytes per pixel and 32-channel chunk), every input byte is fetched once per cout tile.
// The generic implicit GEMM ran these three launches at 0.63 / 0.80 / 0.97 PF (a GEMM tile streams BOTH operands through LDS
// on every K-step: fill-bound); here only the weights stream per K-step and the X bytes per K-step are 7.7 KB instead of 32 KB.
//
// Workgroup = 4 waves (2 output-row groups x 2 cout groups) x <= 256 VGPRs, 78.75 KB LDS: two workgroups per CU.
// Tile = 16 x 16 output pixels x 128 couts; wave tile = 8 rows x 16 px x 64 couts = 8 x 4 v_mfma_f32_16x16x32_bf16 per K-step.
// K-steps of one 32-channel chunk, in plane order (the host packs the weights in this order, vt_s2_tap_of_step):
//     step 0..3  plane (0,0): (dy,dx) = (0,0) (1,0) (0,1) (1,1)  = taps (0,0) (2,0) (0,2) (2,2)     [dx-major: rows reused over dy]
//     step 4..5  plane (0,1): (0,0) (1,0)                         = taps (0,1) (2,1)
//     step 6..7  plane (1,0): (0,0) (0,1)                         = taps (1,0) (1,2)
//     step 8     plane (1,1): (0,0)                               = tap  (1,1)
// X staging: a ring of three plane buffers; plane s + 3 of the sequence (chunk-major, plane-minor) is DMA'd into the buffer of
// plane s two barriers after that buffer's last fragment read (the WAR rule of conv3x3_halo.hip), three or four K-steps before
// its first use.  Weights: ring of three stages, W(t + 2) issued during step t.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int HB = 64;                       // bytes per LDS row (32 bf16)
constexpr int TW = 16, ROWS = 16;            // output tile
constexpr int WP = 2, WC = 2, TP = 8, TC = 4;
constexpr int NWV = WP * WC, NT = 64 * NWV, BC = WC * 64;
constexpr int NW = 3, LEAD = NW - 1;         // weight ring
constexpr int WBUF = BC * HB;                // 8 KB per stage
constexpr int WPW = BC / 16 / NWV;           // W pieces per wave and K-step: 2
constexpr int NXB = 3;                       // plane-buffer ring
#ifndef S2_W_AT
#define S2_W_AT (TP / 2 - 1)                 // MFMA row group after which a K-step issues its weight DMA
#endif
constexpr int XSTRIDE = 289 * HB;            // plane (0,0): 17 x 17 halo rows
constexpr int SMEM = NXB * XSTRIDE + NW * WBUF;   // 80 064 B
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");

// plane pl = 2 py + px
__host__ __device__ constexpr int pl_py(int pl) { return pl >> 1; }
__host__ __device__ constexpr int pl_px(int pl) { return pl & 1; }
__host__ __device__ constexpr int pl_pitch(int pl) { return pl_px(pl) ? 16 : 17; }      // halo pixels per halo row
__host__ __device__ constexpr int pl_hrows(int pl) { return pl_py(pl) ? 16 : 17; }
__host__ __device__ constexpr int pl_rows(int pl) { return pl_pitch(pl) * pl_hrows(pl); }  // 289, 272, 272, 256
__host__ __device__ constexpr int pl_pieces(int pl) { return pl_rows(pl) / 16; }        // full 16-row DMA pieces: 18, 17, 17, 16
__host__ __device__ constexpr int nx_pl(int pl) { return pl == 3 ? 4 : 5; }              // X DMA instructions per wave and plane
// K-step p of a chunk -> plane / shifts
__host__ __device__ constexpr int st_plane(int p) { return p < 4 ? 0 : p < 6 ? 1 : p < 8 ? 2 : 3; }
__host__ __device__ constexpr int st_dx(int p) { return (p == 2 || p == 3 || p == 7) ? 1 : 0; }
__host__ __device__ constexpr int st_dy(int p) { return (p == 1 || p == 3 || p == 5) ? 1 : 0; }
__host__ __device__ constexpr bool st_two(int p) { return p < 6; }                       // the step's group walks two dy
// group that follows step p's group: (plane, dx, rows); 9 = the next chunk's first group
__host__ __device__ constexpr int ng_plane(int p) { return p < 2 ? 0 : p < 4 ? 1 : p < 7 ? 2 : p < 8 ? 3 : 0; }
__host__ __device__ constexpr int ng_dx(int p) { return (p < 2 || p == 6) ? 1 : 0; }
__host__ __device__ constexpr int ng_rows(int p) { return (p < 4 || p == 8) ? 9 : 8; }

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// the same for a wave-uniform value: the plane-buffer bases must not be combined with the per-lane fragment offsets into
// loop-invariant address tables (3 slots x 8 bases x 2 pitches live across the whole K loop spill)
__device__ __forceinline__ int opaque_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// F16: X and Wp hold fp16 bits (v_mfma_f32_16x16x32_f16) -- the fp16-operand mode of the convs (vt_set_flag 18)
template <bool F16>
__global__ __launch_bounds__(NT, 2) void conv3x3_s2_halo_kernel(const Conv3x3S2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbase = smem;                    // NXB plane buffers
    char* const wbase = smem + NXB * XSTRIDE;    // NW weight stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WC, wc = wave % WC;
    const int fr = lane & 15, fq = lane >> 4;

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
    const int ty0 = tyi * ROWS, tx0 = (tile - tyi * a.tiles_x) * TW;       // output coordinates
    const int c0 = ct * BC;
    const bf16_t* Xb = a.X + (long long)b * a.H * a.W * a.Cin;
    const int xpix = a.x_planar ? 32 : a.Cin, xchunk = a.x_planar ? a.H * a.W * 32 : 32;      // element strides of a pixel / a 32-channel chunk (wave-uniform)
    const int nchunk = a.Cin >> 5;
    const int nk = nchunk * 9;

    // ---- DMA bookkeeping: one wave-instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical 16-B chunk (l & 3),
    // logical chunk = physical ^ swz(row), swz(row) = ((row >> 2) & 1) << 1 (pieces start at multiples of 16 rows)
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const int wsrc0 = (c0 + wave * 16 + drow) * 32 + dchunk * 8;
    const int wstep = a.Cout * 32;

    // Every wave issues the SAME number of X pieces per plane (5, 5, 5, 4), so that the counted waits below are immediates (a run-time
    // count costs a 15-way branch in front of every barrier and splits each K-step into basic blocks the register allocator cannot
    // balance).  Plane (0,0) has 289 halo rows = 18 sixteen-row pieces + its corner pixel: a 19th piece covers rows 273..288 (15 of
    // them a second time -- same bytes to the same place); planes of 19 / 17 / 17 pieces are topped up to 20 by re-loading their last
    // piece (an L2 hit).  The chunk swizzle is taken from the piece's actual first row, so the shifted piece needs no special case.
    // plane `pl` (compile-time) of channel chunk `chunk` -> ring buffer `slot`
    auto issue_x = [&](auto pl_tag, int chunk, int slot) {
        constexpr int PL = decltype(pl_tag)::value;
        constexpr int PITCH = pl_pitch(PL), PY = pl_py(PL), PX = pl_px(PL), NP = pl_pieces(PL), NXP = nx_pl(PL);
        constexpr int NPT = PL == 0 ? NP + 1 : NP;                       // with the shifted corner piece
        char* dst = xbase + slot * XSTRIDE;
#pragma nounroll
        for (int j = 0; j < NXP; ++j) {
            int piece = j * NWV + wave;
            if (piece >= NPT) piece = NPT - 1;
            const int row0 = (PL == 0 && piece == NP) ? pl_rows(0) - 16 : piece * 16;
            const int hr = row0 + opaque(drow);
            const int hy = hr / PITCH, hx = hr - hy * PITCH;
            const int iy = 2 * (ty0 + hy) + PY, ix = 2 * (tx0 + hx) + PX;
            const int lch = (lane & 3) ^ (((hr >> 2) & 1) << 1);         // logical chunk this lane's physical slot holds
            // NHWC: pixel stride Cin; chunk-planar ([Cin/32][H][W][32], a.x_planar): a 128-B line holds the SAME chunk of two neighbouring pixels,
            // i.e. the two halves a row's px = 0 / px = 1 planes take three K-steps apart -- instead of two chunks of one pixel taken nine steps
            // apart, by when the line has left the L2 (rocprofv3: every line of X was fetched twice)
            const void* src = (iy < a.H && ix < a.W) ? (const void*)(Xb + ((iy * a.W + ix) * xpix + chunk * xchunk + lch * 8)) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(dst + row0 * HB), 16, 0, 0);
        }
    };
    auto issue_w = [&](int t) {
        char* dst = wbase + (t % NW) * WBUF;
        const bf16_t* wt = a.Wp + (long long)t * wstep + opaque(wsrc0);
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NWV * 16 * 32), VT_LDS_PTR(dst + (j * NWV + wave) * 1024), 16, 0, 0);
    };

    // ---- prologue.  VM issue order: X(plane 0), W(0), W(1), X(plane 1) -- the same "W before X" order every K-step keeps, so that
    // the operations younger than the weight tile a barrier needs are exactly the X pieces issued in the step before it.
    issue_x(std::integral_constant<int, 0>{}, 0, 0);
    issue_w(0);
    issue_w(1);
    issue_x(std::integral_constant<int, 1>{}, 0, 1);
    asm volatile("" ::: "memory");

    // ---- fragment addressing.  W fragment i: stage row wc*64 + i*16 + fr.  X fragment: halo row R + fr of its plane buffer with
    // R = (wp*8 + r) * pitch + dx; the swizzle bit is bit 2 of (R + fr) = bit 2 of ((R & 7) + fr), and R & 7 is a compile-time
    // constant (wp * 8 * pitch is a multiple of 8): eight per-lane bases + immediates cover every read.
    const int wfoff = (wc * 64 + fr) * HB + ((fq ^ (((fr >> 2) & 1) << 1)) << 4);
    int xsel[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xsel[k] = fr * HB + ((fq ^ ((((k + fr) >> 2) & 1) << 1)) << 4);
    const int wpoff17 = wp * TP * 17 * HB, wpoff16 = wp * TP * 16 * HB;
    auto xfrag = [&](const char* buf /* plane buffer + this wave's row-group offset */, int pitch /* compile-time */, int r, int dx) -> bf16x8 {
        const int rel = r * pitch + dx;
        return *(const bf16x8*)(buf + xsel[rel & 7] + rel * HB);
    };
    // the same read, ordered behind the MFMA that wrote `after`: left alone the scheduler lifts a step's refills above the MFMAs that
    // still read the old rows, into fresh registers (the kernel then wants 360 VGPRs and spills); the empty asm emits nothing
    auto xfrag_after = [&](const char* buf, int pitch, int r, int dx, const f32x4& after) -> bf16x8 {
        const int rel = r * pitch + dx;
        int off = xsel[rel & 7];
        asm volatile("" : "+v"(off) : "v"(after));
        return *(const bf16x8*)(buf + off + rel * HB);
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 wfc[TC], xr[TP + 1];
    {
        wait_vmcnt(WPW + nx_pl(1));              // X(plane 0) and W(0) landed (this wave's pieces) ...
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();            // ... and everybody else's
        asm volatile("" ::: "memory");
        const char* ws0 = wbase + opaque(wfoff);
#pragma unroll
        for (int i = 0; i < TC; ++i) wfc[i] = *(const bf16x8*)(ws0 + i * 16 * HB);
#pragma unroll
        for (int r = 0; r < TP + 1; ++r) xr[r] = xfrag(xbase + opaque_s(wpoff17), 17, r, 0);
    }

    // ring slot of plane pl of chunk c: (4 c + pl) % 3 = (c + pl) % 3
    int cm = 0;                                  // chunk % 3
    auto slot_of = [&](int cmod, int pl) -> int { const int s = cmod + pl; return s >= 3 ? (s >= 6 ? s - 6 : s - 3) : s; };

    auto do_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int tbase = chunk * 9;
        const int cmn = cm == 2 ? 0 : cm + 1;    // (chunk + 1) % 3
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int t = tbase + p;
            // barrier of step t: W(t + 1) (issued in step t - 1, first thing) must have landed; younger operations = the X pieces
            // step t - 1 issued after it.  Steps that issue X: p = 0 (plane 2), 3 (plane 3), 5 (next chunk's plane 0), 8 (its plane 1).
            {
                int n = 0;
                if (p == 1) n = nx_pl(2);
                if (p == 4) n = nx_pl(3);
                if (p == 6 && !LAST) n = nx_pl(0);
                if (p == 0) n = nx_pl(1);        // previous chunk's step 8 (or the prologue)
                wait_vmcnt(n);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            constexpr int dummy = 0; (void)dummy;
            const int pl = st_plane(p), dy = st_dy(p);
            const bool two = st_two(p);
            const bool has_next = !LAST || p < 8;                    // a next K-step exists
            const int npl = ng_plane(p), ndx = ng_dx(p), nrows = ng_rows(p);
            const int npitch = pl_pitch(npl);
            const bool next_chunk = p == 8;
            const char* xs_n = xbase + opaque_s(slot_of(next_chunk ? cmn : cm, npl) * XSTRIDE + (npitch == 17 ? wpoff17 : wpoff16));
            const char* ws_n = wbase + ((t + 1) % NW) * WBUF + opaque(wfoff);
            (void)pl;
            auto refill = [&](int r, const f32x4& after) {
                if (has_next && r < nrows) xr[r] = xfrag_after(xs_n, npitch, r, ndx, after);
            };
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
                    if constexpr (F16) {
                        typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8m, wfc[i]), __builtin_bit_cast(f16x8m, xr[two ? j + dy : j]), acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfc[i], xr[two ? j + dy : j], acc[i][j], 0, 0, 0);
                    }
                // rows whose last reader has just issued are refilled with the next group's rows
                if (two) {
                    if (dy == 0) { if (j == 0) refill(0, acc[TC - 1][j]); }
                    else refill(j + 1, acc[TC - 1][j]);
                } else {
                    refill(j, acc[TC - 1][j]);
                    if (j == TP - 1) refill(TP, acc[TC - 1][j]);
                }
                if (j == S2_W_AT) {
                    if (!LAST || p + LEAD < 9) issue_w(t + LEAD);
                }
                if (j == TP / 2 - 1) {
                    if (p == 0) issue_x(std::integral_constant<int, 2>{}, chunk, slot_of(cm, 2));
                    if (p == 3) issue_x(std::integral_constant<int, 3>{}, chunk, slot_of(cm, 3));
                    if constexpr (!LAST) {
                        if (p == 5) issue_x(std::integral_constant<int, 0>{}, chunk + 1, slot_of(cmn, 0));
                        if (p == 8) issue_x(std::integral_constant<int, 1>{}, chunk + 1, slot_of(cmn, 1));
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (has_next) {
#pragma unroll
                for (int i = 0; i < TC; ++i) wfc[i] = *(const bf16x8*)(ws_n + i * 16 * HB);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cm = cmn;
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) do_chunk(chunk, std::false_type{});
    do_chunk(nchunk - 1, std::true_type{});
    (void)nk;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) asm volatile("" : "+v"(acc[i][j]));
    asm volatile("" ::: "memory");

    // ---- epilogue: accumulator register r of tile i in lane (fq, fr) is cout c0 + wc*64 + 16 fq + 4 i + r of output pixel
    // (ty0 + wp*8 + j, tx0 + fr) (interleaved cout rows, vt_halo_row_of_cout): 32-byte runs per lane
    const long long ob = (long long)b * a.Ho * a.Wo * a.Cout;
    const int x = tx0 + fr;
    const int cw = c0 + wc * 64 + 16 * fq;
    f32x4 bv[4];                                 // (added here: holding the bias through the K loop costs the loop four registers it does not have)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = a.bias ? *(const f32x4*)(a.bias + cw + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int y = ty0 + wp * TP + j;
        if (y >= a.Ho || x >= a.Wo) continue;
        valid |= 1u << j;
        const long long o = ob + ((long long)y * a.Wo + x) * a.Cout + cw;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            f32x4 v = acc[i][j] + bv[i];
            if (a.res) v += *(const f32x4*)(a.res + o + 4 * i);
            if (a.out_f32) *(f32x4*)(a.out_f32 + o + 4 * i) = v;
            acc[i][j] = v;
        }
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        if (a.out_f16) {
#pragma unroll
            for (int i = 0; i < TC; i += 2) {
                f16x8 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j][r]; h[4 + r] = (f16_t)acc[i + 1][j][r]; }
                *(f16x8*)(a.out_f16 + o + 4 * i) = h;
            }
        }
        if (a.out_bf16) {
#pragma unroll
            for (int i = 0; i < TC; i += 2) {
                if (a.out16_f16) {                             // the copy carries fp16 bits when its consumer (a fused shortcut) runs on fp16 operands
                    f16x8 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[r] = (f16_t)acc[i][j][r]; h[4 + r] = (f16_t)acc[i + 1][j][r]; }
                    *(f16x8*)((f16_t*)a.out_bf16 + o + 4 * i) = h;
                } else {
                    bf16x8 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[r] = (bf16_t)acc[i][j][r]; h[4 + r] = (bf16_t)acc[i + 1][j][r]; }
                    *(bf16x8*)(a.out_bf16 + o + 4 * i) = h;
                }
            }
        }
    }
    if (a.gn_partial) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave is done with the staging LDS
        const int G = a.Cout / a.gn_cpg;
        float* out = a.gn_partial + (((long long)b * a.ptiles + tile) * G + c0 / a.gn_cpg) * 3;
        vt_gn_epilogue_partials_il<TC, TP>(acc, valid, a.gn_cpg, wp, WP, wc * 64, BC, (float*)smem, out);
    }
}

// [Cout][9][Cin] (the generic kernel's layout) -> Wp[Cin/32][9 steps][Cout rows][32]; device-side, for the op-level entry
__global__ void repack_ohwi_s2_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int Cin, int Cout) {
    const long long n = (long long)Cout * 9 * Cin;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 9);
    const int co = (int)(i / ((long long)Cin * 9));
    const int row = (co & ~63) + vt_halo_row_of_cout(co & 63);
    wp[(((long long)(ci >> 5) * 9 + vt_s2_step_of_tap(tap)) * Cout + row) * 32 + (ci & 31)] = w[i];
}

}  // namespace

bool vt_conv3x3_s2_supported(int Cin, int Cout) { return Cin >= 32 && (Cin % 32) == 0 && (Cout % 128) == 0; }
int vt_conv3x3_s2_tiles(int Ho, int Wo) { return ((Wo + TW - 1) / TW) * ((Ho + ROWS - 1) / ROWS); }

hipError_t vt_launch_repack_ohwi_to_s2(const bf16_t* w, bf16_t* wp, int Cin, int Cout, hipStream_t s) {
    const long long n = (long long)Cout * 9 * Cin;
    hipLaunchKernelGGL(repack_ohwi_s2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cin, Cout);
    return hipGetLastError();
}

hipError_t vt_launch_conv3x3_s2(const Conv3x3S2Args& a, hipStream_t s) {
    if (!a.X || !a.Wp || !a.zeros || (!a.out_f32 && !a.out_bf16 && !a.out_f16)) return hipErrorInvalidValue;
    if (!vt_conv3x3_s2_supported(a.Cin, a.Cout) || a.batch <= 0 || a.H < 2 || a.W < 2) return hipErrorInvalidValue;
    if (a.gn_partial && a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;        // 32-bit per-image offsets
    if ((long long)(a.Cin / 32) * 9 * a.Cout * 32 >= (1LL << 31)) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_s2_halo_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_s2_halo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        return e;
    });
    if (ea != hipSuccess) return ea;
    Conv3x3S2Args k = a;
    k.Ho = a.H / 2; k.Wo = a.W / 2;                       // pad (0,1,0,1), 3x3, stride 2: floor((H + 1 - 3) / 2) + 1
    const long long tiles = vt_conv3x3_s2_tiles(k.Ho, k.Wo);
    const long long nblk = tiles * (a.Cout / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    k.tiles_x = (k.Wo + TW - 1) / TW; k.ctiles = a.Cout / BC; k.per_img = (int)(tiles * k.ctiles); k.ptiles = (int)tiles;
    auto magic = [&](long long d) -> unsigned long long {
        return (nblk * d < (1LL << 40) && nblk < (1LL << 23)) ? ((1ULL << 40) / (unsigned long long)d + 1ULL) : 0ULL;
    };
    k.m_per_img = magic(k.per_img); k.m_ctiles = magic(k.ctiles); k.m_tiles_x = magic(k.tiles_x);
    if (a.f16) hipLaunchKernelGGL(conv3x3_s2_halo_kernel<true>, dim3((unsigned)nblk), dim3(NT), SMEM, s, k);
    else hipLaunchKernelGGL(conv3x3_s2_halo_kernel<false>, dim3((unsigned)nblk), dim3(NT), SMEM, s, k);
    return hipGetLastError();
}
