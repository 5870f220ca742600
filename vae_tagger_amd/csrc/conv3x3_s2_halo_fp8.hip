// Downsample2D's conv (3x3, stride 2, pad (0,1,0,1)) on fp8 (OCP e4m3) operands -- the fp8 mode's (vt_set_flag 11) stride-2 layers.
// conv3x3_s2_halo.hip's phase-plane scheme (out(y,x) = sum w[ky][kx] . in(2y+ky, 2x+kx); tap (ky,kx) reads plane (ky&1, kx&1) at
// shifts 0 / 1; ring of three plane buffers filled by gathering LDS-DMA, K-steps in plane order, every wave issuing 5 / 5 / 5 / 4 X
// pieces per plane so that all wait counts are immediates) with conv3x3_halo_fp8.hip's arithmetic: v_mfma_scale_f32_32x32x64_f8f6f4,
// a K-step = (64-channel chunk, tap), LDS rows of 64 B = 64 channels, 32-byte fragments (lane (g, i): channels 32 g .. of row i) under
// the chunk swizzle (row ^ row >> 2) & 3, cout rows permuted so that a lane's 16 accumulator registers are 16 consecutive couts.
// Tile = 8 x 32 output pixels x 128 couts, 4 waves (2 row groups x 2 cout groups), wave tile 4 rows x 32 px x 64 couts = 8 MFMAs per
// K-step; plane halos 9 x 33, 9 x 32, 8 x 33, 8 x 32 pixels; LDS 3 x 18.6 KB + 3 x 8 KB = 79.7 KB: two workgroups per CU.
// Input: the block output as e4m3(x) (scale 1, written by the producing conv2's epilogue); out = acc * mult[cout] + bias.
// The generic fp8 GEMM ran these three launches in 3.4 ms per step (1.08 PF).
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int HB = 64;                       // bytes per LDS row
constexpr int TWX = 32, ROWS = 8;            // output tile
constexpr int WP = 2, WC = 2, TP = 4;
constexpr int NWV = WP * WC, NT = 64 * NWV, BC = WC * 64;
constexpr int NW = 3, LEAD = NW - 1;
constexpr int WBUF = BC * HB;                // 8 KB per stage
constexpr int WPW = BC / 16 / NWV;           // 2 W pieces per wave and K-step
constexpr int NXB = 3;
constexpr int XSTRIDE = 297 * HB;            // plane (0,0): 9 x 33 halo rows
constexpr int SMEM = NXB * XSTRIDE + NW * WBUF;   // 81 600 B
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int pl_py(int pl) { return pl >> 1; }
__host__ __device__ constexpr int pl_px(int pl) { return pl & 1; }
__host__ __device__ constexpr int pl_pitch(int pl) { return pl_px(pl) ? 32 : 33; }
__host__ __device__ constexpr int pl_hrows(int pl) { return pl_py(pl) ? 8 : 9; }
__host__ __device__ constexpr int pl_rows(int pl) { return pl_pitch(pl) * pl_hrows(pl); }      // 297, 288, 264, 256
__host__ __device__ constexpr int pl_pieces(int pl) { return (pl_rows(pl) + 15) / 16; }        // 19, 18, 17, 16 (the last one shifted back when ragged)
__host__ __device__ constexpr int nx_pl(int pl) { return pl == 3 ? 4 : 5; }                    // X DMA instructions per wave and plane
__host__ __device__ constexpr int st_dy(int p) { return (p == 1 || p == 3 || p == 5) ? 1 : 0; }
__host__ __device__ constexpr bool st_two(int p) { return p < 6; }
__host__ __device__ constexpr int ng_plane(int p) { return p < 2 ? 0 : p < 4 ? 1 : p < 7 ? 2 : p < 8 ? 3 : 0; }
__host__ __device__ constexpr int ng_dx(int p) { return (p < 2 || p == 6) ? 1 : 0; }
__host__ __device__ constexpr int ng_rows(int p) { return (p < 4 || p == 8) ? TP + 1 : TP; }

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
__device__ __forceinline__ int opaque_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 2)) & 3; }
// 32-byte fragment of LDS row `row`: logical chunks 2 g, 2 g + 1
__device__ __forceinline__ i32x8 read_frag(const char* base, int row, int g) {
    const int a = row * HB + (((2 * g) ^ swz(row)) << 4);
    const i32x4 lo = *(const i32x4*)(base + a), hi = *(const i32x4*)(base + (a ^ 16));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(NT, 2) void conv3x3_s2_halo_fp8_kernel(const Conv3x3S2Fp8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbase = smem;
    char* const wbase = smem + NXB * XSTRIDE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WC, wc = wave % WC;
    const int g = lane >> 5, li = lane & 31;

    auto fdiv = [](int n, unsigned long long m, int d) -> int {
        return m ? (int)(((unsigned long long)(unsigned)n * m) >> 40) : n / d;
    };
    int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int b = fdiv(logical, a.m_per_img, a.per_img);
    logical -= b * a.per_img;
    const int tile = fdiv(logical, a.m_ctiles, a.ctiles);
    const int ct = logical - tile * a.ctiles;
    const int tyi = fdiv(tile, a.m_tiles_x, a.tiles_x);
    const int ty0 = tyi * ROWS, tx0 = (tile - tyi * a.tiles_x) * TWX;       // output coordinates
    const int c0 = ct * BC;
    const unsigned char* Xb = a.X + (long long)b * a.H * a.W * a.Cin;
    const int xpix = a.x_planar ? 64 : a.Cin, xchunk = a.x_planar ? a.H * a.W * 64 : 64;      // byte strides of a pixel / a 64-channel chunk (wave-uniform)
    const int nchunk = a.Cin >> 6;

    // ---- DMA bookkeeping: one wave-instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical chunk (l & 3),
    // logical chunk = physical ^ swz(row) with the row's ACTUAL index (the ragged planes' last piece is shifted back to end at the plane's end)
    const int drow = lane >> 2;
    const int wdchunk = (lane & 3) ^ swz(drow);                            // weight stages: pieces start at multiples of 16 rows
    const int wsrc0 = (c0 + wave * 16 + drow) * HB + wdchunk * 16;
    const int wstep = a.Cout * HB;

    auto issue_x = [&](auto pl_tag, int chunk, int slot) {
        constexpr int PL = decltype(pl_tag)::value;
        constexpr int PITCH = pl_pitch(PL), PY = pl_py(PL), PX = pl_px(PL), NP = pl_pieces(PL), NXP = nx_pl(PL), NR = pl_rows(PL);
        char* dst = xbase + slot * XSTRIDE;
#pragma nounroll
        for (int j = 0; j < NXP; ++j) {
            int piece = j * NWV + wave;
            if (piece >= NP) piece = NP - 1;
            const int row0 = (piece == NP - 1) ? NR - 16 : piece * 16;        // (NR - 16 = piece * 16 when the plane is a whole number of pieces)
            const int hr = row0 + opaque(drow);
            const int hy = hr / PITCH, hx = hr - hy * PITCH;
            const int iy = 2 * (ty0 + hy) + PY, ix = 2 * (tx0 + hx) + PX;
            const int lch = (lane & 3) ^ swz(hr);
            // (a.x_planar: chunk-planar [Cin/64][H][W][64] instead of NHWC -- see conv3x3_s2_halo.hip)
            const void* src = (iy < a.H && ix < a.W) ? (const void*)(Xb + ((iy * a.W + ix) * xpix + chunk * xchunk + lch * 16)) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(dst + row0 * HB), 16, 0, 0);
        }
    };
    auto issue_w = [&](int t) {
        char* dst = wbase + (t % NW) * WBUF;
        const unsigned char* wt = a.Wp + (long long)t * wstep + opaque(wsrc0);
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + j * NWV * 16 * HB), VT_LDS_PTR(dst + (j * NWV + wave) * 1024), 16, 0, 0);
    };

    // ---- prologue.  VM issue order: X(plane 0), W(0), W(1), X(plane 1) (the "W before X" order of every K-step)
    issue_x(std::integral_constant<int, 0>{}, 0, 0);
    issue_w(0);
    issue_w(1);
    issue_x(std::integral_constant<int, 1>{}, 0, 1);
    asm volatile("" ::: "memory");

    f32x16 acc[2][TP];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][j][r] = 0.f;

    // fragment rows: W stage row wc*64 + 32 h + li; X plane row (wp*TP + r) * pitch + dx + li
    const int wrow0 = wc * 64 + li;
    const int xrow33 = wp * TP * 33 + li, xrow32 = wp * TP * 32 + li;
    i32x8 wfc[2], xr[TP + 1];
    {
        wait_vmcnt(WPW + nx_pl(1));              // X(plane 0) and W(0) landed (this wave's pieces) ...
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();            // ... and everybody else's
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) wfc[h] = read_frag(wbase, opaque(wrow0) + 32 * h, g);
#pragma unroll
        for (int r = 0; r < TP + 1; ++r) xr[r] = read_frag(xbase, opaque(xrow33) + r * 33, g);
    }

    int cm = 0;                                  // chunk % 3: ring slot of plane pl of chunk c = (c + pl) % 3
    auto slot_of = [&](int cmod, int pl) -> int { const int s = cmod + pl; return s >= 3 ? s - 3 : s; };

    auto do_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int tbase = chunk * 9;
        const int cmn = cm == 2 ? 0 : cm + 1;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int t = tbase + p;
            {
                // barrier of step t: W(t + 1) landed; younger operations = the X pieces step t - 1 issued after it
                int n = 0;
                if (p == 1) n = nx_pl(2);
                if (p == 4) n = nx_pl(3);
                if (p == 6 && !LAST) n = nx_pl(0);
                if (p == 0) n = nx_pl(1);
                wait_vmcnt(n);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int dy = st_dy(p);
            const bool two = st_two(p);
            const bool has_next = !LAST || p < 8;
            const int npl = ng_plane(p), ndx = ng_dx(p), nrows = ng_rows(p);
            const int npitch = pl_pitch(npl);
            const char* xs_n = xbase + opaque_s(slot_of(p == 8 ? cmn : cm, npl) * XSTRIDE);
            const char* ws_n = wbase + ((t + 1) % NW) * WBUF;
            auto refill = [&](int r) {
                if (has_next && r < nrows) xr[r] = read_frag(xs_n, opaque(npitch == 33 ? xrow33 : xrow32) + r * npitch + ndx, g);
            };
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wfc[h], xr[two ? j + dy : j], acc[h][j], 0, 0, 0, 127, 0, 127);
                if (two) {
                    if (dy == 0) { if (j == 0) refill(0); }
                    else refill(j + 1);
                } else {
                    refill(j);
                    if (j == TP - 1) refill(TP);
                }
                if (j == TP / 2 - 1) {
                    if (!LAST || p + LEAD < 9) issue_w(t + LEAD);
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
                for (int h = 0; h < 2; ++h) wfc[h] = read_frag(ws_n, opaque(wrow0) + 32 * h, g);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cm = cmn;
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) do_chunk(chunk, std::false_type{});
    do_chunk(nchunk - 1, std::true_type{});
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < TP; ++j) asm volatile("" : "+v"(acc[h][j]));
    asm volatile("" ::: "memory");

    // ---- epilogue: register r of lane (g, x = li) in acc[h][j] is cout cw(h) + r of output pixel (ty0 + wp*TP + j, tx0 + li)
    const long long ob = (long long)b * a.Ho * a.Wo * a.Cout;
    const int x = tx0 + li;
    unsigned valid = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cw = c0 + wc * 64 + 32 * h + 16 * g;
        f32x4 mul[4], bia[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mul[i] = *(const f32x4*)(a.mult + cw + 4 * i);
            bia[i] = a.bias ? *(const f32x4*)(a.bias + cw + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int y = ty0 + wp * TP + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][j][r] = fmaf(acc[h][j][r], mul[r >> 2][r & 3], bia[r >> 2][r & 3]);
            if (y >= a.Ho || x >= a.Wo) continue;
            valid |= 1u << j;
            const long long o = ob + ((long long)y * a.Wo + x) * a.Cout + cw;
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            if (a.res) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 rv = *(const f32x4*)(a.res + o + 4 * i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[h][j][4 * i + q] += rv[q];
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
    if (a.gn_partial) {
        // GroupNorm (n, mean, M2) of this tile's outputs (conv3x3_halo_fp8.hip's scheme: a group lives in one lane, pivot-shifted sums,
        // DPP row sums, the two pixel-row waves merged by Chan's formula in a fixed order)
        float* lds = (float*)(smem + SMEM - WBUF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int cpg = a.gn_cpg;
        const int gpb = BC / cpg;
        float npix = 0.f;
#pragma unroll
        for (int j = 0; j < TP; ++j) npix += (float)__popcll(__ballot((valid >> j) & 1u) & 0xffffffffull);
        const float n = npix * (float)cpg;
        const bool full = __ballot(valid != (1u << TP) - 1u) == 0ull;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto stats = [&](auto cpg_tag, auto full_tag) {
            constexpr int CPG = decltype(cpg_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int NQ = 16 / CPG;
            float piv[2][NQ], s[2][NQ], ss[2][NQ];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
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
                        float* d = lds + (wp * gpb + lg) * 3;
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
            static_assert(WP == 2, "two pixel-row waves are merged");
            const float* d0 = lds + threadIdx.x * 3;
            const float* d1 = lds + (gpb + threadIdx.x) * 3;
            float nn = 0.f, mean = 0.f, m2 = 0.f;
            vt_chan_merge(nn, mean, m2, d0[0], d0[1], d0[2]);
            vt_chan_merge(nn, mean, m2, d1[0], d1[1], d1[2]);
            const int G = a.Cout / cpg;
            float* o = a.gn_partial + (((long long)b * a.ptiles + tile) * G + c0 / cpg + threadIdx.x) * 3;
            o[0] = nn; o[1] = mean; o[2] = m2;
        }
    }
}

}  // namespace

bool vt_conv3x3_s2_fp8_supported(int Cin, int Cout) { return Cin >= 64 && (Cin % 64) == 0 && (Cout % 128) == 0; }
int vt_conv3x3_s2_fp8_tiles(int Ho, int Wo) { return ((Wo + TWX - 1) / TWX) * ((Ho + ROWS - 1) / ROWS); }

hipError_t vt_launch_conv3x3_s2_fp8(const Conv3x3S2Fp8Args& a, hipStream_t s) {
    if (!a.X || !a.Wp || !a.mult || !a.zeros || (!a.out_f32 && !a.out_bf16 && !a.out_f16)) return hipErrorInvalidValue;
    if (!vt_conv3x3_s2_fp8_supported(a.Cin, a.Cout) || a.batch <= 0 || a.H < 2 || a.W < 2) return hipErrorInvalidValue;
    if (a.gn_partial && a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;
    if ((long long)(a.Cin / 64) * 9 * a.Cout * 64 >= (1LL << 31)) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] { return hipFuncSetAttribute((const void*)conv3x3_s2_halo_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
    if (ea != hipSuccess) return ea;
    Conv3x3S2Fp8Args k = a;
    k.Ho = a.H / 2; k.Wo = a.W / 2;
    const long long tiles = vt_conv3x3_s2_fp8_tiles(k.Ho, k.Wo);
    const long long nblk = tiles * (a.Cout / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    k.tiles_x = (k.Wo + TWX - 1) / TWX; k.ctiles = a.Cout / BC; k.per_img = (int)(tiles * k.ctiles); k.ptiles = (int)tiles;
    auto magic = [&](long long d) -> unsigned long long {
        return (nblk * d < (1LL << 40) && nblk < (1LL << 23)) ? ((1ULL << 40) / (unsigned long long)d + 1ULL) : 0ULL;
    };
    k.m_per_img = magic(k.per_img); k.m_ctiles = magic(k.ctiles); k.m_tiles_x = magic(k.tiles_x);
    hipLaunchKernelGGL(conv3x3_s2_halo_fp8_kernel, dim3((unsigned)nblk), dim3(NT), SMEM, s, k);
    return hipGetLastError();
}
