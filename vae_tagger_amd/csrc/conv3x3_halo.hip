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
// Weights are pre-packed on the host as Wp[chunk32][tap][Cout][32] bf16 so that every K-step's tile is
// one contiguous BC*64-byte block.  MFMA orientation as in conv_gemm.hip: weights = A operand (rows =
// cout), pixels = B operand, so a lane holds 4 consecutive couts of one pixel.
#include <type_traits>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int HB = 64;        // bytes per LDS row (32 bf16)
constexpr int TW = 16;        // tile width in pixels (one MFMA column block)
constexpr int HWID = TW + 2;  // halo width
constexpr int NW = 6;         // weight ring depth

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define C(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
        C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int WP, int WC>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const Conv3x3Args a) {
    static_assert(WP * WC == 8, "8 waves");
    constexpr int ROWS = WP * 8;                 // tile rows (each wave: 8 rows x 16 px)
    constexpr int BC = WC * 64;                  // couts per workgroup (each wave: 64)
    constexpr int TP = 8, TC = 4;
    constexpr int HROWS = (ROWS + 2) * HWID;     // halo pixels
    constexpr int NXW = (HROWS + 127) / 128;     // X DMA wave-instructions per wave (16 rows each, 8 waves)
    constexpr int XBUF = NXW * 128 * HB;         // bytes per X halo buffer
    constexpr int WPW = BC / 128;                // W DMA wave-instructions per wave per K-step
    constexpr int WBUF = BC * HB;                // bytes per W stage
    static_assert(WPW >= 1, "BC >= 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbase = smem;                    // 2 halo buffers
    char* const wbase = smem + 2 * XBUF;         // NW weight stages

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WC, wc = wave % WC;

    // ---- tile coordinates
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + ROWS - 1) / ROWS;
    const int ctiles = a.Cout / BC;
    const int per_img = tiles_x * tiles_y * ctiles;
    int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / per_img;
    logical -= b * per_img;
    const int ct = logical % ctiles;
    const int tile = logical / ctiles;
    const int ty0 = (tile / tiles_x) * ROWS, tx0 = (tile % tiles_x) * TW;
    const int c0 = ct * BC;

    const bf16_t* Xb = a.X + (long long)b * a.H * a.W * a.Cin;
    const int nchunk = a.Cin >> 5;
    const int nk = nchunk * 9;

    // ---- DMA bookkeeping.  One wave-instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical
    // chunk (l & 3); logical chunk = physical ^ swz(row), swz(row) = ((row >> 2) & 1) << 1 = ((l >> 4) & 1) << 1.
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    int xsrc[NXW];          // element offset of (halo pixel, logical chunk) in the image, or -1 if outside
#pragma unroll
    for (int j = 0; j < NXW; ++j) {
        const int hr = (j * 8 + wave) * 16 + drow;
        const int hy = hr / HWID, hx = hr - hy * HWID;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool v = hr < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        xsrc[j] = v ? (iy * a.W + ix) * a.Cin + dchunk * 8 : -1;
    }
    // W: Wp[chunk][tap][Cout][32]; this lane's row inside the K-step tile
    int wsrc[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) wsrc[j] = (c0 + (j * 8 + wave) * 16 + drow) * 32 + dchunk * 8;
    const int wstep = a.Cout * 32;               // elements between consecutive K-steps

    auto issue_x = [&](int chunk) {
        char* dst = xbase + (chunk & 1) * XBUF;
#pragma unroll
        for (int j = 0; j < NXW; ++j) {
            const void* src = xsrc[j] >= 0 ? (const void*)(Xb + (xsrc[j] + chunk * 32)) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(dst + (j * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    auto issue_w = [&](int t) {
        char* dst = wbase + (t % NW) * WBUF;
        const bf16_t* wt = a.Wp + (long long)t * wstep;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(wt + wsrc[j]), VT_LDS_PTR(dst + (j * 8 + wave) * 1024), 16, 0, 0);
    };

    // ---- fragment addressing
    const int fr = lane & 15, fq = lane >> 4;
    // W fragment i: row wc*64 + i*16 + fr, swz = ((fr >> 2) & 1) << 1
    const int wfoff = (wc * 64 + fr) * HB + ((fq ^ (((fr >> 2) & 1) << 1)) << 4);
    // X fragment j at tap (dy,dx): halo row hr = R + fr, R = wp*8*18 + (j+dy)*18 + dx (wave-uniform).
    // The swizzle bit of row hr is bit 2 of (R + fr), a function of (R & 7, fr) only, and R & 7 =
    // ((j+dy)*18 + dx) & 7 is a compile-time constant (wp*144 is a multiple of 8).  So 8 per-lane base
    // addresses (one per value of R & 7) plus ds_read immediate offsets cover all 72 (tap, j) reads.
    int xsel[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        xsel[k] = (wp * 8 * HWID + fr) * HB + ((fq ^ ((((k + fr) >> 2) & 1) << 1)) << 4);
    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: X(0), then W(0..NW-2)
    issue_x(0);
#pragma unroll
    for (int t = 0; t < NW - 1; ++t)
        if (t < nk) issue_w(t);

    // One chunk = 9 K-steps (taps).  LAST = the final chunk: no next halo, weight ring drains, so the
    // wait count is computed at run time; every other chunk uses compile-time s_waitcnt immediates.
    auto do_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* xs = xbase + (chunk & 1) * XBUF;
        const int tbase = chunk * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t = tbase + tap;
            // younger-than-W(t) DMA ops of this wave: W(t+1 .. t+NW-2) and, for taps 1..NW-1, the next
            // chunk's halo (issued at tap 0).  Everything older -- W(t) and this chunk's halo -- has
            // landed once vmcnt <= that count (VM ops retire in order).
            if constexpr (!LAST) {
                constexpr int n_in = (NW - 2) * WPW + NXW, n_out = (NW - 2) * WPW;
                if (tap >= 1 && tap <= NW - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_in) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_out) : "memory");
            } else {
                int ahead = nk - 1 - t;
                if (ahead > NW - 2) ahead = NW - 2;
                wait_vmcnt(ahead * WPW);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();        // all waves' pieces of W(t) (and X(chunk)) are in LDS;
            asm volatile("" ::: "memory");       // everyone is done reading stage (t-1) % NW
            if (!LAST || t + NW - 1 < nk) issue_w(t + NW - 1);
            if (!LAST && tap == 0) issue_x(chunk + 1);

            const char* ws = wbase + (t % NW) * WBUF + wfoff;
            const int dy = tap / 3, dx = tap % 3;
            bf16x8 wf[TC], xf[TP];
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *(const bf16x8*)(ws + i * 16 * HB);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int rel = (j + dy) * HWID + dx;                 // compile-time after unrolling
                xf[j] = *(const bf16x8*)(xs + xsel[rel & 7] + rel * HB);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) do_chunk(chunk, std::false_type{});
    do_chunk(nchunk - 1, std::true_type{});

    // ---- epilogue: lane holds couts cg..cg+3 of pixel (y, x) for every (i, j)
    const int HWp = a.H * a.W;
    const long long ob = (long long)b * HWp * a.Cout;
    const int x = tx0 + fr;
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int y = ty0 + wp * 8 + j;
        if (y >= a.H || x >= a.W) continue;
        valid |= 1u << j;
        const long long p = (long long)y * a.W + x;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int cg = c0 + wc * 64 + i * 16 + fq * 4;
            f32x4 v = acc[i][j];
            if (a.bias) v += *(const f32x4*)(a.bias + cg);
            const long long o = ob + p * a.Cout + cg;
            if (a.res) v += *(const f32x4*)(a.res + o);
            if (a.out_f32) *(f32x4*)(a.out_f32 + o) = v;
            if (a.out_bf16) {
                bf16x4 h;
                h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
                *(bf16x4*)(a.out_bf16 + o) = h;
            }
            acc[i][j] = v;
        }
    }
    if (a.gn_partial) {
        // GroupNorm statistics of this tile's outputs for the NEXT layer's norm (replaces a full read pass)
        __syncthreads();                                   // every wave is done with the staging LDS
        const int G = a.Cout / a.gn_cpg;
        float* out = a.gn_partial + (((long long)b * (tiles_x * tiles_y) + tile) * G + c0 / a.gn_cpg) * 3;
        vt_gn_epilogue_partials<TC, TP>(acc, valid, a.gn_cpg, wp, WP, wc * 64, BC, (float*)smem, out);
    }
}

template <int WP, int WC>
hipError_t launch(const Conv3x3Args& a, hipStream_t s) {
    constexpr int ROWS = WP * 8, BC = WC * 64;
    constexpr int HROWS = (ROWS + 2) * HWID;
    constexpr int NXW = (HROWS + 127) / 128;
    constexpr int smem = 2 * NXW * 128 * HB + NW * BC * HB;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto kern = conv3x3_halo_kernel<WP, WC>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long long tiles = (long long)((a.W + TW - 1) / TW) * ((a.H + ROWS - 1) / ROWS);
    const long long nblk = tiles * (a.Cout / BC) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), smem, s, a);
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
    wp[(((long long)(ci >> 5) * 9 + tap) * Cout + co) * 32 + (ci & 31)] = w[i];
}

}  // namespace

hipError_t vt_launch_repack_ohwi_to_halo(const bf16_t* w, bf16_t* wp, int Cin, int Cout, hipStream_t s) {
    const long long n = (long long)Cout * 9 * Cin;
    hipLaunchKernelGGL(repack_ohwi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cin, Cout);
    return hipGetLastError();
}

int vt_conv3x3_halo_tiles(int H, int W, int Cout) {
    const int rows = (Cout % 256) == 0 ? 16 : 32;
    return ((W + TW - 1) / TW) * ((H + rows - 1) / rows);
}

bool vt_conv3x3_halo_supported(int Cin, int Cout) { return Cin >= 32 && (Cin % 32) == 0 && (Cout % 128) == 0; }

int vt_conv3x3_halo_config(const Conv3x3Args& a) { return (a.Cout % 256) == 0 ? 4 : 3; }   // profile slots 3, 4

hipError_t vt_launch_conv3x3_halo(const Conv3x3Args& a, hipStream_t s) {
    if (!a.X || !a.Wp || !a.zeros || (!a.out_f32 && !a.out_bf16)) return hipErrorInvalidValue;
    if (!vt_conv3x3_halo_supported(a.Cin, a.Cout) || a.batch <= 0 || a.H <= 0 || a.W <= 0) return hipErrorInvalidValue;
    if (a.gn_partial && a.gn_cpg != 4 && a.gn_cpg != 8 && a.gn_cpg != 16) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;        // 32-bit per-image offsets
    if ((long long)(a.Cin / 32) * 9 * a.Cout * 32 >= (1LL << 31)) return hipErrorInvalidValue;
    if ((a.Cout % 256) == 0) return launch<2, 4>(a, s);     // 16x16 px x 256 couts
    return launch<4, 2>(a, s);                              // 32x16 px x 128 couts
}
