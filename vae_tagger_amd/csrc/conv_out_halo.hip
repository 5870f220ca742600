// conv_out of the encoder (Conv2d(512, 32, 3, pad 1) on the 128 x 128 mid-block output; the moments / mode() epilogue of
// diffusers_vae_loader.py:78-86) as a halo-tile implicit GEMM with a 32-cout tile (round 4).
//
// On the generic GEMM this launch fetched 2.4 GB for 0.27 GB of input (a 128 x 32 tile sweeps X once per tap: 8.9x) and ran at 0.24 PF
// (0.33 ms per step).  Padding Cout to the 128-cout halo tile would trade the over-fetch for 4x the MFMAs.  Here the halo of a 16 x 16-pixel
// tile is staged ONCE per 32-channel chunk (18 x 18 pixels x 64 B) together with the chunk's nine 32-cout weight tiles (18 KB), double
// buffered, by LDS-DMA; all nine taps read it at shifted rows; one s_barrier per chunk.
//   workgroup = 4 waves, each 4 rows x 16 px x 32 couts = 4 x 2 v_mfma_f32_16x16x32 per tap, 78 KB LDS: two workgroups per CU;
//   per kx the six halo rows a wave needs are read once and serve the three ky from registers;
//   LDS rows are 64 B with the halo kernels' chunk swizzle (physical 16-B chunk = logical ^ ((row >> 2) & 1) << 1), applied to the DMA's
//   per-lane source address and to the fragment reads;
//   epilogue: + bias, * post_scale + post_shift, fp32 NCHW stores of the first `keep` channels (16 lanes = 64 contiguous bytes per row).
// F16: the operands hold fp16 bits (the fp16-operand mode of the convs, vt_set_flag 18).
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int HB = 64;                       // bytes per LDS row (32 x 16-bit channels)
constexpr int TW = 16, TH = 16;              // output tile
constexpr int HWID = TW + 2;
constexpr int NWV = 4, NT = 64 * NWV, TP = TH / NWV;     // 4 rows per wave
constexpr int HPIX = (TH + 2) * HWID;        // 324 halo pixels
constexpr int XPCS = (HPIX + 15) / 16;       // 21 DMA pieces of 16 rows
constexpr int XBUF = XPCS * 16 * HB;         // 21 504 B
constexpr int CO = 32;                       // couts
constexpr int WPCS = 9 * CO / 16;            // 18 pieces: nine taps x 32 cout rows
constexpr int WBUF = WPCS * 16 * HB;         // 18 432 B
constexpr int PCS = XPCS + WPCS;             // 39 pieces per chunk
constexpr int PPW = (PCS + NWV - 1) / NWV;   // 10 per wave (the last wave issues 9)
constexpr int SMEM = 2 * (XBUF + WBUF);      // 79 872 B
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) {
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
}

template <bool F16>
__global__ __launch_bounds__(NT, 2) void conv_out_halo_kernel(const ConvOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / per_img, t = logical - b * per_img;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const bf16_t* Xb = a.X + (long long)b * a.H * a.W * a.Cin;
    const int nchunk = a.Cin >> 5;

    // ---- staging: one wave instruction = 16 LDS rows x 64 B; lane l -> row (l >> 2), physical chunk (l & 3), logical chunk = physical ^ swz(row)
    const int drow = lane >> 2;
    auto stage = [&](int chunk, int buf) {
        char* xd = smem + buf * (XBUF + WBUF);
        char* wd = xd + XBUF;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int piece = j * NWV + wave;
            if (piece >= PCS) break;
            if (piece < XPCS) {
                const int hr = piece * 16 + drow;                       // halo pixel index (row-major 18 x 18); >= 324: padding rows, never read
                const int hy = hr / HWID, hx = hr - hy * HWID;
                const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
                const int lch = (lane & 3) ^ (((hr >> 2) & 1) << 1);
                const bool ok = hr < HPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const void* src = ok ? (const void*)(Xb + ((long long)(iy * a.W + ix) * a.Cin + chunk * 32 + lch * 8)) : a.zeros;
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(xd + piece * 1024), 16, 0, 0);
            } else {
                const int wr = (piece - XPCS) * 16 + drow;              // row of the chunk's [9 taps][32 couts] weight block
                const int lch = (lane & 3) ^ (((wr >> 2) & 1) << 1);
                const void* src = (const void*)(a.Wp + ((long long)chunk * 9 * CO + wr) * 32 + lch * 8);
                __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(wd + (piece - XPCS) * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[2][TP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment offsets: row R of a buffer, logical chunk fq -> byte R * 64 + ((fq ^ (((R >> 2) & 1) << 1)) << 4)
    auto frag = [&](const char* base, int R) -> bf16x8 {
        return *(const bf16x8*)(base + R * HB + ((fq ^ (((R >> 2) & 1) << 1)) << 4));
    };

    stage(0, 0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of `chunk` have landed ...
        __builtin_amdgcn_s_barrier();                      // ... everybody's; and everybody has finished reading the other buffer
        asm volatile("" ::: "memory");
        if (chunk + 1 < nchunk) stage(chunk + 1, (chunk + 1) & 1);
        const char* xs = smem + (chunk & 1) * (XBUF + WBUF);
        const char* ws = xs + XBUF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 xr[TP + 2];
#pragma unroll
            for (int r = 0; r < TP + 2; ++r) xr[r] = frag(xs, (wave * TP + r) * HWID + kx + fr);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int tap = ky * 3 + kx;
                const bf16x8 w0 = frag(ws, tap * CO + fr), w1 = frag(ws, tap * CO + 16 + fr);
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    acc[0][j] = mfma16<F16>(w0, xr[j + ky], acc[0][j]);
                    acc[1][j] = mfma16<F16>(w1, xr[j + ky], acc[1][j]);
                }
            }
        }
    }

    // ---- epilogue: register r of acc[i][j] in lane (fq, fr) = cout 16 i + 4 fq + r of pixel (ty0 + wave * TP + j, tx0 + fr)
    const int x = tx0 + fr;
    const long long HWo = (long long)a.H * a.W;
    float* ob = a.out + (long long)b * a.keep * HWo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c0 = 16 * i + 4 * fq;
        const f32x4 bv = a.bias ? *(const f32x4*)(a.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int y = ty0 + wave * TP + j;
            if (y >= a.H || x >= a.W) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + r < a.keep) ob[(long long)(c0 + r) * HWo + (long long)y * a.W + x] = (acc[i][j][r] + bv[r]) * a.post_scale + a.post_shift;
        }
    }
}

}  // namespace

bool vt_conv_out_halo_supported(int Cin, int Cout) { return Cout == CO && Cin >= 32 && (Cin % 32) == 0; }

hipError_t vt_launch_conv_out_halo(const ConvOutArgs& a, hipStream_t s) {
    if (!a.X || !a.Wp || !a.out || !a.zeros || a.batch <= 0 || a.H <= 0 || a.W <= 0 || !vt_conv_out_halo_supported(a.Cin, a.Cout)) return hipErrorInvalidValue;
    if (a.keep <= 0 || a.keep > CO) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * a.Cin >= (1LL << 31)) return hipErrorInvalidValue;          // 32-bit offsets inside an image
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)conv_out_halo_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_out_halo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        return e;
    });
    if (ea != hipSuccess) return ea;
    const long long nblk = (long long)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) * a.batch;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.f16) hipLaunchKernelGGL(conv_out_halo_kernel<true>, dim3((unsigned)nblk), dim3(NT), SMEM, s, a);
    else hipLaunchKernelGGL(conv_out_halo_kernel<false>, dim3((unsigned)nblk), dim3(NT), SMEM, s, a);
    return hipGetLastError();
}
