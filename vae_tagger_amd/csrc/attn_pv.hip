// P.V of the VAE mid-block attention (1 head, d = 512) with P consumed in the MFMA fragment order attn_qk.hip produces it in.
//
// attn_qk's epilogue holds the probabilities of a (32-query slab, 64-key tile) as four 16-B pieces per lane -- piece (j, h) of
// lane (fq, fr) = query row 16 j + fr, keys 32 h + 8 fq .. + 7 -- which is exactly the B operand of v_mfma_f32_16x16x32_bf16 for
// out^T[c][q] = sum_k v^T[c][k] P[q][k].  So P is stored as it stands (1 KB contiguous per store instruction, no lane permute, a
// wave's stream over the key tiles sequential in memory) and this kernel loads it straight into registers: P never passes through
// LDS, only the v^T tile does (256 channels x 64 keys = 32 KB per key tile, half of what a 256x256 GEMM tile streams per 64 keys; 128 channels on small grids),
// and a wave's 32 x 256 output tile stays in 128 accumulator registers for the whole sweep.
//   workgroup = 8 waves = 256 query rows x 256 channels (grid: query blocks x 2 channel halves x images; the two halves of a
//   query block are neighbours in launch order, so the second read of P hits L2); LDS: two v^T tiles (64 KB); one workgroup per CU (196 VGPRs x 8 waves).
// Rows of the v^T tile are permuted (interleaved cout map of conv_gemm.hip) so that a lane ends up with 16 consecutive channels.
#include <hip/hip_runtime.h>

#include "vt_common.h"
#include "vt_kernels.h"

namespace {

constexpr int QB = 256;        // query rows per workgroup (8 waves x 32)
constexpr int KT = 64;         // keys per tile
constexpr int DCH = 512;       // channels (head dim)
constexpr int ROWB = KT * 2;   // bytes per LDS row (one channel, 64 keys)

// CB = channels per workgroup: 256 (two workgroups per query block), or 128 for small grids -- batch 1 at 1024^2 has 64 query
// blocks, i.e. 128 workgroups of the 256-channel form on 256 CUs; four 128-channel workgroups per block fill the chip
template <int CB>
__global__ __launch_bounds__(512, 2) void attn_pv_kernel(const AttnPvArgs a) {
    constexpr int VBUF = CB * ROWB;                    // 32 KB (16 KB)
    constexpr int NCT = CB / 16;                       // 16-channel MFMA tiles per wave
    constexpr int NCP = DCH / CB;                      // channel parts per query block
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 v^T tiles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int qblocks = (a.S + QB - 1) / QB;
    const int logical = vt_xcd_remap(blockIdx.x, gridDim.x);
    const int cb = logical % NCP;                      // channel part: neighbours share the P stream
    const int rest = logical / NCP;
    const int b = rest / qblocks, qb = rest - b * qblocks;
    const int nkt = (a.S + KT - 1) / KT;
    const int nslab = (a.S + 31) / 32;
    const int slab = qb * 8 + wave;
    const bool slab_real = slab < nslab;
    // P fragments of (slab, kt): 4 pieces x 64 lanes x 16 B
    const bf16_t* pt = a.Pt + (long long)b * a.pt_bs + (long long)(slab_real ? slab : 0) * vt_attn_pt_slab_stride(a.S) + lane * 8;
    const bf16_t* vb = a.vt + (long long)b * a.vt_bs + (long long)cb * CB * a.ldv;
    const int ld8 = (a.S + 7) / 8 * 8;                 // keys [S, ld8) of v^T are written as zero; beyond: the zero page

    // ---- v^T tile staging: one wave-instruction = 8 LDS rows x 128 B; lane l -> row (l >> 3), physical chunk (l & 7),
    // logical chunk = physical ^ (row & 7); LDS row R holds channel (R & ~63) + (R & 3) + 4 ((R >> 4) & 3) + 16 ((R >> 2) & 3)
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < CB / 64; ++jj) {
            const int R = (jj * 8 + wave) * 8 + lrow;
            const int ch = (R & ~63) + (R & 3) + 4 * ((R >> 4) & 3) + 16 * ((R >> 2) & 3);
            const int key = kt * KT + lchunk * 8;
            const void* src = key < ld8 ? (const void*)(vb + (long long)ch * a.ldv + key) : a.zeros;
            __builtin_amdgcn_global_load_lds(VT_GLOBAL_PTR(src), VT_LDS_PTR(smem + buf * VBUF + (jj * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    // A fragment (channel tile ct, key half h): LDS row ct*16 + fr, logical chunk 4 h + fq -> physical ^ (row & 7) = ^ (fr & 7)
    const int foff[2] = {fr * ROWB + (((0 + fq) ^ (fr & 7)) << 4), fr * ROWB + (((4 + fq) ^ (fr & 7)) << 4)};

    f32x4 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 pf[2][2], pn[2][2];
    auto load_p = [&](int kt, bf16x8 (&d)[2][2]) __attribute__((always_inline)) {
        const bf16_t* s = pt + (long long)kt * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
#ifdef PFRAG_PLAIN
            for (int h = 0; h < 2; ++h) d[j][h] = *(const bf16x8*)(s + (j * 2 + h) * 512);
#else
            for (int h = 0; h < 2; ++h) d[j][h] = __builtin_nontemporal_load((const bf16x8*)(s + (j * 2 + h) * 512));
#endif
    };
    stage(0, 0);
    load_p(0, pf);
    // (splitting the DMA / P-load issue between the two waves of a SIMD, which gained 13 % in attn_qk.hip, cost 10 % here:
    //  3.76 -> 4.17 ms per step; with two v^T buffers a late-issued tile does not land before the next barrier.  A ring of four
    //  32-key stages with counted vmcnt waits and one raw barrier per stage -- the halo kernel's scheme -- ran 3.80 -> 5.33 ms:
    //  32 MFMAs per barrier are too few for eight waves)
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                               // vmcnt(0) + barrier: tile kt (and pf) landed, buffer (kt+1)&1 free
        if (kt + 1 < nkt) {
            stage(kt + 1, (kt + 1) & 1);
            load_p(kt + 1, pn);
        }
        const char* vs = smem + (kt & 1) * VBUF;
        // v^T fragments through a ring of eight register sets, read AHEAD fragments before the MFMAs that use them: left to the
        // compiler every ds_read_b128 sat directly in front of its two MFMAs behind an s_waitcnt lgkmcnt(0), i.e. one exposed LDS
        // round trip per 32 matrix-pipe cycles (56 % MFMA busy, 42 % of the wave cycles parked; now 60 % / 34 %, profiles/r02/mfma_util_*.txt)
        constexpr int AHEAD = 6;
        bf16x8 af[8];
        auto frag = [&](int idx) __attribute__((always_inline)) { return *(const bf16x8*)(vs + (idx % NCT) * 16 * ROWB + foff[idx / NCT]); };
#pragma unroll
        for (int p = 0; p < AHEAD; ++p) af[p] = frag(p);
#pragma unroll
        for (int idx = 0; idx < 2 * NCT; ++idx) {
            if (idx + AHEAD < 2 * NCT) af[(idx + AHEAD) & 7] = frag(idx + AHEAD);
            const int h = idx / NCT, ct = idx % NCT;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[idx & 7], pf[j][h], acc[ct][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);         // keep the read-ahead distance: the scheduler would sink the reads to their uses
        }
        if (kt + 1 < nkt) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) pf[j][h] = pn[j][h];
        }
    }
    // ---- epilogue: lane (fq, fr) holds, for query row 16 j + fr of its slab and every 64-channel group G, the 16 consecutive
    // channels 64 G + 16 fq + 4 i + r (tile ct = 4 G + i, register r): scale by 1 / row sum, two 16-B bf16 stores per group
    if (!slab_real) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = slab * 32 + j * 16 + fr;
        if (row >= a.S) continue;
        const float* ps = a.rsum + (long long)b * a.row_bs + row;       // segment sums of attn_qk, added in a fixed order (its comment)
        const float rs = 1.f / (((ps[0] + ps[a.split_stride]) + ps[2 * a.split_stride]) + ps[3 * a.split_stride]);
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

}  // namespace

bool vt_attn_pv_supported(int S, int C) { return C == DCH && S > 0; }
// elements of the fragment-ordered P of one image
long long vt_attn_pt_elems(int S) { return (long long)((S + 31) / 32) * vt_attn_pt_slab_stride(S); }

hipError_t vt_launch_attn_pv(const AttnPvArgs& a, hipStream_t s) {
    if (!a.Pt || !a.vt || !a.rsum || !a.o || !a.zeros || a.batch <= 0 || !vt_attn_pv_supported(a.S, a.C)) return hipErrorInvalidValue;
    if ((a.ldv % 8) || (a.vt_bs % 8) || (a.ldo % 8) || (a.o_bs % 8) || (a.pt_bs % 8) || a.row_bs < a.S) return hipErrorInvalidValue;
    if (a.ldv < (a.S + 7) / 8 * 8 || (long long)a.C * a.ldv >= (1LL << 31)) return hipErrorInvalidValue;
    if (a.split_stride < (long long)a.batch * a.row_bs) return hipErrorInvalidValue;
    const long long qblk = (long long)((a.S + QB - 1) / QB) * a.batch;
    const bool narrow = qblk * 2 < 192;                 // the 256-channel form would leave a quarter or more of the CUs without a workgroup
    const long long nblk = qblk * (narrow ? 4 : 2);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_done{0};
    hipError_t ea = vt_once_per_device(attr_done, [&] {
        hipError_t e = hipFuncSetAttribute((const void*)attn_pv_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * ROWB);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_pv_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * ROWB);
        return e;
    });
    if (ea != hipSuccess) return ea;
    if (narrow) hipLaunchKernelGGL(attn_pv_kernel<128>, dim3((unsigned)nblk), dim3(512), 2 * 128 * ROWB, s, a);
    else hipLaunchKernelGGL(attn_pv_kernel<256>, dim3((unsigned)nblk), dim3(512), 2 * 256 * ROWB, s, a);
    return hipGetLastError();
}
