// conv_in (3 -> C0 channels, fp32 direct) and the row softmax of the unfused attention path.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// conv_in: x fp32 NCHW [B,3,H,W] (the tensor the reference hands to vae.encode, infer_full.py:98)
// -> NHWC rows [B][H*W][Cout].  K = 27 is too thin for MFMA and the op is bound by its 128-channel
// output write, so it is a direct fp32 VALU conv: exact fp32 inputs/weights (no bf16 rounding of the
// image), 64 pixels x Cout per workgroup, weights + 3-row input halo in LDS.
// packed weight layout: wp[k][cout], k = ci*9 + ky*3 + kx.
constexpr int CI_PIX = 64;
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                      const float* __restrict__ bias, float* __restrict__ o32,
                                                      bf16_t* __restrict__ o16, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sw = sm;                       // [27][Cout]
    float* sin = sm + 27 * Cout;          // [3 ch][3 rows][CI_PIX + 2]
    const int tid = threadIdx.x;
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * CI_PIX;
    for (int i = tid; i < 27 * Cout; i += 256) sw[i] = wp[i];
    constexpr int RW = CI_PIX + 2;
    for (int i = tid; i < 9 * RW; i += 256) {
        const int c = i / (3 * RW), r = (i / RW) % 3, xx = i % RW;
        const int iy = y - 1 + r, ix = x0 - 1 + xx;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long long)b * 3 + c) * H + iy) * W + ix];
        sin[i] = v;
    }
    __syncthreads();
    const int cgi = tid & 15, ps = tid >> 4;          // 16 cout-groups of 8, 16 pixel slots of 4
    for (int cg = cgi; cg * 8 < Cout; cg += 16) {
        float acc[4][8];
        {
            const f32x4 b0 = *(const f32x4*)(bias + cg * 8), b1 = *(const f32x4*)(bias + cg * 8 + 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acc[p][0] = b0[0]; acc[p][1] = b0[1]; acc[p][2] = b0[2]; acc[p][3] = b0[3];
                acc[p][4] = b1[0]; acc[p][5] = b1[1]; acc[p][6] = b1[2]; acc[p][7] = b1[3];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float in6[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) in6[i] = sin[(c * 3 + ky) * RW + ps * 4 + i];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* wr = sw + (c * 9 + ky * 3 + kx) * Cout + cg * 8;
                    const f32x4 w0 = *(const f32x4*)wr, w1 = *(const f32x4*)(wr + 4);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float v = in6[p + kx];
                        acc[p][0] = fmaf(v, w0[0], acc[p][0]); acc[p][1] = fmaf(v, w0[1], acc[p][1]);
                        acc[p][2] = fmaf(v, w0[2], acc[p][2]); acc[p][3] = fmaf(v, w0[3], acc[p][3]);
                        acc[p][4] = fmaf(v, w1[0], acc[p][4]); acc[p][5] = fmaf(v, w1[1], acc[p][5]);
                        acc[p][6] = fmaf(v, w1[2], acc[p][6]); acc[p][7] = fmaf(v, w1[3], acc[p][7]);
                    }
                }
            }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int px = x0 + ps * 4 + p;
            if (px >= W) continue;
            const long long o = (((long long)b * H + y) * W + px) * Cout + cg * 8;
            if (o32) {
                *(f32x4*)(o32 + o) = f32x4{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
                *(f32x4*)(o32 + o + 4) = f32x4{acc[p][4], acc[p][5], acc[p][6], acc[p][7]};
            }
            if (o16) {
                bf16x8 h;
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = (bf16_t)acc[p][i];
                *(bf16x8*)(o16 + o) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax, fp32 in -> bf16 out (P operand of the P.V GEMM).  One workgroup per row; the row
// (<= 64 KiB) is L2-resident across the three sweeps.  Pad columns [n, ldp) are written as zeros so
// the GEMM's 8-element k-chunks never see garbage.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                           int n, int lds, int ldp) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    const float* sr = s + row * lds;
    bf16_t* pr = p + row * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, sr[i]);
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += __expf(sr[i] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    for (int i = tid; i < ldp; i += 256) pr[i] = (bf16_t)(i < n ? __expf(sr[i] - m) * inv : 0.f);
}

}  // namespace

hipError_t vt_launch_conv_in(const float* x, const float* wp, const float* bias, float* o32, bf16_t* o16, int B,
                             int H, int W, int Cout, hipStream_t s) {
    if (!x || !wp || !bias || (!o32 && !o16) || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 8))
        return hipErrorInvalidValue;
    const size_t smem = (size_t)(27 * Cout + 9 * (CI_PIX + 2)) * sizeof(float);
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid((W + CI_PIX - 1) / CI_PIX, H, B);
    hipLaunchKernelGGL(conv_in_kernel, grid, dim3(256), smem, s, x, wp, bias, o32, o16, H, W, Cout);
    return hipGetLastError();
}

hipError_t vt_launch_softmax_rows(const float* scores, bf16_t* probs, int rows, int n, int lds, int ldp,
                                  hipStream_t s) {
    if (!scores || !probs || rows <= 0 || n <= 0 || lds < n || ldp < n) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, scores, probs, n, lds, ldp);
    return hipGetLastError();
}
