// Internal launcher interface between the kernel translation units and the C-ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "vt_common.h"

struct ConvGemmArgs {
    const bf16_t* X;        // activations NHWC bf16 / generic row-major A operand
    const bf16_t* W;        // [Cout][taps][Cin] bf16 (k contiguous) / generic row-major B^T operand
    const float* bias;      // fp32, per cout (mode 1) or per pixel-row (mode 2)
    const float* res;       // optional fp32 residual, indexed like the output rows
    const f16_t* res_f16;   // ... or fp16 residual (at most one of res / res_f16)
    float* out_f32;         // optional fp32 output rows [p][ldo]
    bf16_t* out_bf16;       // optional bf16 output rows [p][ldo]
    f16_t* out_f16;         // optional fp16 output rows [p][ldo] (attention scores)
    const void* zeros;      // >= 16 zero bytes in device memory (padding / tail source for the DMA)
    float* gn_partial;      // optional [batch][ptiles][Cout/gn_cpg][3] (n, mean, M2) of the output values
    int Hin, Win, Hout, Wout;
    int Cin, Cout;             // k per tap; output columns written
    int Wrows;                 // rows of W that exist (rows >= Wrows read as zero)
    int ksize, stride, pad;
    int ldx, ldw, ldo, ldr;         // row strides in elements
    long long x_bs, w_bs, o_bs, r_bs;  // batch strides in elements
    int batch;
    float alpha;
    int bias_mode;          // 0 none, 1 per cout, 2 per pixel-row
    int out_mode;           // 0 rows, 1 latent NCHW fp32 (first cout_keep channels, * post_scale + post_shift)
    int cout_keep;
    float post_scale, post_shift;
    int gn_cpg;             // channels per group for gn_partial (0 = off)
    // row softmax pieces of the attention (capi.hip run_attention).  A (row, column slot) is one wave's 16*TC columns of a row:
    // slot = column tile * WC + wave column, vt_conv_gemm_col_slots(a) of them.
    int row_mode;           // 0 off; 1 = no output, row_part = max over the slot of alpha*acc (columns >= Wrows excluded);
                            // 2 = the output is exp(alpha*acc - row_in[p]) (columns >= Wrows: 0), row_part = the slot's sum
                            //     of the fp32 values; 3 = output rows are multiplied by row_in[p]
    const float* row_in;    // [batch][row_bs]
    float* row_part;        // [batch][slots][row_bs]
    long long row_bs;
    const int* gate;        // optional: the whole launch is a no-op unless *gate == gate_expect
    int gate_expect;
    int x_stream;           // 1 = X is read once (P of P.V): its LDS-DMA carries the streaming (nt) cache policy
    int f8;                 // 1 = X and W are e4m3 bytes (ld* / *_bs in elements = bytes); fp8 MFMA, twice the K per K-step
    int f16;                // 1 = X and W hold fp16 bits instead of bf16 (Cout <= 32 tile only: conv_out in the fp16-operand mode, vt_set_flag 18)
    const float* col_scale; // optional per-cout multiplier of the accumulator (before the bias)
    int short_tiles;        // 1 = short-K launches and the 128-cout stride-2 conv use the two-workgroups-per-CU tile (vt_set_flag 6)
};
int vt_conv_gemm_col_slots(const ConvGemmArgs& a);

hipError_t vt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t s);
// which tile configuration the dispatcher picks for these args: 0 = 128x32, 1 = 256x128, 2 = 256x256
int vt_conv_gemm_config(const ConvGemmArgs& a);
const char* vt_conv_gemm_config_name(int cfg);
// profile slots: 0..2 conv_gemm tiles, 3..8 conv3x3_halo variants, 9 conv_gemm two-workgroups-per-CU tile, 10 attn_qk, 11 fp8 halo conv
constexpr int VT_PROF_ATTN_QK = 10;
constexpr int VT_PROF_HALO_FP8 = 11;
constexpr int VT_PROF_ATTN_PV = 12;
constexpr int VT_PROF_GEMM_FP8 = 13;
constexpr int VT_PROF_S2_HALO = 14;      // stride-2 phase-plane halo conv (conv3x3_s2_halo.hip)
constexpr int VT_PROF_ATTN_QK8 = 15;     // fp8 Q.K^T / P.V (attn_fp8.hip)
constexpr int VT_PROF_ATTN_PV8 = 16;
constexpr int VT_PROF_S2_HALO_FP8 = 17;  // stride-2 phase-plane conv on e4m3 operands (conv3x3_s2_halo_fp8.hip)
constexpr int VT_PROF_PROJ_FP8 = 18;     // fp8 mode's q | k and v projections (attn_fp8.hip, proj_fp8_kernel)
constexpr int VT_PROF_HALO_FP8_C128 = 19; // the fp8 halo conv's launches with Cin <= 128 (18 K-steps per tile), same kernel name as slot 11: tools read them apart
constexpr int VT_PROF_PROJ_BF16 = 20;    // bf16 q | k and v^T projections on attn_qk_kernel<4> (round 4)
constexpr int VT_PROF_CONV_OUT = 21;     // conv_out on its 32-cout halo tile (conv_out_halo.hip, round 4)
constexpr int VT_PROF_GN_APPLY = 22;     // HBM-bound GroupNorm(+SiLU) apply pass: 'flops' slot carries algorithmic BYTES (last slot)
constexpr int VT_NUM_PROF_SLOTS = 23;

// Q.K^T of the mid-block attention with the softmax numerators in the epilogue (attn_qk.hip; d = 512 only)
struct AttnQkArgs {
    const bf16_t* q; const bf16_t* k;   // row-major [S][ldq] bf16 (the q | k buffer: ldq = 2 C), batch stride qk_bs
    int S, C, ldq; long long qk_bs;
    bf16_t* P; int ldp; long long p_bs; // mode 2: [S][ldp] bf16 out, columns [S, ldp) written as 0
    int p_frag;                         // mode 2: 1 = P is written in MFMA fragment order for attn_pv.hip (ldp unused, p_bs = vt_attn_pt_elems)
    const float* rowin;                 // mode 2: per-row exponent shift [batch][row_bs]
    float* rowout;                      // mode 1: row maxima of alpha q.k; mode 2: 1 / row sum   [batch][row_bs]
    long long row_bs;
    float alpha;
    int mode;                           // 1 or 2
    int batch;
    const void* zeros;                  // >= 16 zero bytes on the device
    const int* gate; int gate_expect;   // optional: the launch is a no-op unless *gate == gate_expect
    // fragment-order mode (p_frag): rowout receives FOUR segment sums per row, [4][split_stride] (split_stride >= batch * row_bs) --
    // key tiles [i n / 4, (i + 1) n / 4) -- which attn_pv adds in a fixed order and inverts; nsplit = 2 or 4 spreads the segments of a
    // query block over that many workgroups (small grids: batch 1 at 1024^2 has 64 query blocks for 256 CUs) without changing a bit
    int nsplit; long long split_stride;
    // mode 4 (round 4): the same skeleton as a LINEAR layer of K = 512 -- P[b][row][key] = bf16(alpha * q[row] . k[key] + kbias[key] + qbias[row]),
    // rows = S rows of q (batch stride qk_bs), keys = nk rows of k [nk][ldk] (batch stride k_bs; 0 = shared weights), columns [nk, ldp rounded
    // to the 64-key tile) written as 0; nsplit (any value >= 1) spreads a row block's key tiles over that many workgroups.  The attention's
    // q | k projection (rows = tokens, keys = [Wq; Wk]) and v^T projection (rows = Wv, keys = tokens) run on it instead of the generic GEMM.
    int nk, ldk; long long k_bs; const float* kbias; const float* qbias;
    // mode 5 (round 4): mode 4's linear layer with the attention's to_out epilogue -- out[b][row][key] = alpha * q[row] . k[key] + kbias[key] +
    // residual[b][row][key], stored as fp16 (res_f16 / out_f16: the fp16 residual stream) or fp32 (res_f32 / out_f32), row pitch ldp, batch stride
    // p_bs -- plus GroupNorm (n, mean, M2) partials of what was stored: one triple per (32-row slab, group of 16 keys),
    // gn_partial[((b * gn_parts + slab) * (nk / 16) + group) * 3], gn_parts = vt_attn_linear_parts(S) slabs per image
    const f16_t* res_f16; const float* res_f32; f16_t* out_f16; float* out_f32; float* gn_partial; int gn_parts;
};
int vt_attn_linear_parts(int S);
bool vt_attn_qk_supported(int S, int C);
hipError_t vt_launch_attn_qk(const AttnQkArgs& a, hipStream_t s);

// P.V with P in attn_qk's fragment order (attn_pv.hip; d = 512 only): o[q][c] = rinv[q] * sum_k P[q][k] v^T[c][k]
struct AttnPvArgs {
    const bf16_t* Pt; long long pt_bs;  // fragment-ordered P: [slab of 32 queries][64-key tile][piece j*2+h][lane][8] bf16
    const bf16_t* vt; int ldv; long long vt_bs;   // v^T [C][ldv] bf16 (keys contiguous; keys [S, round8(S)) zero)
    const float* rsum; long long row_bs;          // attn_qk's four segment sums per row, [4][split_stride] of [batch][row_bs]
    long long split_stride;
    bf16_t* o; int ldo; long long o_bs;           // [S][ldo] bf16
    int S, C, batch;
    const void* zeros;
};
bool vt_attn_pv_supported(int S, int C);
// elements between consecutive 32-query slabs of the fragment-ordered P: key tiles x 4 pieces x 512, plus 2304 B so that the
// eight waves of a workgroup (same key tile, consecutive slabs) do not all hit one HBM channel at a power-of-two stride
__host__ __device__ inline long long vt_attn_pt_slab_stride(int S) { return (long long)((S + 63) / 64) * 2048 + 1152; }
long long vt_attn_pt_elems(int S);
hipError_t vt_launch_attn_pv(const AttnPvArgs& a, hipStream_t s);

// ---- the same two contractions on fp8 (e4m3) operands (attn_fp8.hip; vt_set_flag 11 + 14)
struct AttnQk8Args {
    const unsigned char* qk8; int ldq; long long qk_bs;   // [S][ldq] e4m3 bytes: q8 (C bytes) | k8 (C bytes) per row, batch stride qk_bs
    int S, C;
    unsigned char* P8; long long p_bs;                    // fragment-ordered e4m3 P: vt_attn_p8_bytes(S) per image
    const float* rowin;                                   // per-row exponent shift [batch][row_bs]
    float* rowout;                                        // mode 1: row maxima [batch][row_bs]
    long long row_bs;
    float alpha;                                          // scale applied to the e4m3 dot product (1 / (sqrt(C) * qscale * kscale))
    float pscale_log2;                                    // mode 3: P8 = e4m3(2^pscale_log2 * exp(alpha q8.k8 - rowin))
    int mode;                                             // 1: rowout[batch][row_bs] = row maxima of alpha q8.k8, no P; 3: P8 (P.V takes the row sums itself)
    int kstride;                                          // mode 1: sweep every kstride-th 128-key tile only (0 / 1 = all): a sampled maximum
    int* flag;                                            // mode 3, optional: bit 0 is raised when a numerator exceeded 448 and was clamped
    const int* gate; int gate_expect;                     // optional: the launch is a no-op unless *gate == gate_expect
    int batch, nsplit;
    const void* zeros;
};
// Linear projections of the fp8 mode (attn_fp8.hip, proj_fp8_kernel): out8[q][k] = e4m3(clamp(oscale * (alpha * q8[q] . k8[k] + kbias[k] + qbias[q])))
struct ProjFp8Args {
    const unsigned char* q8; int ldq; long long q_bs; int nq;   // rows held in registers: [nq][ldq] e4m3, C k-bytes each (batch stride may be 0)
    const unsigned char* k8; int ldk; long long k_bs; int nk;   // rows streamed through LDS: [nk][ldk]
    unsigned char* out8; int ldo; long long o_bs;               // [nq][ldo] e4m3; columns [nk, kext) are written as zero, kext <= ldo, kext % 16 == 0
    int kext;
    const float* kbias;                                         // optional, per k row
    const float* qbias;                                         // optional, per q row
    float alpha, oscale;
    int* status;                                                // optional: bit 1 (VT_STATUS_FP8_SATURATED) when a value met the +-448 clamp
    int C, batch, nsplit;                                       // nsplit: the k tiles of a query block over this many workgroups (small grids)
    const void* zeros;
};
hipError_t vt_launch_proj_fp8(const ProjFp8Args& a, hipStream_t s);
struct AttnPv8Args {
    const unsigned char* P8; long long p_bs;
    const unsigned char* vt8; int ldv; long long vt_bs;   // v^T [C][ldv] e4m3 (keys contiguous); keys >= kext are not read (zero page)
    int kext;
    bf16_t* o; int ldo; long long o_bs;                   // [S][ldo] bf16
    float out_scale;                                      // 1 / vscale
    const int* gate; int gate_expect;                     // optional: the launch is a no-op unless *gate == gate_expect
    int S, C, batch;
    const void* zeros;
};
bool vt_attn_fp8_supported(int S, int C);
// bytes between consecutive 32-query slabs of the fragment-ordered e4m3 P: 128-key blocks x 4 KB, plus 2304 B (see vt_attn_pt_slab_stride)
__host__ __device__ inline long long vt_attn_p8_slab_stride(int S) { return (long long)((S + 127) / 128) * 4096 + 2304; }
long long vt_attn_p8_bytes(int S);
hipError_t vt_launch_attn_qk_fp8(const AttnQk8Args& a, hipStream_t s);
hipError_t vt_launch_attn_pv_fp8(const AttnPv8Args& a, hipStream_t s);
hipError_t vt_launch_attn_vt_to_fp8(const bf16_t* vt, long long vt_bs, int ldv, unsigned char* v8, long long v8_bs, int ld8, int S, int kext,
                                    int C, int batch, float scale, int* status, hipStream_t s);
// row norms of q | k as attn_row_norms, taken from the e4m3(scale x) values this kernel also writes (qk8: [rows][2C] bytes)
hipError_t vt_launch_attn_row_norms_fp8(const bf16_t* qk, long long rows, int C, float scale, unsigned char* qk8, float* qn, float* kn, float* sd,
                                        int* status, hipStream_t s);

// 3x3 stride-1 pad-1 conv, halo-tile kernel (conv3x3_halo.hip)
struct Conv3x3Args {
    const bf16_t* X;        // NHWC bf16 [batch][H][W][Cin]          (exactly one of X / Xf32)
    const float* Xf32;      // NHWC fp32 [batch][H][W][Cin]: only with scale_shift
    const float* scale_shift;  // optional [batch][Cin][2]: the input is silu(x*scale + shift), fused into staging
    const bf16_t* Wp;       // packed [Cin/32][9][Cout][32] bf16
    const float* bias;      // [Cout] or null
    const float* res;       // optional fp32 residual [batch][H][W][Cout]
    const f16_t* res_f16;   // ... or fp16 residual (at most one of res / res_f16)
    float* out_f32;         // optional
    bf16_t* out_bf16;       // optional
    f16_t* out_f16;         // optional (fp16 residual stream)
    const void* zeros;
    float* gn_partial;      // optional [batch][tiles][Cout/gn_cpg][3] (n, mean, M2) of the output values
    int gn_cpg;             // channels per GroupNorm group of the OUTPUT (4, 8 or 16)
    int batch, H, W, Cin, Cout;
    // optional fused 1x1 conv (a resnet block's conv_shortcut): out += scW . scX at the centre tap, as scCin / 32 extra
    // K-steps after the 3x3 loop (raw bf16 input only; its bias is expected inside `bias`)
    const bf16_t* scX;      // NHWC bf16 [batch][H][W][scCin]
    const bf16_t* scW;      // packed [scCin/32][Cout][32] bf16, rows in the interleaved cout order
    int scCin;
    int occ2;               // two-workgroups-per-CU tile mode 0..3 (vt_set_flag 3; see conv3x3_halo.hip)
    int f16;                // 1 = X, Wp, scX, scW hold fp16 bits (v_mfma_f32_16x16x32_f16): the fp16-operand mode, vt_set_flag 18
    int out16_f16;          // 1 = out_bf16 receives fp16 bits (its consumer is a conv in that mode)
    int out16_planar;       // 1 = out_bf16 is laid out [Cout/32][H][W][32] per image (chunk-planar) for the stride-2 phase-plane kernel (x_planar)
    // filled by the launcher: tile-grid constants of the chosen variant and their division multipliers (0 = divide)
    int tiles_x, ctiles, per_img, ptiles;
    unsigned long long m_per_img, m_ctiles, m_tiles_x;
};
// GroupNorm partials per image the epilogue of this (Cout, input mode xt 0..2, occ2 mode, fused shortcut) launch writes
int vt_conv3x3_halo_tiles(int H, int W, int Cout, int xt, int occ2, int has_sc);
int vt_conv3x3_halo_tiles_max(int H, int W);       // upper bound over the variants (buffer sizing)
int vt_conv_gemm_ptiles(int HWo, int Cout);        // upper bound over configurations (buffer sizing)
int vt_conv_gemm_ptiles_of(const ConvGemmArgs& a);  // of this launch (GroupNorm partials per image its epilogue writes)
bool vt_conv3x3_halo_supported(int Cin, int Cout);
bool vt_conv3x3_halo_f16_supported(int Cout, int occ2, int has_sc);
int vt_conv3x3_halo_config(const Conv3x3Args& a);
hipError_t vt_launch_conv3x3_halo(const Conv3x3Args& a, hipStream_t s);
hipError_t vt_launch_repack_ohwi_to_halo(const bf16_t* w_ohwi, bf16_t* wp, int Cin, int Cout, hipStream_t s);

// 3x3 stride-2 conv with Downsample2D's (0,1,0,1) padding, phase-plane halo kernel (conv3x3_s2_halo.hip): out = W . x + bias (+ res)
struct Conv3x3S2Args {
    const bf16_t* X;        // NHWC bf16 [batch][H][W][Cin]
    const bf16_t* Wp;       // packed [Cin/32][9 steps (vt_s2_step_of_tap)][Cout rows (vt_halo_row_of_cout)][32] bf16
    const float* bias;      // [Cout] or null
    const float* res;       // optional fp32 residual [batch][Ho][Wo][Cout]
    float* out_f32; bf16_t* out_bf16; f16_t* out_f16;     // at least one; [batch][Ho][Wo][Cout], Ho = H / 2, Wo = W / 2
    const void* zeros;
    float* gn_partial; int gn_cpg;                         // optional [batch][tiles][Cout/gn_cpg][3]
    int batch, H, W, Cin, Cout;
    int f16, out16_f16;                                    // 1 = X and Wp hold fp16 bits (vt_set_flag 18) / out_bf16 receives fp16 bits
    int x_planar;                                          // 1 = X is chunk-planar [Cin/32][H][W][32] per image instead of NHWC (vt_set_flag 19)
    int Ho, Wo, tiles_x, ctiles, per_img, ptiles;          // filled by the launcher
    unsigned long long m_per_img, m_ctiles, m_tiles_x;
};
bool vt_conv3x3_s2_supported(int Cin, int Cout);
int vt_conv3x3_s2_tiles(int Ho, int Wo);                  // GroupNorm partials per image its epilogue writes
hipError_t vt_launch_conv3x3_s2(const Conv3x3S2Args& a, hipStream_t s);
hipError_t vt_launch_repack_ohwi_to_s2(const bf16_t* w_ohwi, bf16_t* wp, int Cin, int Cout, hipStream_t s);

// the same on fp8 (e4m3) operands (conv3x3_s2_halo_fp8.hip): out = acc * mult[cout] + bias (+ res)
struct Conv3x3S2Fp8Args {
    const unsigned char* X;   // NHWC e4m3 [batch][H][W][Cin] (input scale folded into mult)
    const unsigned char* Wp;  // packed [Cin/64][9 steps (vt_s2_step_of_tap)][Cout rows (vt_halo_fp8_row_of_cout)][64] e4m3
    const float* mult;        // [Cout]
    const float* bias;        // [Cout] or null
    const float* res;         // optional fp32 residual [batch][Ho][Wo][Cout]
    float* out_f32; bf16_t* out_bf16; f16_t* out_f16;
    const void* zeros;
    float* gn_partial; int gn_cpg;
    int batch, H, W, Cin, Cout;
    int x_planar;                                          // 1 = X is chunk-planar [Cin/64][H][W][64] per image instead of NHWC (vt_set_flag 19)
    int Ho, Wo, tiles_x, ctiles, per_img, ptiles;          // filled by the launcher
    unsigned long long m_per_img, m_ctiles, m_tiles_x;
};
bool vt_conv3x3_s2_fp8_supported(int Cin, int Cout);
int vt_conv3x3_s2_fp8_tiles(int Ho, int Wo);
hipError_t vt_launch_conv3x3_s2_fp8(const Conv3x3S2Fp8Args& a, hipStream_t s);

// 3x3 stride-1 pad-1 conv on fp8 (OCP e4m3) operands (conv3x3_halo_fp8.hip): out = acc * mult[cout] + bias[cout] (+ residual)
struct Conv3x3Fp8Args {
    const unsigned char* X;   // NHWC e4m3 [batch][H][W][Cin]: act_scale * activation
    const unsigned char* Wp;  // packed [Cin/64][9 (kx-major)][Cout][64] e4m3, cout rows permuted (vt_halo_fp8_row_of_cout)
    const float* mult;        // [Cout]: weight scale / act_scale
    const float* bias;        // [Cout] or null
    const float* res; const f16_t* res_f16;               // optional residual (at most one)
    float* out_f32; bf16_t* out_bf16; f16_t* out_f16;     // at least one of these four
    unsigned char* out_e4m3; float out_e4m3_scale;        // e4m3(scale * out), saturated: the operand of a following fp8 conv
    int out8_planar;                                       // 1 = out_e4m3 is laid out [Cout/64][H][W][64] per image (chunk-planar; the fp8 stride-2 kernel's x_planar)
    int* status;                                           // optional device word: bit 1 raised when an e4m3 output was clamped
    const void* zeros;
    float* gn_partial; int gn_cpg;                         // optional [batch][tiles][Cout/gn_cpg][3]
    int batch, H, W, Cin, Cout;
    // optional fused 1x1 conv (conv_shortcut) on bf16 operands: out += scW . scX; then no residual.  scX NHWC bf16
    // [batch][H][W][scCin]; scW [scCin/32][Cout rows in the fp8 kernel's permuted order][32] bf16, pre-divided by mult[cout]
    const bf16_t* scX; const bf16_t* scW; int scCin;
    int shape;                                             // tile: 0 = 8 x 32 px on 4 waves (two workgroups per CU), 1 = 16 x 32 px, 2 = 8 x 64 px on 8 waves (one per CU)
    int tiles_x, ctiles, per_img, ptiles;                  // filled by the launcher
    unsigned long long m_per_img, m_ctiles, m_tiles_x;
};
bool vt_conv3x3_halo_fp8_supported(int Cin, int Cout);
int vt_conv3x3_halo_fp8_tiles(int H, int W);              // GroupNorm partials per image its epilogue writes (default shape: the upper bound)
int vt_conv3x3_halo_fp8_tiles_shape(int H, int W, int shape);
int vt_halo_fp8_row_of_cout(int cout_local /*0..31*/);
hipError_t vt_launch_conv3x3_halo_fp8(const Conv3x3Fp8Args& a, hipStream_t s);

// conv_out (Conv2d(Cin, 32, 3, pad 1)) on a 32-cout halo tile (conv_out_halo.hip): out[b][c][y][x] = (W . x + bias)[c] * post_scale + post_shift for the
// first `keep` channels, fp32 NCHW -- the moments / mode() * scaling + shift epilogue of DiffusersVAEWrapper.encode (diffusers_vae_loader.py:78-86)
struct ConvOutArgs {
    const bf16_t* X;        // NHWC 16-bit [batch][H][W][Cin] (bf16, or fp16 bits with f16)
    const bf16_t* Wp;       // packed [Cin/32][9 taps (ky * 3 + kx)][32 couts][32] 16-bit
    const float* bias;      // [32] or null
    float* out;             // [batch][keep][H][W] fp32
    const void* zeros;
    int batch, H, W, Cin, Cout, keep;
    float post_scale, post_shift;
    int f16;
};
bool vt_conv_out_halo_supported(int Cin, int Cout);
hipError_t vt_launch_conv_out_halo(const ConvOutArgs& a, hipStream_t s);

// conv_in: fp32 NCHW image -> NHWC 128-channel fp32 (+ optional bf16) rows, direct fp32 conv 3x3 p1.
// gn_partial (optional): (n, mean, M2) triples of the output, [B][parts][Cout/gn_cpg][3]; *gn_parts receives `parts`.
hipError_t vt_launch_conv_in(const float* x_nchw, const float* w_packed /*[27][Cout]*/, const float* bias,
                             float* out_f32, bf16_t* out_bf16, f16_t* out_f16, float* gn_partial, int gn_cpg, int* gn_parts,
                             int B, int H, int W, int Cout, hipStream_t s);
int vt_conv_in_parts(int H, int W);
// MFMA variant (Cout == 128, 32 GroupNorm groups): wpk = [128 rows][32 k] bf16, rows in the interleaved cout order
hipError_t vt_launch_conv_in_mfma(const float* x_nchw, const bf16_t* wpk, const float* bias, float* out_f32, bf16_t* out_bf16,
                                  f16_t* out_f16, float* gn_partial, int* gn_parts, int B, int H, int W, hipStream_t s);
int vt_conv_in_mfma_parts(int H, int W);

// x_dtype below: 0 = bf16, 1 = fp32, 2 = fp16.
// GroupNorm statistics: x rows [B][HW][C] -> partial (count, mean, M2) per
// (b, chunk, group), then finalize -> per (b, c) scale/shift so that y = x*scale + shift.
hipError_t vt_launch_gn_stats(const void* x, int x_dtype, int B, int HW, int C, int groups,
                              float* partial, int* nchunks_out, hipStream_t s);
hipError_t vt_launch_gn_finalize(const float* partial /*[B][nparts][groups][3]*/, int nparts, int B, int C, int groups,
                                 float eps, const float* gamma, const float* beta, float* scale_shift /*[B][C][2]*/,
                                 hipStream_t s, int* status = nullptr /* device word: bit 0 raised on non-finite statistics */);
int vt_gn_max_chunks(int HW, int C);
// y = act(x*scale + shift) -> bf16 rows, or (out_fp8_scale > 0) e4m3 rows of out_fp8_scale * y, saturated at +-448
hipError_t vt_launch_gn_apply(const void* x, int x_dtype, const float* scale_shift, void* y, int B, int HW,
                              int C, int silu, hipStream_t s, float out_fp8_scale = 0.f,
                              int* status = nullptr /* e4m3 output: bit 1 of this device word is raised if a value was clamped */,
                              int out_f16 = 0 /* 1: y is written as fp16 rows instead of bf16 (the convs' fp16-operand mode) */);

hipError_t vt_launch_preprocess_u8(const unsigned char* in_hwc, float* out_nchw, int B, int H, int W, hipStream_t s);
// Pillow's two-pass 8-bit resample; tab_*: [n_out][2 + ksize] int32 (first, count, 22-bit coefficients) on the device;
// tmp: crop_h * dst_w * 3 bytes (or crop_h * crop_w * 3 when only the height changes)
hipError_t vt_launch_resize_u8(const unsigned char* src, int src_h, int src_w, int left, int top, int crop_w, int crop_h,
                               unsigned char* dst, int dst_h, int dst_w, const int* tab_h, int ksize_h, const int* tab_v,
                               int ksize_v, unsigned char* tmp, hipStream_t s);

// attention without a softmax pass (misc_kernels.hip): row norms of q|k, the per-row exponent shift + "bound too loose" flag
// per launch group, and the reduction of the Q.K^T epilogue's (row, column slot) partials (op 0 max, op 1 reciprocal of the sum)
hipError_t vt_launch_attn_row_norms(const bf16_t* qk, long long rows, int C, float* qn, float* kn, float* sd, hipStream_t s);
hipError_t vt_launch_attn_shift(const float* qn, const float* kn, const float* sd, int images, int S, float alpha, float max_gap,
                                float* shift, int* flags, int group, hipStream_t s);
hipError_t vt_launch_attn_row_reduce(const float* part, int slots, long long row_bs, int S, int batch, int op, float* out,
                                     const int* gate, int gate_expect, hipStream_t s);
// row softmax: scores fp32 or fp16 [rows][lds] -> probs bf16 [rows][ldp]; columns [n, ldp) are written as zero.
hipError_t vt_launch_softmax_rows(const void* scores, int scores_f16, bf16_t* probs, long long rows, int n, int lds,
                                  int ldp, hipStream_t s);

// decoder (all fp32)
struct DecoderWeights;   // defined in capi.hip
