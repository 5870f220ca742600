/*
 * vae_tagger_hip.h -- C ABI of libvae_tagger_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the inference hot path of spawner1145/vae-tagger.  The reference has no
 * FFI of its own: the boundary is a Python object protocol.  Each entry
 * point below names the reference call it stands in for; INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned DEVICE memory unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); no call synchronises the host;
 *   - every call returns VT_OK or an error code; vt_last_error(ctx) gives the message; nothing aborts;
 *   - one context per device / thread; no global state: every option of vt_set_flag lives in the context, every call runs on
 *     the context's device and restores the caller's current device before it returns;
 *   - workspace is caller-provided (query the *_workspace_bytes function first), 256-B aligned.
 */
#ifndef VAE_TAGGER_HIP_H
#define VAE_TAGGER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vt_context vt_context;

enum { VT_OK = 0, VT_ERR_INVALID = 1, VT_ERR_HIP = 2, VT_ERR_STATE = 3, VT_ERR_MISSING_WEIGHT = 4,
       VT_ERR_WORKSPACE = 5 };
enum { VT_F32 = 0, VT_BF16 = 1, VT_F16 = 2 };

const char* vt_version(void);
int vt_create(int device, vt_context** out);
void vt_destroy(vt_context* ctx);
const char* vt_last_error(const vt_context* ctx);

/* ---- model construction --------------------------------------------------------------------
 * vt_encoder_configure   <- AutoencoderKL(...) hyper-parameters, diffusers_vae_loader.py:8-35
 * vt_set_weight          <- vae.load_state_dict(state_dict, strict=False), diffusers_vae_loader.py:44
 *                           and decoder.load_state_dict(...), infer_full.py:63.  `name` is the
 *                           state-dict key (diffusers / reference naming); `data` is a HOST pointer.
 * vt_encoder_finalize    packs weights for the MFMA kernels ([Cout][tap][Cin] bf16) and uploads them.
 * vt_decoder_configure   <- create_attention_decoder / ClassificationDecoder, infer_full.py:42-58
 */
int vt_encoder_configure(vt_context* ctx, int in_channels, int latent_channels, const int* block_out_channels,
                         int n_blocks, int layers_per_block, int norm_num_groups, float scaling_factor,
                         int has_scaling_factor, float shift_factor, int has_shift_factor);
int vt_set_weight(vt_context* ctx, const char* name, const void* host_data, int dtype, const int64_t* shape,
                  int ndim);
int vt_encoder_finalize(vt_context* ctx);
int vt_decoder_configure(vt_context* ctx, int num_classes, int latent_channels, int plain_decoder,
                         int use_spatial_attention, int use_self_attention, int use_cross_attention,
                         int attention_heads);
int vt_decoder_finalize(vt_context* ctx);

/* ---- hot path ------------------------------------------------------------------------------
 * vt_encode          <- DiffusersVAEWrapper.encode(x), diffusers_vae_loader.py:78-86:
 *                       vae.encode(x).latent_dist.mode() * scaling_factor + shift_factor.
 *                       x: fp32 NCHW [B,3,H,W] in [-1,1].  mode selects what is written (fp32 NCHW):
 *                         0: moments [B,2*latent,H/8,W/8] (mean | logvar) -- AutoencoderKL.encode surface
 *                         1: latent_dist.mode() = mean [B,latent,H/8,W/8]
 *                         2: mode() * scaling_factor + shift_factor -- DiffusersVAEWrapper.encode
 *                       (H/8, W/8 for the four-block FLUX configuration; in general H >> (num_blocks - 1), W >> (num_blocks - 1):
 *                       one Downsample2D per block but the last.  The library cannot see the size of latent_out: the caller's
 *                       buffer must hold B x channels x that many fp32 values.)
 * vt_decode_logits   <- decoder.forward(latent), modules.py:424-468 / :333-349 -> fp32 [B,N]
 * vt_get_confidence  <- sigmoid + descending sort, modules.py:470-475 (ties: ascending tag index; NaN logits sort last;
 *                       any N: up to 16384 tags in one LDS pass, more through global-memory merge passes)
 * vt_summarize_confidence <- the per-image summary loop of infer_full.py:106-125 on the sorted outputs: per image the
 *                       first K (confidence fp32, tag index int32; -1 / 0 beyond N) pairs and stats[4] = {number of tags with
 *                       confidence >= threshold, max confidence, (sum of the first five) / 5, number of non-finite confidences}
 * vt_status          sticky device-side health word of the context (SYNCHRONISES `stream`): bit 0 (VT_STATUS_NONFINITE) =
 *                       some GroupNorm saw non-finite statistics since the last clear -- an activation left the fp16 range
 *                       of the residual-stream storage (rerun with vt_set_flag(ctx, 4, 0)) or the weights hold inf / NaN;
 *                       bit 1 (VT_STATUS_FP8_SATURATED, fp8 mode only) = some activation exceeded the e4m3 range (+-448 after its
 *                       scale) and was clamped -- the checkpoint's activations are too large for flag 11; rerun without it
 * vt_encode_tag      <- the loop body of infer_full.py:101-105 for a whole batch
 */
size_t vt_encode_workspace_bytes(const vt_context* ctx, int B, int H, int W);
int vt_encode(vt_context* ctx, const float* x_nchw, int B, int H, int W, int mode, float* latent_out,
              void* workspace, size_t workspace_bytes, void* stream);
size_t vt_decode_workspace_bytes(const vt_context* ctx, int B, int h, int w);
int vt_decode_logits(vt_context* ctx, const float* latent_nchw, int B, int h, int w, float* logits_out,
                     void* workspace, size_t workspace_bytes, void* stream);
int vt_get_confidence(vt_context* ctx, const float* logits, int B, int N, float* conf_sorted_out,
                      int64_t* indices_out, void* stream);
int vt_summarize_confidence(vt_context* ctx, const float* conf_sorted, const int64_t* indices, int B, int N, float threshold,
                            int K, float* top_conf_out /* [B][K] */, int32_t* top_idx_out /* [B][K] */,
                            float* stats_out /* [B][4] */, void* stream);
enum { VT_STATUS_NONFINITE = 1, VT_STATUS_FP8_SATURATED = 2 };
int vt_status(vt_context* ctx, int clear, int* status_out /* host */, void* stream);
/* the same word WITHOUT a host synchronisation: copied (and optionally cleared) in stream order into `status_out`, which is pinned host
 * memory or device memory and is valid once work recorded on `stream` behind this call has completed (an event / a later sync).  The
 * pipelined CLIs read batch n's word this way while batch n + 1 runs (infer_full.py:95-128 is the serial loop they replace). */
int vt_status_async(vt_context* ctx, int clear, int* status_out /* pinned host or device */, void* stream);
size_t vt_encode_tag_workspace_bytes(const vt_context* ctx, int B, int H, int W);
int vt_encode_tag(vt_context* ctx, const float* x_nchw, int B, int H, int W, float* latent_out /* may be NULL */,
                  float* logits_out, void* workspace, size_t workspace_bytes, void* stream);

/* vt_preprocess_u8 <- transforms.ToTensor() + Normalize([0.5]*3, [0.5]*3), modules.py:136-140, on the device:
 * uint8 HWC RGB [B,H,W,3] -> fp32 NCHW [B,3,H,W] in [-1,1] (the resize stays with PIL on the host). */
int vt_preprocess_u8(vt_context* ctx, const uint8_t* in_hwc, int B, int H, int W, float* out_nchw, void* stream);

/* vt_resize_u8 <- transforms.Resize((r, r)) (filter 0, bilinear) and SmartResize's crop + Image.resize(LANCZOS) (filter 1),
 * modules.py:126-178: Pillow's two-pass 8-bit resample (libImaging/Resample.c) reproduced bit for bit on the device.
 * src: uint8 HWC RGB [src_h][src_w][3]; the crop box (left, top, crop_w, crop_h) inside it is resized to dst uint8 HWC
 * [dst_h][dst_w][3].  The coefficient tables are built on the host exactly as Pillow builds them and copied with the
 * stream; the call does not wait for the GPU (it may wait for the PREVIOUS call's table copy). */
size_t vt_resize_workspace_bytes(int crop_h, int crop_w, int dst_h, int dst_w, int filter);
/* Host-only: one axis' table as vt_resize_u8 builds it, table_out[out_size][2 + ksize] = (first sample, count,
 * 22-bit coefficients...); returns ksize (call with table_out = NULL to size the buffer), -1 on a bad argument. */
int vt_resize_table(int in_size, int out_size, int filter, int* table_out, int table_ints);
int vt_resize_u8(vt_context* ctx, const uint8_t* src_hwc, int src_h, int src_w, int crop_left, int crop_top, int crop_w, int crop_h,
                 uint8_t* dst_hwc, int dst_h, int dst_w, int filter, void* workspace, size_t workspace_bytes, void* stream);

/* algorithmic FLOPs of one encoder forward at HxW -- for roofline reporting */
double vt_encoder_flops(const vt_context* ctx, int H, int W);

/* ---- options ---------------------------------------------------------------------------------
 * flag 0: 1 (default) = 3x3 stride-1 convs use the halo-tile kernel (conv3x3_halo.hip),
 *         0 = every contraction uses the generic implicit-GEMM kernel (conv_gemm.hip).
 * flag 1: 1 (default) = conv epilogues emit GroupNorm partial statistics for the next norm,
 *         0 = every GroupNorm runs its own statistics pass.
 * flag 2: 1 = GroupNorm-apply + SiLU in front of a 3x3 stride-1 conv runs inside that conv's halo staging,
 *         0 (default) = as a standalone HBM-bound pass (one read + one bf16 write of the tensor).
 * flag 3: two-workgroups-per-CU tiles of the halo conv: 3 (default) = every plain-input layer on the
 *         4-wave x 256-VGPR tile (16x16 px x 128 couts), 2 = only the 128-cout layers on it, 1 = the 128-cout layers on
 *         the 8-wave x 128-VGPR tile, 0 = one workgroup per CU (16x16 px x 256 couts / 32x16 px x 128 couts);
 *         4 = the round-4 experiment tile: ONE wave per SIMD (4 waves x 512 registers, accumulators in AGPRs, 32x16 px x 128 couts) --
 *         bit-identical outputs, 16-21 % slower on bare layer loops; kept for the microbenchmark, never the default.
 * flag 4: 1 (default) = the residual stream between resnet blocks is STORED as fp16 (all arithmetic stays fp32;
 *         halves the HBM traffic of the conv2 epilogues and of norm1) and each block's conv1 output as fp16 instead of
 *         bf16 (read only by norm2), 0 = stored as fp32 / bf16.
 * flag 5: 1 (default) = conv_in (3 -> 128 channels) runs on the matrix cores with split (hi + lo) bf16 operands
 *         (products to ~2^-16 relative), 0 = exact fp32 VALU conv.  vt_op_conv_in follows it when Cout == 128.
 * flag 6: 1 (default) = 1x1 / GEMM launches with K <= 512 (resnet shortcuts, attention
 *         projections, Q.K^T) and the 128-cout stride-2 conv use a 192x128 tile at two workgroups per CU,
 *         0 = the 256x256 / 256x128 tiles.
 * flag 7: mid-block attention softmax. 0 (default) = no softmax pass: Q.K^T stores exp(s - c_i) (c_i from operand norms),
 *         P.V divides by the row sums; a launch group whose norm bound is too loose is flagged on the device and takes c_i
 *         = the exact row maximum from an extra, otherwise gated-off Q.K^T pass.  1 = always the exact row maximum.
 *         2 = fp16 scores, a row-softmax pass, bf16 P.
 * flag 8: 1 (default) = a resnet block's 1x1 conv_shortcut runs inside its conv2 launch (extra K-steps on a bf16 copy of the
 *         block input): no shortcut tensor is written or read back.  0 = separate GEMM launch + residual add.
 * flag 9: 1 (default) = Q.K^T of the mid-block attention (modes 0 / 1 of flag 7, 512 channels) on its own kernel (Q rows in
 *         registers, keys streamed through LDS, row sums in registers); 0 = the generic GEMM with the exp epilogue.
 * flag 10: 1 (default) = P.V reads the probabilities (4+ GB per launch, read once) with the streaming (nt) cache policy so they
 *         do not displace the rest of the working set from L2 / Infinity Cache; 0 = default policy.
 * flag 12: 1 (default) = with flag 9, Q.K^T stores the probabilities in the MFMA fragment order it holds them in and P.V runs on its
 *         own kernel that loads them straight into registers (only v^T passes through LDS); 0 = row-major P + the generic GEMM.
 * flag 11: 1 = BASELINE.json configs[4]: all 23 3x3 convolutions of the resnet / downsample stack run on fp8 (OCP e4m3) operands on
 *         the fp8 MFMA (2x the bf16 rate): the 20 stride-1 convs on v_mfma_scale_f32_32x32x64_f8f6f4 (conv3x3_halo_fp8.hip), the
 *         three stride-2 convs on the same instruction over the input's phase planes (conv3x3_s2_halo_fp8.hip; flag 13).  Weights e4m3
 *         with per-output-channel scales; activations e4m3(8 x) written by the GroupNorm-apply pass, the block output feeding a
 *         stride-2 conv e4m3(x); fp32 accumulate.  The mid-block attention follows (flags 14, 15); conv_in, conv_out, the 1x1
 *         shortcuts and to_out stay bf16 / fp32.  OPT-IN, for tagging only: latents move by ~1e-1 (max; rms 2e-2), logits stay
 *         within 1e-2 of the CPU reference (measured 4-5e-3; tests/diagnostics/fp8_study.py).  0 (default) = bf16.
 * flag 13: 1 (default) = the three stride-2 convs (Downsample2D) run on the phase-plane halo kernels (conv3x3_s2_halo.hip, and
 *         conv3x3_s2_halo_fp8.hip in fp8 mode); 0 = on the generic implicit GEMM (bf16, or its e4m3 variant).
 * flag 14: 1 (default) = in fp8 mode Q.K^T and P.V multiply e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4 (attn_fp8.hip: q8 | k8 =
 *         e4m3(8 q | 8 k), v8^T, numerators e4m3(32 exp(s - sampled row maximum)) with a device-side overflow flag and a gated exact
 *         redo; flag 7 = 1: always the exact maximum); 0 = the bf16 attention kernels in fp8 mode too.
 * flag 15: 1 (default) = with flag 14, the q | k and v projections multiply e4m3 operands too (tokens e4m3(8 x) from the GroupNorm pass,
 *         [Wq; Wk] and Wv as e4m3(W / s), one scale per matrix) and write q8 | k8 and v8^T directly (proj_fp8_kernel); 0 = bf16
 *         projections followed by conversion passes.
 * flag 16: tile shape of the fp8 halo conv (flag 11) = value & 3: 0 (default) = 8 rows x 32 px x 128 couts on 4 waves, two workgroups
 *         per CU; 1 = 16 x 32 px, 2 = 8 x 64 px, both on 8 waves, one workgroup per CU (a staged weight tile serves twice the pixels);
 *         applied to the layers with Cin <= 128, or to every layer with value & 4.  The conv outputs are bit-identical for every value (the
 *         GroupNorm partials are per tile, so their merge order -- the last bits of the statistics -- follows the shape).
 * flag 17: 1 (default) = the attention's bf16 linear layers run on attn_qk.hip's skeleton: q | k and v^T (mode 4: one operand's rows in
 *         registers, the other's streamed through LDS, bias + bf16 store in the epilogue) and to_out (mode 5: + residual stream, fp16 / fp32
 *         stores, GroupNorm partials of the result); 0 = the generic GEMM (conv_gemm.hip).
 * flag 18: 1 = fp16 instead of bf16 MFMA operands for the convolutions (GroupNorm outputs, the 16-bit operand copies and the packed
 *         weights carry fp16 bits: the same 2 B per element, 11 significand bits instead of 8; v_mfma_f32_16x16x32_f16).  Latents move
 *         ~6x closer to the fp32 reference (max |dlatent| 1.5e-3 instead of 1e-2 .. 2e-2 on smooth pictures, where bf16's rounding
 *         errors add coherently); the matrix pipe draws more power on fp16 data, about 4 % of the images/s.  Values must fit fp16
 *         (|x| <= 65504: GroupNorm + SiLU outputs and weights do; an overflow shows as status bit 0).  Ignored in fp8 mode (flag 11);
 *         needs the default kernel selection (flags 0, 2, 3, 13 at their defaults), other settings keep bf16 for the convs they affect.
 *         The attention keeps bf16 (its softmax numerators need bf16's range).  0 (default) = bf16 operands, BASELINE.json's dtype.
 * flag 19: 1 (default) = the 16-bit (fp8 mode: e4m3) copy of a stage's output that feeds its stride-2 conv is written chunk-planar --
 *         [C/32][H][W][32] (e4m3: [C/64][H][W][64]) per image instead of NHWC -- when a phase-plane kernel (flag 13) reads it: a 128-B line
 *         then holds one channel chunk of two neighbouring pixels, the halves of a row's two planes staged three K-steps apart, instead of
 *         two chunks of one pixel staged nine K-steps apart, by when the line has left the L2 (every line was fetched twice);
 *         0 = NHWC.  Same arithmetic, same bits.
 * flag 20: 1 (default) = conv_out (512 -> 32 channels, the moments / mode() epilogue) on its own 32-cout halo tile (conv_out_halo.hip: the
 *         18 x 18 halo of a 16 x 16-pixel tile and the chunk's nine weight tiles staged once per 32-channel chunk); 0 = the generic GEMM.
 */
int vt_set_flag(vt_context* ctx, int flag, int value);

/* ---- measurement ----------------------------------------------------------------------------
 * Between vt_profile_begin and vt_profile_end every launch of the implicit-GEMM MFMA kernel is
 * bracketed by hipEvents recorded on the launch stream.  vt_profile_end synchronises on them and
 * returns, per kernel/tile configuration (vt_profile_num_configs() of them), the launch count, summed duration (ms) and summed
 * ALGORITHMIC FLOPs (2*B*Hout*Wout*Cout*taps*Cin).  The LAST slot is the HBM-bound GroupNorm(+SiLU) apply
 * pass: its 'flops' entry carries algorithmic BYTES (one read + one bf16 write).  bench.py derives
 * roofline.achieved from these.
 */
/* diagnostics: between vt_debug_trace(ctx, 1, ...) and vt_debug_trace(ctx, 0, sums, max, &n) every GroupNorm of the encoder records two
 * order-independent checksums on the launch stream -- of the (n, mean, M2) partials it consumed and of its (scale, shift) table -- in launch order;
 * the second call synchronises the device and copies them out.  Two runs of the same input give the same list unless some producer's statistics
 * are not deterministic; the first differing index names the layer (tests/diagnostics/gn_trace_diff.py). */
int vt_debug_trace(vt_context* ctx, int enable, unsigned long long* sums_out, int max_sums, int* n_out);
int vt_profile_num_configs(void);
int vt_profile_begin(vt_context* ctx);
int vt_profile_end(vt_context* ctx, int max_cfg, long long* launches, double* total_ms, double* total_flops,
                   const char** kernel_names);

/* ---- single operators (parity tests drive each kernel through the same ABI) ------------------
 * NHWC bf16 activations, weights in the reference's own layouts (fp32 host order is converted by
 * the caller to bf16 [Cout][kh][kw][Cin]); fp32 accumulate.
 */
int vt_op_conv2d(vt_context* ctx, const void* x_bf16_nhwc, const void* w_bf16_ohwi, const float* bias,
                 const float* residual_f32, float* out_f32, void* out_bf16, int B, int Hin, int Win, int Cin,
                 int Cout, int ksize, int stride, int pad_lo, int pad_hi, void* stream);
/* 3x3 stride-1 pad-1 conv of silu(x*scale + shift): x is fp32 or bf16 NHWC, scale_shift [B][Cin][2]; the
 * normalise + SiLU runs inside the conv's LDS staging (no separate pass).  Cin % 32 == 0, Cout % 128 == 0. */
int vt_op_norm_silu_conv3x3(vt_context* ctx, const void* x_nhwc, int x_dtype, const float* scale_shift,
                            const void* w_bf16_ohwi, const float* bias, const float* residual_f32, float* out_f32,
                            void* out_bf16, int B, int H, int W, int Cin, int Cout, void* stream);
/* conv2d whose epilogue also produces the GroupNorm statistics of its output: returns per (image, channel)
 * (scale, shift) with GroupNorm(out) = out*scale + shift.  The encoder uses this fusion between layers. */
size_t vt_op_conv2d_gn_workspace_bytes(int B, int Hout, int Wout, int Cout);
int vt_op_conv2d_gn(vt_context* ctx, const void* x_bf16_nhwc, const void* w_bf16_ohwi, const float* bias,
                    const float* residual_f32, float* out_f32, void* out_bf16, int B, int Hin, int Win, int Cin,
                    int Cout, int ksize, int stride, int pad_lo, int pad_hi, int groups, float eps, const float* gamma,
                    const float* beta, float* scale_shift_out, void* workspace, void* stream);
/* the fp8 conv of flag 11 as a single operator: x fp32 NHWC and w fp32 OIHW (both on the device) are quantised exactly as the
 * encoder quantises them (x -> e4m3(8 x), w -> e4m3 with per-cout absmax scales); out fp32 NHWC.  Cin % 64 == 0, Cout % 128 == 0. */
size_t vt_op_conv3x3_fp8_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vt_op_conv3x3_fp8(vt_context* ctx, const float* x_f32_nhwc, const float* w_f32_oihw, const float* bias, const float* residual_f32,
                      float* out_f32, int B, int H, int W, int Cin, int Cout, int stride /* 1: pad 1; 2: pad (0,1,0,1), x -> e4m3(x) */,
                      void* workspace, void* stream);
int vt_op_gemm_nt(vt_context* ctx, const void* a_bf16, const void* b_bf16, const float* bias, float* out_f32,
                  void* out_bf16, int batch, int M, int N, int K, int lda, int ldb, int ldo, long long a_bs,
                  long long b_bs, long long o_bs, float alpha, int bias_per_row, void* stream);
int vt_op_conv_in(vt_context* ctx, const float* x_nchw, const float* w_oihw, const float* bias, float* out_f32,
                  void* out_bf16, int B, int H, int W, int Cout, void* workspace, void* stream);
size_t vt_op_groupnorm_workspace_bytes(int B, int HW, int C);
int vt_op_groupnorm(vt_context* ctx, const void* x, int x_dtype, int B, int HW, int C, int groups, float eps,
                    const float* gamma, const float* beta, int silu, void* y_bf16, void* workspace, void* stream);
int vt_op_softmax_rows(vt_context* ctx, const float* scores, void* probs_bf16, int rows, int n, int lds, int ldp,
                       void* stream);
size_t vt_op_attention_workspace_bytes(int B, int S, int C);
int vt_op_attention(vt_context* ctx, const void* x_bf16 /* [B][S][C] normed tokens */, const float* residual_f32,
                    float* out_f32, int B, int S, int C, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAE_TAGGER_HIP_H */
