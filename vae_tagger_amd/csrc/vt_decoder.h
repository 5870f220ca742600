// Decoder weight table (device fp32 pointers) + launch entry points, shared by decoder.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct DecSelfAttnW {
    const float *ln_w, *ln_b, *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;
};

struct DecoderWeights {
    int num_classes = 0;
    int latent_channels = 16;
    int heads = 8;
    int plain = 0;            // ClassificationDecoder (--no_attention)
    int use_spatial = 0, use_self = 0, use_cross = 0;
    int ca_hidden = 2;
    const float *ca_w0 = nullptr, *ca_w2 = nullptr, *sa_w = nullptr;
    const float *fc_w = nullptr, *fc_b = nullptr, *bn_scale = nullptr, *bn_shift = nullptr;
    DecSelfAttnW sa{};
    const float *qg_w = nullptr, *qg_b = nullptr;
    const float *cx_q_w = nullptr, *cx_q_b = nullptr, *cx_k_w = nullptr, *cx_k_b = nullptr;
    const float *cx_v_w = nullptr, *cx_v_b = nullptr, *cx_o_w = nullptr, *cx_o_b = nullptr;
    const float* cls_w[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* cls_b[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* cls_ln_w[3] = {nullptr, nullptr, nullptr};
    const float* cls_ln_b[3] = {nullptr, nullptr, nullptr};
};

hipError_t vt_decoder_forward(const DecoderWeights& w, const float* latent_nchw, int B, int H, int W, float* ws,
                              float* logits, hipStream_t s);
size_t vt_decoder_workspace_floats(int B, int C, int H, int W);
hipError_t vt_decoder_sort(const float* logits, int B, int N, float* conf, long long* idx, hipStream_t s);
hipError_t vt_decoder_summary(const float* conf, const long long* idx, int B, int N, float threshold, int K, float* top_conf,
                              int* top_idx, float* stats, hipStream_t s);
