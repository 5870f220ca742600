"""Seeded synthetic weights, one generator per state-dict key.

There is no FLUX VAE checkpoint and no trained decoder in this environment, so
benchmarks, smoke runs and parity fixtures use random-initialised weights of the
reference architecture, drawn from PyTorch's own default initialisation distributions
(conv/linear: U(+-1/sqrt(fan_in)) for weight and bias; norms: 1 / 0, lightly perturbed).  Every tensor is drawn from its own generator seeded by
crc32(key) ^ seed, so a fixture never depends on module construction order and
no multi-MB weight file has to be committed.

Key manifests:
  * encoder: the diffusers-format names the reference loads with strict=False
    (reference diffusers_vae_loader.py:39-44; architecture literal :102-134).
  * decoder: the names observed on the reference module's state_dict()
    (reference modules.py:358-422, :303-356).
"""
import zlib

import torch

FLUX_BLOCK_OUT = (128, 256, 512, 512)
FLUX_LATENT_CHANNELS = 16
FLUX_IN_CHANNELS = 3
FLUX_LAYERS_PER_BLOCK = 2


def encoder_manifest(block_out=FLUX_BLOCK_OUT, in_channels=FLUX_IN_CHANNELS,
                     latent_channels=FLUX_LATENT_CHANNELS,
                     layers_per_block=FLUX_LAYERS_PER_BLOCK):
    """Ordered {key: shape} of every encoder tensor in diffusers AutoencoderKL naming."""
    m = {}

    def conv(name, co, ci, k):
        m[name + ".weight"] = (co, ci, k, k)
        m[name + ".bias"] = (co,)

    def norm(name, c):
        m[name + ".weight"] = (c,)
        m[name + ".bias"] = (c,)

    def lin(name, co, ci):
        m[name + ".weight"] = (co, ci)
        m[name + ".bias"] = (co,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    conv("encoder.conv_in", block_out[0], in_channels, 3)
    ci = block_out[0]
    for i, co in enumerate(block_out):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i != len(block_out) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    c = block_out[-1]
    resnet("encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", c)
    lin(a + ".to_q", c, c)
    lin(a + ".to_k", c, c)
    lin(a + ".to_v", c, c)
    lin(a + ".to_out.0", c, c)
    resnet("encoder.mid_block.resnets.1", c, c)
    norm("encoder.conv_norm_out", c)
    conv("encoder.conv_out", 2 * latent_channels, c, 3)
    return m


def attention_decoder_manifest(num_classes, latent_channels=16, use_spatial_attention=True,
                               use_self_attention=True, use_cross_attention=False):
    """{key: shape} of reference AttentionClassificationDecoder (modules.py:358-422)."""
    c = latent_channels
    h = c // 2
    m = {}
    if use_spatial_attention:
        m["spatial_attention.channel_att.0.weight"] = (c // 8, c, 1, 1)
        m["spatial_attention.channel_att.2.weight"] = (c, c // 8, 1, 1)
        m["spatial_attention.spatial_att.0.weight"] = (1, 2, 7, 7)
    m["feature_compress.0.weight"] = (h, c, 3, 3)
    m["feature_compress.0.bias"] = (h,)
    m["feature_compress.1.weight"] = (h,)
    m["feature_compress.1.bias"] = (h,)
    m["feature_compress.1.running_mean"] = (h,)
    m["feature_compress.1.running_var"] = (h,)
    m["feature_compress.1.num_batches_tracked"] = ()
    if use_self_attention:
        p = "self_attention_post."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            m[p + n + ".weight"] = (h, h)
            m[p + n + ".bias"] = (h,)
        m[p + "norm.weight"] = (h,)
        m[p + "norm.bias"] = (h,)
    if use_cross_attention:
        p = "cross_attention."
        m[p + "q_proj.weight"] = (256, 512)
        m[p + "q_proj.bias"] = (256,)
        m[p + "k_proj.weight"] = (256, h)
        m[p + "k_proj.bias"] = (256,)
        m[p + "v_proj.weight"] = (256, h)
        m[p + "v_proj.bias"] = (256,)
        m[p + "out_proj.weight"] = (512, 256)
        m[p + "out_proj.bias"] = (512,)
    dims = [h * 64, 1024, 512, 256]
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        m[f"classifier.{4 * i}.weight"] = (b, a)
        m[f"classifier.{4 * i}.bias"] = (b,)
        m[f"classifier.{4 * i + 1}.weight"] = (b,)
        m[f"classifier.{4 * i + 1}.bias"] = (b,)
    m["classifier.12.weight"] = (num_classes, 256)
    m["classifier.12.bias"] = (num_classes,)
    if use_cross_attention:
        m["query_generator.weight"] = (512, h * 64)
        m["query_generator.bias"] = (512,)
    return m


def plain_decoder_manifest(num_classes, latent_channels=16):
    """{key: shape} of reference ClassificationDecoder (modules.py:303-331), adaptive pooling."""
    m = {}
    dims = [latent_channels * 16, 512, 256]
    m["classifier.0.weight"] = (512, dims[0])
    m["classifier.0.bias"] = (512,)
    m["classifier.1.weight"] = (512,)
    m["classifier.1.bias"] = (512,)
    m["classifier.4.weight"] = (256, 512)
    m["classifier.4.bias"] = (256,)
    m["classifier.5.weight"] = (256,)
    m["classifier.5.bias"] = (256,)
    m["classifier.8.weight"] = (num_classes, 256)
    m["classifier.8.bias"] = (num_classes,)
    return m


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0):
    """Deterministic fp32 tensor for `key`; distribution chosen by the key's role."""
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(100, dtype=torch.int64)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if len(shape) == 1:
        if _is_norm(key):
            # torch initialises norms to (1, 0); perturb so gamma/beta paths are exercised
            if leaf == "weight":
                return 1.0 + 0.1 * torch.randn(shape, generator=g)
            return 0.1 * torch.randn(shape, generator=g)
        # conv / linear bias: torch default U(-1/sqrt(fan_in), 1/sqrt(fan_in)) of the matching weight
        bound = 1.0 / _FAN_IN.get(key, 256) ** 0.5
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
    # conv / linear weight: torch default kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    bound = 1.0 / fan_in ** 0.5
    return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound


_FAN_IN = {}


def _is_norm(key):
    stem = key.rsplit(".", 1)[0]
    last = stem.rsplit(".", 1)[-1]
    if last in ("norm1", "norm2", "group_norm", "conv_norm_out", "norm"):
        return True
    if stem == "feature_compress.1":
        return True
    if stem.startswith("classifier."):
        return int(last) % 4 == 1          # classifier.{1,5,9} are LayerNorms
    return False


def _note_fan_in(manifest):
    for k, s in manifest.items():
        if k.endswith(".weight") and len(s) >= 2:
            f = 1
            for d in s[1:]:
                f *= d
            _FAN_IN[k[:-len("weight")] + "bias"] = f


def synth_state_dict(manifest, seed=0):
    _note_fan_in(manifest)
    return {k: synth_tensor(k, s, seed) for k, s in manifest.items()}


def synth_images(batch, height, width, seed=0):
    """Uniform [-1, 1] fp32 NCHW, the range Normalize(0.5, 0.5) produces (modules.py:139)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.rand(batch, 3, height, width, generator=g) * 2.0 - 1.0
