"""Drop-in counterpart of the reference's diffusers_vae_loader.py (same function and class names,
same arguments, same printed messages, same strict=False loading contract), backed by the HIP
encoder instead of `diffusers.models.AutoencoderKL`.

Reference: diffusers_vae_loader.py:7-53 (load_diffusers_vae_from_config), :67-94
(DiffusersVAEWrapper), :96-100 (create_vae_from_config_file), :102-134 (get_diffusers_vae_config).
"""
import json
import os

import torch

from .autoencoder_kl import AutoencoderKL

_DEFAULT_BLOCKS = ["DownEncoderBlock2D"] * 4
_DEFAULT_UP = ["UpDecoderBlock2D"] * 4


def load_diffusers_vae_from_config(config_dict, model_path=None):
    g = config_dict.get
    vae = AutoencoderKL(
        in_channels=g("in_channels", 3), out_channels=g("out_channels", 3),
        down_block_types=g("down_block_types", _DEFAULT_BLOCKS), up_block_types=g("up_block_types", _DEFAULT_UP),
        block_out_channels=g("block_out_channels", [128, 256, 512, 512]), layers_per_block=g("layers_per_block", 2),
        act_fn=g("act_fn", "silu"), latent_channels=g("latent_channels", 16), norm_num_groups=g("norm_num_groups", 32),
        sample_size=g("sample_size", 1024), scaling_factor=g("scaling_factor", 0.3611),
        shift_factor=g("shift_factor", 0.1159), use_quant_conv=g("use_quant_conv", False),
        use_post_quant_conv=g("use_post_quant_conv", False), force_upcast=g("force_upcast", True),
        mid_block_add_attention=g("mid_block_add_attention", True))

    if model_path and os.path.exists(model_path):
        print(f"加载预训练权重: {model_path}")
        if model_path.endswith(".safetensors"):
            from safetensors.torch import load_file as load_safetensors
            state_dict = load_safetensors(model_path)
        else:
            # the reference unpickles here (torch.load default); this build refuses to execute
            # checkpoint code and accepts tensor-only files
            state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
        missing_keys, unexpected_keys = vae.load_state_dict(state_dict, strict=False)
        # the HIP object holds encoder tensors only: decoder.* / post_quant_conv.* are expected extras
        if missing_keys:
            print(f"缺失的键: {missing_keys}")
        if unexpected_keys:
            print(f"意外的键: {unexpected_keys}")
        print("成功加载预训练VAE权重")
    return vae


def load_diffusers_vae_from_pretrained(model_name_or_path, subfolder=None):
    """Local-directory form only (config.json + diffusion_pytorch_model.safetensors); hub names need
    network access, which this build never uses.  Returns None on failure like the reference (:55-65)."""
    try:
        root = os.path.join(model_name_or_path, subfolder) if subfolder else model_name_or_path
        with open(os.path.join(root, "config.json"), "r", encoding="utf-8") as f:
            cfg = json.load(f)
        weights = None
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            if os.path.exists(os.path.join(root, name)):
                weights = os.path.join(root, name)
                break
        vae = load_diffusers_vae_from_config(cfg, weights)
        print(f"成功从 {model_name_or_path} 加载预训练VAE")
        return vae
    except Exception as e:  # noqa: BLE001 - mirrors the reference's catch-all
        print(f"从 {model_name_or_path} 加载VAE失败: {e}")
        return None


class DiffusersVAEWrapper(torch.nn.Module):
    # The reference's `vae_model.encode(x)` either returns finite latents or raises.  The HIP encoder never synchronises the host, so an
    # overflow of its fp16 residual-stream storage (or an e4m3 clamp in fp8 mode) only raises a sticky device word; this wrapper -- the
    # object the reference's callers hold -- reads it after every encode (one 4-byte copy + a stream synchronise) and raises.  Callers
    # that poll the word themselves where they synchronise anyway (the CLIs, EncodeTagPipeline, bench.py) switch it off.
    check_finite = True

    def __init__(self, vae_model):
        super().__init__()
        self.vae = vae_model

    def forward(self, x):
        raise NotImplementedError("reconstruction needs the VAE decoder, which is outside the inference hot path")

    def encode(self, x):
        if isinstance(self.vae, AutoencoderKL):
            # fused: conv_out epilogue writes mode()*scaling_factor + shift_factor directly
            latent = self.vae.encode_mode_scaled(x)
            if self.check_finite and not self.vae.check_finite:
                self.vae.raise_on_status()
            return latent
        posterior = self.vae.encode(x).latent_dist
        latent = posterior.mode()
        if getattr(self.vae.config, "scaling_factor", None) is not None:
            latent = latent * self.vae.config.scaling_factor
        if getattr(self.vae.config, "shift_factor", None) is not None:
            latent = latent + self.vae.config.shift_factor
        return latent

    def decode(self, z):
        raise NotImplementedError("the VAE decoder is outside the inference hot path")


def create_vae_from_config_file(config_path, model_path=None):
    with open(config_path, "r", encoding="utf-8") as f:
        config = json.load(f)
    return DiffusersVAEWrapper(load_diffusers_vae_from_config(config, model_path))


def get_diffusers_vae_config():
    return {
        "_class_name": "AutoencoderKL", "_diffusers_version": "0.30.0.dev0", "act_fn": "silu",
        "block_out_channels": [128, 256, 512, 512], "down_block_types": list(_DEFAULT_BLOCKS),
        "force_upcast": True, "in_channels": 3, "latent_channels": 16, "latents_mean": None, "latents_std": None,
        "layers_per_block": 2, "mid_block_add_attention": True, "norm_num_groups": 32, "out_channels": 3,
        "sample_size": 1024, "scaling_factor": 0.3611, "shift_factor": 0.1159, "up_block_types": list(_DEFAULT_UP),
        "use_post_quant_conv": False, "use_quant_conv": False,
    }
