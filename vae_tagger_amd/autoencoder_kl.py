"""AutoencoderKL-shaped object over the HIP encoder.

Stands in for `diffusers.models.AutoencoderKL` at exactly the surface the reference uses
(diffusers_vae_loader.py:8-35, :44, :73-86; infer_full.py:27-28; train_full.py:213,220):
constructor keywords, `.config.scaling_factor / .shift_factor`, `load_state_dict(sd, strict=False)`,
`.to()`, `.eval()`, `.parameters()`, and `encode(x).latent_dist.{mode,sample,kl}()`.
Only the ENCODER is implemented (the inference hot path); `decode` raises.
"""
from types import SimpleNamespace

import torch

from . import _lib, synth
from ._runtime import HipModule, as_input, stream_ptr, vp, workspace

import ctypes


class DiagonalGaussianDistribution:
    """moments = [mean | logvar] along dim 1; logvar clamped to [-30, 20] as diffusers does."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def kl(self, other=None):
        if other is None:
            return 0.5 * torch.sum(self.mean.pow(2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean - other.mean).pow(2) / other.var + self.var / other.var
                               - 1.0 - self.logvar + other.logvar, dim=[1, 2, 3])


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKL(HipModule):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                 up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, act_fn="silu", latent_channels=16, norm_num_groups=32, sample_size=1024,
                 scaling_factor=0.18215, shift_factor=None, use_quant_conv=True, use_post_quant_conv=True,
                 force_upcast=True, mid_block_add_attention=True, latents_mean=None, latents_std=None, **unused):
        if any(t != "DownEncoderBlock2D" for t in down_block_types):
            raise ValueError("only DownEncoderBlock2D encoder blocks are supported")
        if act_fn != "silu":
            raise ValueError("only act_fn='silu' is supported")
        if use_quant_conv:
            raise ValueError("use_quant_conv=True is not supported (the FLUX VAE config sets it False, "
                             "diffusers_vae_loader.py:31,133)")
        if not mid_block_add_attention:
            raise ValueError("mid_block_add_attention=False is not supported")
        if len(down_block_types) != len(block_out_channels):
            raise ValueError("down_block_types and block_out_channels must have the same length")
        manifest = synth.encoder_manifest(tuple(block_out_channels), in_channels, latent_channels, layers_per_block)
        super().__init__(manifest)
        cfg = dict(in_channels=in_channels, out_channels=out_channels, down_block_types=list(down_block_types),
                   up_block_types=list(up_block_types), block_out_channels=list(block_out_channels),
                   layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                   norm_num_groups=norm_num_groups, sample_size=sample_size, use_quant_conv=use_quant_conv,
                   use_post_quant_conv=use_post_quant_conv, force_upcast=force_upcast,
                   mid_block_add_attention=mid_block_add_attention, latents_mean=latents_mean, latents_std=latents_std)
        # diffusers registers every constructor argument on .config; the reference probes these two
        # with hasattr (diffusers_vae_loader.py:81-84)
        cfg["scaling_factor"] = scaling_factor
        cfg["shift_factor"] = shift_factor
        self.config = SimpleNamespace(**cfg)

    # -- HIP plumbing ------------------------------------------------------------------------------
    def _upload(self, ctx):
        c = self.config
        blocks = (ctypes.c_int * len(c.block_out_channels))(*c.block_out_channels)
        sf, sh = c.scaling_factor, c.shift_factor
        ctx.call("vt_encoder_configure", c.in_channels, c.latent_channels, blocks, len(c.block_out_channels),
                 c.layers_per_block, c.norm_num_groups, float(sf if sf is not None else 1.0), int(sf is not None),
                 float(sh if sh is not None else 0.0), int(sh is not None))
        for k, v in self.state_dict().items():
            ctx.set_weight(k, v)
        ctx.call("vt_encoder_finalize")

    def _run_encode(self, x, mode):
        ctx = self._context()
        x = as_input(x).to(next(self.parameters()).device)
        B, _, H, W = x.shape
        nd = len(self.config.block_out_channels) - 1
        h, w = H >> nd, W >> nd
        ch = self.config.latent_channels * (2 if mode == _lib.ENCODE_MOMENTS else 1)
        out = torch.empty(B, ch, h, w, dtype=torch.float32, device=x.device)
        need = ctx.lib.vt_encode_workspace_bytes(ctx.handle, B, H, W)
        if need == 0:
            raise _lib.VTError(f"vt_encode_workspace_bytes({B},{H},{W}) = 0: unsupported shape")
        ws, ptr = workspace(x.device, need)
        ctx.call("vt_encode", vp(x), B, H, W, mode, vp(out), ctypes.c_void_p(ptr), need, stream_ptr(x.device))
        if self.check_finite:
            self.raise_on_status()
        return out

    def raise_on_status(self):
        """Read (and clear) the sticky health word -- synchronises -- and raise what it says."""
        st = self.status()
        if st & _lib.VT_STATUS_NONFINITE:
            raise FloatingPointError("non-finite activations in the encoder: the fp16 residual-stream storage overflowed "
                                     "(set_fp32_residual() stores it as fp32) or the input / checkpoint holds inf / NaN")
        if st & _lib.VT_STATUS_FP8_SATURATED:
            raise FloatingPointError("fp8 mode: activations exceeded the e4m3 range and were clamped "
                                     "(vt_set_flag(ctx, 11, 0) returns to the bf16 path)")

    # encode() never synchronises the host, so an overflow of the fp16 residual-stream storage cannot raise from it by
    # itself: the library keeps a sticky device-side status word instead (vt_status).  Callers check it where they
    # synchronise anyway (the CLIs and evaluation.py do, after their .cpu()); check_finite = True makes every encode
    # synchronise and raise.
    check_finite = False

    def status(self, clear=True):
        dev = next(self.parameters()).device
        return self._context().status(clear, stream_ptr(dev))

    def set_fp32_residual(self, on=True):
        self._context().call("vt_set_flag", 4, 0 if on else 1)

    def set_fp16_operands(self, on=True):
        """fp16 instead of bf16 MFMA operands for the convolutions (vt_set_flag 18): latents ~6x closer to the fp32 reference -- what a
        smooth picture needs to stay inside 1e-2 -- for about 4 % of the images/s."""
        self._context().call("vt_set_flag", 18, 1 if on else 0)

    # -- diffusers surface ---------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        moments = self._run_encode(x, _lib.ENCODE_MOMENTS)
        out = AutoencoderKLOutput(DiagonalGaussianDistribution(moments))
        return out if return_dict else (out.latent_dist,)

    @torch.no_grad()
    def encode_mode_scaled(self, x):
        """Fused fast path of DiffusersVAEWrapper.encode: mode()*scaling_factor + shift_factor written
        straight from the conv_out epilogue (no moments tensor, no elementwise pass)."""
        return self._run_encode(x, _lib.ENCODE_MODE_SCALED)

    def decode(self, z, *a, **kw):
        raise NotImplementedError("vae_tagger_amd implements the encoder (inference hot path) only")

    @classmethod
    def from_config(cls, config_dict):
        keys = ("in_channels", "out_channels", "down_block_types", "up_block_types", "block_out_channels",
                "layers_per_block", "act_fn", "latent_channels", "norm_num_groups", "sample_size", "scaling_factor",
                "shift_factor", "use_quant_conv", "use_post_quant_conv", "force_upcast", "mid_block_add_attention")
        return cls(**{k: config_dict[k] for k in keys if k in config_dict})

    def encoder_flops(self, H, W):
        ctx = self._context()
        return ctx.lib.vt_encoder_flops(ctx.handle, H, W)
