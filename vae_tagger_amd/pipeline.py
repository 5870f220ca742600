"""Batched encode+tag pipeline: the loop body of infer_full.py:95-105 for a whole batch, one
vt_context holding both models, one C call per batch (vt_encode_tag), plus the data-parallel
sharding of SURVEY.md section 8e (images are independent; the only exchange is an all-gather of logits).
"""
import ctypes

import torch

from . import _lib
from ._runtime import as_input, stream_ptr, vp, workspace
from .autoencoder_kl import AutoencoderKL
from .diffusers_vae_loader import DiffusersVAEWrapper
from .modules import _HipDecoder, smart_crop_box


class EncodeTagPipeline:
    # vt_encode_tag never synchronises the host: a residual stream that left the fp16 range of its storage, or (fp8 mode) an activation that was
    # clamped to the e4m3 range, only raises the context's sticky word (vt_status).  Like the reference-shaped wrapper
    # (DiffusersVAEWrapper.check_finite), the pipeline reads the word after every logits() / tag() -- one 4-byte copy and a stream synchronise -- and
    # raises, so no caller gets non-finite or clamped tags silently.  Callers that synchronise elsewhere and poll the word themselves (the pipelined
    # CLIs through status_async, bench.py, tools/) set check_finite = False.
    check_finite = True

    def __init__(self, vae_model, decoder, device=None):
        vae = vae_model.vae if isinstance(vae_model, DiffusersVAEWrapper) else vae_model
        if not isinstance(vae, AutoencoderKL) or not isinstance(decoder, _HipDecoder):
            raise TypeError("EncodeTagPipeline needs vae_tagger_amd's AutoencoderKL (or its wrapper) and decoder")
        self.vae, self.decoder = vae, decoder
        dev = next(vae.parameters()).device if device is None else torch.device(device)
        if dev.type != "cuda":
            raise _lib.VTError("EncodeTagPipeline runs on a HIP device only; move the models with .to('cuda')")
        self.device = dev
        self.ctx = _lib.Context(dev.index if dev.index is not None else torch.cuda.current_device())
        vae._upload(self.ctx)
        decoder._upload(self.ctx)
        self.num_classes = decoder.num_classes
        self.latent_channels = vae.config.latent_channels
        self._nd = len(vae.config.block_out_channels) - 1

    def flops_per_image(self, H, W):
        return self.ctx.lib.vt_encoder_flops(self.ctx.handle, H, W)

    @torch.no_grad()
    def logits(self, x, return_latent=False):
        """x fp32 [B,3,H,W] on the device -> logits fp32 [B,N] (and the scaled latent if asked)."""
        x = as_input(x)
        if x.device != self.device:
            x = x.to(self.device)
        B, _, H, W = x.shape
        out = torch.empty(B, self.num_classes, dtype=torch.float32, device=self.device)
        lat = None
        if return_latent:
            lat = torch.empty(B, self.latent_channels, H >> self._nd, W >> self._nd, dtype=torch.float32,
                              device=self.device)
        need = self.ctx.lib.vt_encode_tag_workspace_bytes(self.ctx.handle, B, H, W)
        if need == 0:
            raise _lib.VTError(f"unsupported batch shape {tuple(x.shape)}")
        ws, ptr = workspace(self.device, need)
        self.ctx.call("vt_encode_tag", vp(x), B, H, W, vp(lat), vp(out), ctypes.c_void_p(ptr), need,
                      stream_ptr(self.device))
        if self.check_finite:
            self.raise_on_status()
        return (out, lat) if return_latent else out

    def raise_on_status(self):
        """Read (and clear) the sticky health word -- synchronises -- and raise what it says."""
        st = self.status()
        if st & _lib.VT_STATUS_NONFINITE:
            raise FloatingPointError("non-finite activations in the encoder: the fp16 residual-stream storage overflowed "
                                     "(vt_set_flag(ctx, 4, 0) stores it as fp32) or the input / checkpoint holds inf / NaN")
        if st & _lib.VT_STATUS_FP8_SATURATED:
            raise FloatingPointError("fp8 mode: activations exceeded the e4m3 range and were clamped (set_fp8(False) returns to the bf16 path)")

    @torch.no_grad()
    def confidence(self, logits):
        logits = logits.contiguous()
        B, N = logits.shape
        conf = torch.empty_like(logits)
        idx = torch.empty(B, N, dtype=torch.int64, device=logits.device)
        self.ctx.call("vt_get_confidence", vp(logits), B, N, vp(conf), vp(idx), stream_ptr(logits.device))
        return conf, idx

    @torch.no_grad()
    def summarize_async(self, conf, idx, threshold, top_k=64):
        """Device-side counterpart of the per-image loop of infer_full.py:106-125 on sorted (conf, idx), WITHOUT a host synchronisation:
        launches vt_summarize_confidence and an asynchronous copy of its B x (2K + 4) values into pinned host memory on the current stream.
        Returns (packed pinned tensor, K); `unpack_summary` reads it once the stream has passed this point (an event / a sync)."""
        B, N = conf.shape
        K = int(max(5, min(top_k, N)))
        tc = torch.empty(B, K, dtype=torch.float32, device=conf.device)
        ti = torch.empty(B, K, dtype=torch.int32, device=conf.device)
        st = torch.empty(B, 4, dtype=torch.float32, device=conf.device)
        self.ctx.call("vt_summarize_confidence", vp(conf), vp(idx), B, N, float(threshold), K, vp(tc), vp(ti), vp(st),
                      stream_ptr(conf.device))
        packed = torch.cat([tc, ti.view(torch.float32), st], dim=1)
        host = torch.empty(packed.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(packed, non_blocking=True)                                         # one D2H copy
        return host, K

    @staticmethod
    def unpack_summary(host, K):
        """(top_conf [B,K] fp32, top_idx [B,K] int32, stats [B,4] = count >= threshold, max, top-5 sum / 5, non-finite count) as host arrays."""
        return (host[:, :K].numpy(), host[:, K:2 * K].contiguous().view(torch.int32).numpy(), host[:, 2 * K:].numpy())

    @torch.no_grad()
    def summarize(self, conf, idx, threshold, top_k=64):
        """`summarize_async` + wait: host arrays from ONE small copy -- B x (2K + 4) values instead of the B x N x 12 bytes of the full
        sorted arrays."""
        host, K = self.summarize_async(conf, idx, threshold, top_k)
        torch.cuda.current_stream(conf.device).synchronize()
        return self.unpack_summary(host, K)

    def status(self, clear=True):
        """Sticky health word of the context (synchronises): bit 0 = non-finite GroupNorm statistics were seen, i.e. an
        activation left the fp16 range of the residual-stream storage (or the weights hold inf / NaN); bit 1 (fp8 mode) = an
        activation exceeded the e4m3 range and was clamped -- this checkpoint needs the bf16 path (set_fp8(False))."""
        return self.ctx.status(clear, stream_ptr(self.device))

    def set_fp8(self, on=True):
        """BASELINE configs[4]: the 3x3 convs of the resnet / downsample stack on fp8 (e4m3) operands (vt_set_flag 11); tagging only."""
        self.ctx.call("vt_set_flag", 11, 1 if on else 0)

    def set_fp16_operands(self, on=True):
        """fp16 instead of bf16 MFMA operands for the convolutions (vt_set_flag 18): same bytes, 11 significand bits instead of 8 -- latents
        ~6x closer to the fp32 reference, about 4 % fewer images/s (the matrix pipe draws more power on fp16 data).  Not in fp8 mode."""
        self.ctx.call("vt_set_flag", 18, 1 if on else 0)

    def set_fp32_residual(self, on=True):
        """Store the residual stream as fp32 (and conv1 outputs as bf16) instead of fp16: for checkpoints whose activations
        exceed +-65504 (about 7 % slower)."""
        self.ctx.call("vt_set_flag", 4, 0 if on else 1)

    @torch.no_grad()
    def normalize_u8(self, u8_hwc):
        """ToTensor + Normalize(0.5, 0.5) on the device: uint8 [B,H,W,3] -> fp32 NCHW in [-1,1] (modules.py:136-140).
        Uploading uint8 moves 4x fewer bytes over PCIe than the fp32 tensor the reference builds on the CPU."""
        u8 = u8_hwc.to(self.device).contiguous()
        if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[-1] != 3:
            raise ValueError(f"expected uint8 [B,H,W,3], got {u8.dtype} {tuple(u8.shape)}")
        B, H, W, _ = u8.shape
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device)
        self.ctx.call("vt_preprocess_u8", vp(u8), B, H, W, vp(out), stream_ptr(self.device))
        return out

    # ---- device-side resize: the reference's PIL transforms (modules.py:126-178) reproduced bit for bit on the GPU ----
    FILTER_BILINEAR, FILTER_LANCZOS = 0, 1

    @torch.no_grad()
    def resize_u8(self, u8_hwc, out_w, out_h, filt, box=None):
        """uint8 [H,W,3] (tensor or array) -> uint8 [out_h,out_w,3] on the device = Image.resize((out_w, out_h), filt)
        of the image cropped to box = (left, top, width, height); Pillow's 8-bit two-pass resample, exactly."""
        t = torch.as_tensor(u8_hwc)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[-1] != 3:
            raise ValueError(f"expected uint8 [H,W,3], got {t.dtype} {tuple(t.shape)}")
        t = t.to(self.device).contiguous()
        H, W, _ = t.shape
        left, top, cw, ch = box if box is not None else (0, 0, W, H)
        out = torch.empty(out_h, out_w, 3, dtype=torch.uint8, device=self.device)
        need = self.ctx.lib.vt_resize_workspace_bytes(ch, cw, out_h, out_w, filt)
        if need == 0:
            raise _lib.VTError(f"unsupported resize {cw}x{ch} -> {out_w}x{out_h}")
        ws, ptr = workspace(self.device, need, "resize")
        self.ctx.call("vt_resize_u8", vp(t), H, W, left, top, cw, ch, vp(out), out_h, out_w, filt, ctypes.c_void_p(ptr), need,
                      stream_ptr(self.device))
        return out

    @torch.no_grad()
    def resize_u8_into(self, src, out, filt, box=None, tag="resize"):
        """`resize_u8` for a uint8 [H,W,3] DEVICE tensor `src` into the caller's contiguous uint8 [out_h,out_w,3] device view `out`
        (one slot of a batch buffer), on the current stream; `tag` names the scratch buffer (one per stream that resizes)."""
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[-1] != 3 or not src.is_contiguous() or src.device != self.device:
            raise ValueError(f"expected a contiguous uint8 [H,W,3] tensor on {self.device}, got {src.dtype} {tuple(src.shape)} on {src.device}")
        if out.dtype != torch.uint8 or out.dim() != 3 or out.shape[-1] != 3 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("output must be a contiguous uint8 [h,w,3] view on the same device")
        H, W, _ = src.shape
        out_h, out_w, _ = out.shape
        left, top, cw, ch = box if box is not None else (0, 0, W, H)
        need = self.ctx.lib.vt_resize_workspace_bytes(ch, cw, out_h, out_w, filt)
        if need == 0:
            raise _lib.VTError(f"unsupported resize {cw}x{ch} -> {out_w}x{out_h}")
        ws, ptr = workspace(self.device, need, tag)
        self.ctx.call("vt_resize_u8", vp(src), H, W, left, top, cw, ch, vp(out), out_h, out_w, filt, ctypes.c_void_p(ptr), need,
                      stream_ptr(self.device))
        return out

    def status_async(self, out, clear=True):
        """The context's health word copied (and cleared) in stream order into `out` (int32 [1], pinned host or device) without a host
        synchronisation (vt_status_async): valid once the current stream has passed this point."""
        if out.dtype != torch.int32 or out.numel() < 1:
            raise ValueError("status_async: int32 tensor expected")
        self.ctx.call("vt_status_async", int(clear), ctypes.c_void_p(out.data_ptr()), stream_ptr(self.device))

    def load_image(self, img, resolution=None, bucket=None):
        """Device counterpart of `get_image_transform(resolution, bucket is not None, bucket)(img)`: a PIL image (or a
        uint8 HWC array) -> fp32 [3,H,W] in [-1,1] on the device.  Only the decoded uint8 pixels cross PCIe; the
        distorting bilinear `Resize((r, r))` or SmartResize's centre crop + LANCZOS run on the GPU with Pillow's arithmetic."""
        import numpy as np
        a = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img, dtype=np.uint8)
        H, W, _ = a.shape
        if bucket is not None:
            tw, th = bucket
            box = smart_crop_box(W, H, tw, th)
            u8 = self.resize_u8(torch.from_numpy(a.copy()), tw, th, self.FILTER_LANCZOS, box)
        else:
            u8 = self.resize_u8(torch.from_numpy(a.copy()), resolution, resolution, self.FILTER_BILINEAR)
        return self.normalize_u8(u8[None])[0]

    def tag(self, x):
        return self.confidence(self.logits(x))
