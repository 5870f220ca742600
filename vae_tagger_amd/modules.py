"""Drop-in counterparts of the hot-path pieces of the reference's modules.py, backed by HIP.

  AttentionClassificationDecoder / ClassificationDecoder / create_attention_decoder
        reference modules.py:358-475, :303-356, :731-748 -- same constructor arguments, same
        state_dict keys, forward(latent) -> logits, get_confidence(latent) -> (sorted conf, indices)
  get_vae_latent_info                    modules.py:244-254
  get_image_transform / SmartResize      modules.py:126-178  (PIL + torch, no torchvision here)
  AspectRatioBucketing                   modules.py:180-222
  get_image_paths                        modules.py:265-286
"""
import ctypes
import os
from pathlib import Path

import torch

from . import synth
from ._runtime import HipModule, stream_ptr, vp, workspace


# ---------------------------------------------------------------------------------------------------
# decoders
class _HipDecoder(HipModule):
    num_classes = 0
    _cfg = (0, 0, 0, 0, 8)      # plain, spatial, self, cross, heads

    def _upload(self, ctx):
        plain, sp, sa, ca, heads = self._cfg
        ctx.call("vt_decoder_configure", self.num_classes, self.latent_channels, plain, sp, sa, ca, heads)
        for k, v in self.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            ctx.set_weight(k, v)
        ctx.call("vt_decoder_finalize")

    def _latent(self, latent_vectors):
        dev = next(self.parameters()).device
        x = latent_vectors.detach().to(device=dev, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != self.latent_channels:
            raise ValueError(f"expected latent [B,{self.latent_channels},h,w], got {tuple(x.shape)}")
        return x

    @torch.no_grad()
    def forward(self, latent_vectors):
        ctx = self._context()
        x = self._latent(latent_vectors)
        B, _, h, w = x.shape
        logits = torch.empty(B, self.num_classes, dtype=torch.float32, device=x.device)
        need = ctx.lib.vt_decode_workspace_bytes(ctx.handle, B, h, w)
        ws, ptr = workspace(x.device, need)
        ctx.call("vt_decode_logits", vp(x), B, h, w, vp(logits), ctypes.c_void_p(ptr), need, stream_ptr(x.device))
        return logits

    @torch.no_grad()
    def confidence_from_logits(self, logits):
        """sigmoid + descending sort on the device (modules.py:470-475).  The reference's torch.sort is
        not stable; here ties are ordered by ascending tag index."""
        ctx = self._context()
        logits = logits.contiguous()
        B, N = logits.shape
        conf = torch.empty_like(logits)
        idx = torch.empty(B, N, dtype=torch.int64, device=logits.device)
        ctx.call("vt_get_confidence", vp(logits), B, N, vp(conf), vp(idx), stream_ptr(logits.device))
        return conf, idx

    def get_confidence(self, latent_vectors):
        return self.confidence_from_logits(self(latent_vectors))


class ClassificationDecoder(_HipDecoder):
    def __init__(self, latent_channels, latent_height, latent_width, num_classes, use_adaptive_pooling=True):
        if not use_adaptive_pooling:
            raise NotImplementedError("use_adaptive_pooling=False is never used by the reference's inference path")
        super().__init__(synth.plain_decoder_manifest(num_classes, latent_channels))
        self.latent_channels, self.latent_height, self.latent_width = latent_channels, latent_height, latent_width
        self.num_classes = num_classes
        self.use_adaptive_pooling = True
        self._cfg = (1, 0, 0, 0, 8)


class AttentionClassificationDecoder(_HipDecoder):
    def __init__(self, latent_channels, latent_height, latent_width, num_classes, use_spatial_attention=True,
                 use_self_attention=True, use_cross_attention=False, attention_heads=8, attention_dropout=0.1):
        super().__init__(synth.attention_decoder_manifest(num_classes, latent_channels, use_spatial_attention,
                                                          use_self_attention, use_cross_attention))
        # latent_height/width are stored but size no layer (adaptive pooling): one set of weights
        # serves every bucket
        self.latent_channels, self.latent_height, self.latent_width = latent_channels, latent_height, latent_width
        self.num_classes = num_classes
        self.use_spatial_attention = use_spatial_attention
        self.use_self_attention = use_self_attention
        self.use_cross_attention = use_cross_attention
        self._cfg = (0, int(use_spatial_attention), int(use_self_attention), int(use_cross_attention),
                     int(attention_heads))
        if use_spatial_attention:
            print("启用spatial attn机制")
        if use_self_attention:
            print("启用多头self attn机制（8x8压缩后）")
        if use_cross_attention:
            print("启用cross attn机制")

    def get_attention_maps(self, latent_vectors):
        return {}


def create_attention_decoder(latent_channels, latent_height, latent_width, num_classes, attention_config=None):
    if attention_config is None:
        print("使用标准分类解码器")
        return ClassificationDecoder(latent_channels, latent_height, latent_width, num_classes)
    print("使用注意力增强分类解码器")
    g = attention_config.get
    return AttentionClassificationDecoder(
        latent_channels=latent_channels, latent_height=latent_height, latent_width=latent_width,
        num_classes=num_classes, use_spatial_attention=g("use_spatial_attention", True),
        use_self_attention=g("use_self_attention", True), use_cross_attention=g("use_cross_attention", False),
        attention_heads=g("attention_heads", 8), attention_dropout=g("attention_dropout", 0.1))


def get_vae_latent_info(resolution, latent_channels=16):
    downsample_factor = 8
    lh = lw = resolution // downsample_factor
    return {"latent_channels": latent_channels, "latent_height": lh, "latent_width": lw,
            "total_dim": latent_channels * lh * lw}


# ---------------------------------------------------------------------------------------------------
# input side
def _to_normalized_tensor(img):
    """ToTensor + Normalize([0.5]*3, [0.5]*3): uint8 HWC -> fp32 CHW in [-1, 1]."""
    import numpy as np
    a = np.asarray(img.convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
    return (t - 0.5) / 0.5


def smart_crop_box(width, height, target_width, target_height, offset=lambda slack: slack // 2):
    """(left, top, crop_w, crop_h) SmartResize cuts out of a width x height image before resizing to the target
    (modules.py:149-178): the wider side is cropped to int(other * ratio), `offset(slack)` picks where (centre by default)."""
    target_ratio = target_width / target_height
    ratio = width / height
    if ratio > target_ratio:
        nw = int(height * target_ratio)
        return offset(width - nw), 0, nw, height
    if ratio < target_ratio:
        nh = int(width / target_ratio)
        return 0, offset(height - nh), width, nh
    return 0, 0, width, height


class SmartResize:
    """Centre/random/edge crop to the bucket's aspect ratio, then LANCZOS resize (modules.py:142-178)."""

    def __init__(self, target_width, target_height, crop_mode="center"):
        self.target_width, self.target_height, self.crop_mode = target_width, target_height, crop_mode

    def _offset(self, slack):
        if self.crop_mode == "center":
            return slack // 2
        if self.crop_mode == "random":
            import random
            return random.randint(0, slack)
        return 0

    def __call__(self, img):
        from PIL import Image
        ow, oh = img.size
        left, top, cw, ch = smart_crop_box(ow, oh, self.target_width, self.target_height, self._offset)
        if (cw, ch) != (ow, oh):
            img = img.crop((left, top, left + cw, top + ch))
        return img.resize((self.target_width, self.target_height), Image.LANCZOS)


def get_image_transform(resolution, use_bucketing=False, aspect_ratio_bucket=None):
    from PIL import Image
    if use_bucketing and aspect_ratio_bucket is not None:
        tw, th = aspect_ratio_bucket
        resize = SmartResize(tw, th)
        return lambda img: _to_normalized_tensor(resize(img))
    # torchvision Resize((r, r)) on a PIL image = distorting bilinear resize
    return lambda img: _to_normalized_tensor(img.resize((resolution, resolution), Image.BILINEAR))


class AspectRatioBucketing:
    def __init__(self, base_resolution=512, max_resolution=1024, bucket_step=64):
        self.base_resolution, self.max_resolution, self.bucket_step = base_resolution, max_resolution, bucket_step
        self.buckets = self._generate_buckets()
        self.image_buckets = {}

    def _generate_buckets(self):
        r = range(self.base_resolution, self.max_resolution + 1, self.bucket_step)
        cap = self.max_resolution * self.max_resolution
        return sorted((w, h) for w in r for h in r if w * h <= cap)

    def bucket_for_ratio(self, ratio):
        """Nearest bucket aspect ratio; strict '<' so ties go to the first (smallest) bucket."""
        best, best_diff = None, float("inf")
        for w, h in self.buckets:
            d = abs(w / h - ratio)
            if d < best_diff:
                best, best_diff = (w, h), d
        return best

    def assign_bucket(self, image_path):
        from PIL import Image
        try:
            with Image.open(image_path) as img:
                ow, oh = img.size
            bucket = self.bucket_for_ratio(ow / oh)
            self.image_buckets[image_path] = bucket
            return bucket
        except Exception as e:  # noqa: BLE001 - reference behaviour: warn and fall back
            print(f"警告: 无法分析图像 {image_path}: {e}")
            return (self.base_resolution, self.base_resolution)

    def get_bucket_statistics(self):
        counts = {}
        for b in self.image_buckets.values():
            counts[b] = counts.get(b, 0) + 1
        return counts


def get_image_paths(path):
    exts = [".png", ".jpg", ".jpeg", ".bmp", ".tiff", ".webp"]
    if os.path.isdir(path):
        found = set()
        for ext in exts:
            for pat in (f"*{ext}", f"*{ext.upper()}"):
                for p in Path(path).rglob(pat):
                    found.add(p.resolve())
        return list(found)
    if os.path.isfile(path):
        if any(path.lower().endswith(e) for e in exts):
            return [Path(path)]
        print(f"警告: {path} 不是支持的图像格式")
        return []
    print(f"错误: 路径 {path} 不存在")
    return []
