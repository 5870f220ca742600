"""Counterpart of the reference's infer_vae.py: encode images to latents on MI355X and write
latent_vectors.json ({path: [flattened latent floats, C-major]}).  Same flags plus --batch_size, --workers, --host_resize; the loop is
pipelined (prefetch.py: decode pool + side-stream upload / resize while the previous batch encodes) and shards over ranks under torchrun.
Reference: infer_vae.py:15-29 (load_vae), :31-81 (infer_and_save_latents), :83-92 (flags)."""
import argparse
import json
import os
from pathlib import Path

import numpy as np
import torch

from ._lib import VT_STATUS_FP8_SATURATED, VT_STATUS_NONFINITE
from ._runtime import stream_ptr
from .diffusers_vae_loader import (DiffusersVAEWrapper, create_vae_from_config_file, get_diffusers_vae_config,
                                   load_diffusers_vae_from_config)
from .modules import get_image_paths, get_image_transform


def load_vae(args, device="cuda"):
    if args.vae_config_path and os.path.exists(args.vae_config_path):
        print(f"从配置文件创建VAE: {args.vae_config_path}")
        model = create_vae_from_config_file(args.vae_config_path, args.vae_checkpoint)
    elif args.vae_checkpoint and os.path.exists(args.vae_checkpoint):
        print(f"直接加载预训练VAE模型: {args.vae_checkpoint}")
        model = DiffusersVAEWrapper(load_diffusers_vae_from_config(get_diffusers_vae_config(), args.vae_checkpoint))
    else:
        raise RuntimeError("必须提供 VAE 模型检查点或配置文件")
    model.to(device)
    model.eval()
    return model


class _EncoderInput:
    """What prefetch.BatchFeeder needs from a pipeline object (device, uint8 resize / normalise on the current stream), for the VAE alone:
    infer_vae has no decoder, so it drives the VAE mirror's own context through the same vt_resize_u8 / vt_preprocess_u8 entry points."""
    FILTER_BILINEAR, FILTER_LANCZOS = 0, 1

    def __init__(self, vae_model):
        import ctypes
        from ._runtime import vp, workspace
        self._ct, self._vp, self._ws = ctypes, vp, workspace
        self.ctx = vae_model.vae._context()
        self.device = next(vae_model.vae.parameters()).device

    def resize_u8_into(self, src, out, filt, box=None, tag="resize"):
        H, W, _ = src.shape
        oh, ow, _ = out.shape
        left, top, cw, ch = box if box is not None else (0, 0, W, H)
        need = self.ctx.lib.vt_resize_workspace_bytes(ch, cw, oh, ow, filt)
        ws, ptr = self._ws(self.device, need, tag)
        self.ctx.call("vt_resize_u8", self._vp(src), H, W, left, top, cw, ch, self._vp(out), oh, ow, filt, self._ct.c_void_p(ptr), need,
                      stream_ptr(self.device))
        return out

    def normalize_u8(self, u8):
        B, H, W, _ = u8.shape
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device)
        self.ctx.call("vt_preprocess_u8", self._vp(u8), B, H, W, self._vp(out), stream_ptr(self.device))
        return out


def infer_and_save_latents(args):
    from .infer_full import _dist_setup, _grouped, gather_results
    world, rank, dev_index = _dist_setup()
    if not torch.cuda.is_available():
        raise RuntimeError("vae_tagger_amd needs an MI355X (no HIP device visible; there is no CPU fallback)")
    device = "cuda" if dev_index is None else f"cuda:{dev_index}"
    if dev_index is not None:
        torch.cuda.set_device(dev_index)
    print(f"Using device: {device}")
    vae_model = load_vae(args, device)
    vae_model.check_finite = False          # this loop polls the status word itself (and redoes a batch with fp32 storage)
    if getattr(args, "fp16_operands", False):
        vae_model.vae.set_fp16_operands(True)
    transform = get_image_transform(args.resolution)
    if not os.path.exists(args.image_path):
        raise FileNotFoundError(f"图像路径未找到: {args.image_path}")
    image_paths = get_image_paths(args.image_path)
    if _grouped():
        import torch.distributed as dist
        box = [image_paths]
        dist.broadcast_object_list(box, src=0)
        image_paths = box[0]
    if not image_paths:
        print("未找到任何图像文件，请检查路径。")
        return
    from collections import deque
    from . import sharding
    from .prefetch import BatchFeeder
    lo, hi = sharding.shard_range(len(image_paths), rank, world)
    my_paths = image_paths[lo:hi]
    latent_data, processed, errors = {}, 0, 0
    bs = max(1, int(getattr(args, "batch_size", 8)))
    ctx = vae_model.vae._context()
    inp = _EncoderInput(vae_model)
    main = torch.cuda.current_stream(inp.device)

    def encode_checked(x):
        """Synchronous leg: latents of one batch as a host array, with the health word's fall-backs (the serial loop's logic)."""
        latent = vae_model.encode(x)
        flat = latent.reshape(latent.size(0), -1).cpu().numpy()
        st = ctx.status(stream=stream_ptr(latent.device))
        if st & VT_STATUS_FP8_SATURATED:
            # (only when somebody switched this context to fp8 mode: latents are outside that mode's claim)
            print("警告: 激活值超出fp8(e4m3)范围，改用bf16路径重新计算该批次")
            ctx.call("vt_set_flag", 11, 0)
            flat = vae_model.encode(x).reshape(latent.size(0), -1).cpu().numpy()
            st = ctx.status(stream=stream_ptr(latent.device))
        if st & VT_STATUS_NONFINITE:
            # an activation left the fp16 range of the residual-stream storage: keep fp32 storage from here on
            print("警告: 激活值超出fp16范围，改用fp32残差存储重新计算该批次")
            ctx.call("vt_set_flag", 4, 0)
            flat = vae_model.encode(x).reshape(latent.size(0), -1).cpu().numpy()
            if ctx.status(stream=stream_ptr(latent.device)) & VT_STATUS_NONFINITE:
                ctx.call("vt_set_flag", 4, 1)
                raise FloatingPointError("non-finite activations even with fp32 residual storage (inf / NaN pixels or weights?)")
        # the status word only sees GroupNorm statistics and e4m3 clamps: values that go bad after the last norm (conv_out
        # weights, the scale / shift stage) are caught on the array that is on the host anyway
        if not np.isfinite(flat).all():
            raise FloatingPointError("non-finite latents (inf / NaN in the weights behind the last GroupNorm?)")
        return flat

    def enqueue(x, names):
        rec = {"x": x, "names": names, "ok": False}
        try:
            latent = vae_model.encode(x)
            host = torch.empty(latent.size(0), latent[0].numel(), dtype=torch.float32, pin_memory=True)
            host.copy_(latent.reshape(latent.size(0), -1), non_blocking=True)
            word = torch.empty(1, dtype=torch.int32, pin_memory=True)
            ctx.call("vt_status_async", 1, word.data_ptr(), stream_ptr(latent.device))
            ev = torch.cuda.Event()
            ev.record(main)
            rec.update(host=host, word=word, ev=ev, ok=True)
        except Exception as e:  # noqa: BLE001 - resolved by the synchronous leg
            rec["error"] = e
        return rec

    def finish(rec):
        nonlocal processed, errors
        names = rec["names"]
        try:
            flat = None
            if rec["ok"]:
                rec["ev"].synchronize()
                if int(rec["word"][0]) == 0:
                    flat = rec["host"].numpy()
                    if not np.isfinite(flat).all():
                        flat = None
            if flat is None:
                try:
                    ctx.status(stream=stream_ptr(inp.device))            # (clear what this batch raised before the checked rerun)
                except Exception:  # noqa: BLE001
                    pass
                flat = encode_checked(rec["x"])
            for k, p in enumerate(names):
                latent_data[str(p)] = flat[k].tolist()
                processed += 1
        except Exception as e:  # noqa: BLE001 - skip-and-count (infer_vae.py:70-72)
            errors += len(names)
            print(f"跳过图像 {[str(n) for n in names]}，错误原因: {e}")

    inflight = deque()
    feeder = BatchFeeder(inp, my_paths, bs, args.resolution, workers=getattr(args, "workers", None) or None,
                         host_resize=bool(getattr(args, "host_resize", False)), transform=transform)
    for names, x, ready, failed in feeder:
        for p, e in failed:
            errors += 1
            print(f"跳过图像 {p}，错误原因: {e}")
        if names:
            main.wait_event(ready)
            x.record_stream(main)
            inflight.append(enqueue(x, names))
        while len(inflight) > 1:
            finish(inflight.popleft())
    while inflight:
        finish(inflight.popleft())
    items = [(str(p), latent_data[str(p)]) for p in my_paths if str(p) in latent_data]
    items, processed, errors = gather_results(items, processed, errors, world, rank)
    if rank != 0:
        return None
    latent_data = dict(items)
    print(f"处理完成！成功: {processed}, 失败: {errors}, 总计: {len(image_paths)}")
    out = Path(args.output_dir) / "latent_vectors.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out, "w") as f:
        json.dump(latent_data, f, indent=4)
    print(f"潜在向量已保存到: {out}")
    return latent_data


def build_parser():
    p = argparse.ArgumentParser(description="使用VAE模型进行推理，输出潜在向量。")
    p.add_argument("--vae_checkpoint", type=str, required=True, help="预训练VAE模型文件路径 (.safetensors)")
    p.add_argument("--vae_config_path", type=str, default=None, help="VAE配置文件路径 (JSON格式)")
    p.add_argument("--image_path", type=str, required=True, help="单个图像文件或包含图像的目录")
    p.add_argument("--output_dir", type=str, default="inference_output", help="潜在向量保存目录")
    p.add_argument("--resolution", type=int, default=1024, help="VAE模型训练时的分辨率")
    p.add_argument("--batch_size", type=int, default=8, help="images per device batch (not in the reference)")
    p.add_argument("--host_resize", action="store_true",
                   help="the reference's own route: PIL Resize + ToTensor + Normalize on the CPU (same latents; default: uint8 pixels over PCIe, "
                        "Pillow's resample reproduced bit for bit on the GPU)")
    p.add_argument("--workers", type=int, default=0, help="image decode threads (0 = min(16, cores); not in the reference)")
    p.add_argument("--fp16_operands", action="store_true",
                   help="fp16 instead of bf16 MFMA operands for the convolutions: latents ~6x closer to the fp32 reference (smooth pictures stay "
                        "inside 1e-2), ~4 %% slower (not in the reference)")
    return p


def main(argv=None):
    return infer_and_save_latents(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
