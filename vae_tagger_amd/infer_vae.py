"""Counterpart of the reference's infer_vae.py: encode images to latents on MI355X and write
latent_vectors.json ({path: [flattened latent floats, C-major]}).  Same flags plus --batch_size.
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


def infer_and_save_latents(args):
    if not torch.cuda.is_available():
        raise RuntimeError("vae_tagger_amd needs an MI355X (no HIP device visible; there is no CPU fallback)")
    device = "cuda"
    print(f"Using device: {device}")
    vae_model = load_vae(args, device)
    vae_model.check_finite = False          # this loop polls the status word itself (and redoes a batch with fp32 storage)
    transform = get_image_transform(args.resolution)
    if not os.path.exists(args.image_path):
        raise FileNotFoundError(f"图像路径未找到: {args.image_path}")
    image_paths = get_image_paths(args.image_path)
    if not image_paths:
        print("未找到任何图像文件，请检查路径。")
        return
    from PIL import Image
    latent_data, processed, errors = {}, 0, 0
    bs = max(1, int(getattr(args, "batch_size", 8)))
    for start in range(0, len(image_paths), bs):
        batch, names = [], []
        for p in image_paths[start:start + bs]:
            try:
                batch.append(transform(Image.open(p).convert("RGB")))
                names.append(p)
            except Exception as e:  # noqa: BLE001 - skip-and-count (infer_vae.py:70-72)
                errors += 1
                print(f"跳过图像 {p}，错误原因: {e}")
        if not batch:
            continue
        try:
            x = torch.stack(batch).to(device)
            latent = vae_model.encode(x)
            flat = latent.reshape(latent.size(0), -1).cpu().numpy()
            ctx = vae_model.vae._context()
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
            for k, p in enumerate(names):
                latent_data[str(p)] = flat[k].tolist()
                processed += 1
        except Exception as e:  # noqa: BLE001
            errors += len(names)
            print(f"跳过图像 {[str(n) for n in names]}，错误原因: {e}")
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
    return p


def main(argv=None):
    return infer_and_save_latents(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
