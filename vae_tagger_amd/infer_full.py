"""Counterpart of the reference's infer_full.py (same flags, same classification_results.json schema,
same skip-and-count behaviour), running encode+tag on MI355X through libvae_tagger_hip.so.

    python -m vae_tagger_amd.infer_full --vae_checkpoint ae.safetensors --decoder_checkpoint dec.pth \
        --image_path imgs/ --tags_csv_path tags.csv [--batch_size 8]

Differences from the reference, all outside the numbers it writes (`--fp8` is the exception: an opt-in faster mode whose logits
stay within 1e-2 of the default path's):
  * images are processed in same-shape batches (`--batch_size`, new flag; the reference runs one at a time);
  * sigmoid + sort run on the device and come back in one copy per batch (the reference does 2*N .item() syncs
    per image, infer_full.py:109-111);
  * checkpoints are read with tensor-only loaders (safetensors / torch.load(weights_only=True)).
Reference: infer_full.py:16-71 (load_models), :73-141 (infer_and_classify), :143-186 (flags).
"""
import argparse
import json
import os
from pathlib import Path

import torch

from .diffusers_vae_loader import (DiffusersVAEWrapper, create_vae_from_config_file, get_diffusers_vae_config,
                                   load_diffusers_vae_from_config)
from .modules import (ClassificationDecoder, create_attention_decoder, get_image_paths, get_image_transform,
                      get_vae_latent_info)
from ._lib import VT_STATUS_FP8_SATURATED, VT_STATUS_NONFINITE
from .pipeline import EncodeTagPipeline


def load_state_dict_file(path):
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(str(path))
    return torch.load(path, map_location="cpu", weights_only=True)


def load_models(args, device="cuda"):
    if args.vae_config_path and os.path.exists(args.vae_config_path):
        print(f"从配置文件创建VAE: {args.vae_config_path}")
        vae_model = create_vae_from_config_file(args.vae_config_path, args.vae_checkpoint)
    elif args.vae_checkpoint and os.path.exists(args.vae_checkpoint):
        print(f"直接加载预训练VAE模型: {args.vae_checkpoint}")
        vae_model = DiffusersVAEWrapper(load_diffusers_vae_from_config(get_diffusers_vae_config(), args.vae_checkpoint))
    else:
        raise RuntimeError("必须提供 VAE 模型检查点或配置文件")
    vae_model.to(device)
    vae_model.eval()
    latent_info = get_vae_latent_info(args.resolution)
    print(f"VAE潜在空间信息: {latent_info}")
    import pandas as pd
    tags_df = pd.read_csv(args.tags_csv_path)
    num_classes = len(tags_df)
    if args.use_attention:
        print("使用注意力分类解码器")
        decoder = create_attention_decoder(
            latent_channels=latent_info["latent_channels"], latent_height=latent_info["latent_height"],
            latent_width=latent_info["latent_width"], num_classes=num_classes,
            attention_config={"use_spatial_attention": getattr(args, "use_spatial_attention", True),
                              "use_self_attention": getattr(args, "use_self_attention", True),
                              "use_cross_attention": getattr(args, "use_cross_attention", False),
                              "attention_heads": getattr(args, "attention_heads", 8),
                              "attention_dropout": getattr(args, "attention_dropout", 0.1)})
    else:
        print("使用标准分类解码器")
        decoder = ClassificationDecoder(latent_info["latent_channels"], latent_info["latent_height"],
                                        latent_info["latent_width"], num_classes, use_adaptive_pooling=True)
    if not os.path.exists(args.decoder_checkpoint):
        raise RuntimeError(f"解码器模型文件不存在: {args.decoder_checkpoint}")
    try:
        decoder.load_state_dict(load_state_dict_file(args.decoder_checkpoint), strict=False)
        print(f"成功加载Decoder模型: {args.decoder_checkpoint}")
    except Exception as e:  # noqa: BLE001 - reference behaviour
        raise RuntimeError(f"无法加载Decoder模型: {e}")
    decoder.to(device)
    decoder.eval()
    return vae_model, decoder, tags_df["name"].tolist()


TOP_K = 64          # (confidence, tag) pairs fetched per image with the summary; images with more tags above the threshold fetch their prefix


def summarize(conf_row, idx_row, tag_names, threshold):
    """One image's JSON entry from its sorted confidences / indices on the host (infer_full.py:106-125) -- the reference
    formulation; the CLI uses `summarize_batch`, which takes the same numbers from the device-side summary."""
    predicted = []
    for c, i in zip(conf_row, idx_row):
        c = float(c)
        if c < threshold:
            break                                   # sorted descending: nothing further passes
        predicted.append({"tag": tag_names[int(i)], "confidence": float(f"{c:.4f}")})
    top5 = [float(c) for c in conf_row[:5]]
    return {"predicted_tags": predicted, "total_tags_above_threshold": len(predicted),
            "max_confidence": float(f"{float(conf_row[0]):.4f}"),
            "avg_confidence_top5": float(f"{sum(top5) / 5:.4f}")}      # always divides by 5, like the reference


def summarize_batch(pipe, conf, idx, tag_names, threshold, top_k=TOP_K):
    """conf / idx: sorted device tensors [B,N] (pipe.tag).  Threshold count, top-k, max and top-5 mean come from the
    device (vt_summarize_confidence): formatting only on the host.  Raises FloatingPointError on non-finite confidences."""
    top_conf, top_idx, stats = pipe.summarize(conf, idx, threshold, top_k)
    out = []
    for b in range(conf.shape[0]):
        count, mx, avg5, bad = int(stats[b, 0]), float(stats[b, 1]), float(stats[b, 2]), int(stats[b, 3])
        if bad:
            raise FloatingPointError(f"{bad} non-finite confidences: activations left the fp16 range of the residual stream "
                                     "or the checkpoint holds inf / NaN")
        if count <= top_conf.shape[1]:
            cs, ix = top_conf[b, :count], top_idx[b, :count]
        else:                                       # rare: more tags above the threshold than the summary carries
            cs, ix = conf[b, :count].cpu().numpy(), idx[b, :count].cpu().numpy()
        predicted = [{"tag": tag_names[int(i)], "confidence": float(f"{float(c):.4f}")} for c, i in zip(cs, ix)]
        out.append({"predicted_tags": predicted, "total_tags_above_threshold": count,
                    "max_confidence": float(f"{mx:.4f}"), "avg_confidence_top5": float(f"{avg5:.4f}")})
    return out


def infer_and_classify(args):
    if not torch.cuda.is_available():
        raise RuntimeError("vae_tagger_amd needs an MI355X (no HIP device visible; there is no CPU fallback)")
    device = "cuda"
    print(f"Using device: {device}")
    vae_model, decoder, tag_names = load_models(args, device)
    transform = get_image_transform(args.resolution)
    if not os.path.exists(args.image_path):
        raise FileNotFoundError(f"图像路径未找到: {args.image_path}")
    image_paths = get_image_paths(args.image_path)
    if not image_paths:
        print("未找到任何图像文件，请检查路径。")
        return
    pipe = EncodeTagPipeline(vae_model, decoder)
    if getattr(args, "fp8", False):
        pipe.set_fp8(True)
    from PIL import Image
    results, processed, errors = {}, 0, 0
    bs = max(1, int(getattr(args, "batch_size", 8)))

    def load(p):
        img = Image.open(p).convert("RGB")
        return pipe.load_image(img, resolution=args.resolution) if getattr(args, "device_resize", False) else transform(img)

    def tag_batch(x):
        """One device batch -> JSON entries.  Both status bits are re-read after every run; a mode switch is permanent."""
        conf, idx = pipe.tag(x)
        st = pipe.status()
        if st & VT_STATUS_FP8_SATURATED:
            # --fp8 and this checkpoint's activations exceed the e4m3 range: the clamped values are not worth tags; bf16 from here on
            print("警告: 激活值超出fp8(e4m3)范围，改用bf16路径重新计算该批次")
            pipe.set_fp8(False)
            state["fp8"] = False
            conf, idx = pipe.tag(x)
            st = pipe.status()
        if st & VT_STATUS_NONFINITE:
            # an activation left the fp16 range of the residual-stream storage: keep fp32 storage from here on
            print("警告: 激活值超出fp16范围，改用fp32残差存储重新计算该批次")
            pipe.set_fp32_residual(True)
            conf, idx = pipe.tag(x)
            st = pipe.status()
            if st & VT_STATUS_FP8_SATURATED:            # (fp8 still on and only the fp32-storage run clamps)
                pipe.set_fp8(False)
                state["fp8"] = False
                conf, idx = pipe.tag(x)
                st = pipe.status()
            if st & VT_STATUS_NONFINITE:
                pipe.set_fp32_residual(False)           # storage was not the cause: one bad image must not slow the rest of the run
                raise FloatingPointError("non-finite activations even with fp32 residual storage (inf / NaN pixels or weights?)")
        return summarize_batch(pipe, conf, idx, tag_names, args.confidence_threshold)

    def clear_status():
        try:
            pipe.status(clear=True)                      # the sticky word must not leak into the next healthy batch
        except Exception:  # noqa: BLE001
            pass

    def device_leg(tensors, names):
        """Tag one batch; when the batch fails, retry it image by image so that an error costs ONE image, as in the reference's
        per-image loop (infer_full.py:130-132).  Returns [(path, entry)] and the number of images lost."""
        try:
            return list(zip(names, tag_batch(torch.stack(tensors).to(device)))), 0
        except Exception as e:  # noqa: BLE001
            clear_status()
            if len(names) == 1:
                print(f"跳过图像 {names[0]}，错误原因: {e}")
                return [], 1
        done, lost = [], 0
        for t, p in zip(tensors, names):
            try:
                done.append((p, tag_batch(t[None].to(device))[0]))
            except Exception as e:  # noqa: BLE001 - skip-and-count (infer_full.py:130-132)
                clear_status()
                lost += 1
                print(f"跳过图像 {p}，错误原因: {e}")
        return done, lost

    state = {"fp8": bool(getattr(args, "fp8", False))}
    fp8_done = []                                        # paths whose entries were computed in fp8 mode
    for start in range(0, len(image_paths), bs):
        batch, names = [], []
        for p in image_paths[start:start + bs]:
            try:
                batch.append(load(p))
                names.append(p)
            except Exception as e:  # noqa: BLE001 - skip-and-count (infer_full.py:130-132)
                errors += 1
                print(f"跳过图像 {p}，错误原因: {e}")
        if not batch:
            continue
        was_fp8 = state["fp8"]
        done, lost = device_leg(batch, names)
        errors += lost
        for p, entry in done:
            results[str(p)] = entry
            processed += 1
        if was_fp8 and state["fp8"]:
            fp8_done.extend(p for p, _ in done)
        elif was_fp8 and fp8_done:
            # fp8 was abandoned in this batch: ONE output file holds ONE numeric mode -- the earlier fp8 batches are redone in bf16
            print(f"重新以bf16计算之前的 {len(fp8_done)} 张图像")
            for s0 in range(0, len(fp8_done), bs):
                redo = fp8_done[s0:s0 + bs]
                try:
                    again, lost = device_leg([load(p) for p in redo], redo)
                except Exception as e:  # noqa: BLE001
                    again, lost = [], len(redo)
                    print(f"跳过图像 {[str(n) for n in redo]}，错误原因: {e}")
                for p in redo:
                    results.pop(str(p), None)
                for p, entry in again:
                    results[str(p)] = entry
                processed -= lost
                errors += lost
            fp8_done = []
        if (start // bs + 1) % max(1, 100 // bs) == 0:
            print(f"已处理 {processed}/{len(image_paths)} 图像 (跳过 {errors} 个错误)")
    print(f"处理完成！成功: {processed}, 失败: {errors}, 总计: {len(image_paths)}")
    out = Path(args.output_dir) / "classification_results.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out, "w") as f:
        json.dump(results, f, indent=4, ensure_ascii=False)
    print(f"分类结果已保存到: {out}")
    return results


def build_parser():
    p = argparse.ArgumentParser(description="使用VAE和分类解码器进行图像分类。")
    p.add_argument("--vae_checkpoint", type=str, required=True, help="预训练VAE模型文件路径 (.safetensors)")
    p.add_argument("--vae_config_path", type=str, default=None, help="VAE配置文件路径 (JSON格式)")
    p.add_argument("--decoder_checkpoint", type=str, required=True, help="Decoder模型文件路径 (.bin/.pth)")
    p.add_argument("--image_path", type=str, required=True, help="单个图像文件或包含图像的目录")
    p.add_argument("--tags_csv_path", type=str, required=True, help="包含所有分类头的CSV文件")
    p.add_argument("--output_dir", type=str, default="inference_output", help="结果保存目录")
    p.add_argument("--resolution", type=int, default=1024, help="模型训练时的分辨率")
    p.add_argument("--confidence_threshold", type=float, default=0.5, help="置信度阈值")
    p.add_argument("--use_attention", action="store_true", default=True, help="使用注意力机制 (默认开启)")
    p.add_argument("--no_attention", action="store_true", help="禁用注意力机制")
    p.add_argument("--use_spatial_attention", action="store_true", default=True, help="启用空间注意力")
    p.add_argument("--use_self_attention", action="store_true", default=True, help="启用自注意力")
    p.add_argument("--use_cross_attention", action="store_true", help="启用交叉注意力")
    p.add_argument("--attention_heads", type=int, default=8, help="注意力头数")
    p.add_argument("--attention_dropout", type=float, default=0.1, help="注意力dropout率")
    p.add_argument("--model_checkpoint", type=str, default=None, help="(已弃用) 包含VAE和Decoder权重的父目录")
    p.add_argument("--batch_size", type=int, default=8, help="images per device batch (not in the reference)")
    p.add_argument("--device_resize", action="store_true",
                   help="resize + normalise on the GPU (bit-exact with the PIL transform; not in the reference)")
    p.add_argument("--fp8", action="store_true",
                   help="3x3 convs of the encoder on fp8 (e4m3) operands / the fp8 MFMA: ~1.35x faster, logits within 1e-2 of the "
                        "bf16 path's reference, latents only to ~1e-1 (tagging only; not in the reference)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_attention:
        args.use_attention = False
    if args.model_checkpoint and (not args.vae_checkpoint or not args.decoder_checkpoint):
        print("使用向后兼容模式，从model_checkpoint参数推导VAE和Decoder路径")
        args.vae_checkpoint = args.model_checkpoint
        args.decoder_checkpoint = args.model_checkpoint
    return infer_and_classify(args)


if __name__ == "__main__":
    main()
