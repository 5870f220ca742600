"""CLI counterparts of infer_full.py / infer_vae.py: flag surface (CPU) and end-to-end run on MI355X (GPU)."""
import json

import pytest
import torch

from vae_tagger_amd import infer_full, infer_vae, synth

REF_FULL_FLAGS = {  # reference infer_full.py:143-180
    "--vae_checkpoint", "--vae_config_path", "--decoder_checkpoint", "--image_path", "--tags_csv_path", "--output_dir",
    "--resolution", "--confidence_threshold", "--use_attention", "--no_attention", "--use_spatial_attention",
    "--use_self_attention", "--use_cross_attention", "--attention_heads", "--attention_dropout", "--model_checkpoint"}
REF_VAE_FLAGS = {"--vae_checkpoint", "--vae_config_path", "--image_path", "--output_dir", "--resolution"}   # infer_vae.py:83-91


def _flags(parser):
    return {o for a in parser._actions for o in a.option_strings if o.startswith("--") and o != "--help"}


def test_cli_flags_match_reference():
    assert REF_FULL_FLAGS <= _flags(infer_full.build_parser())
    assert _flags(infer_full.build_parser()) - REF_FULL_FLAGS == {"--batch_size", "--device_resize", "--host_resize", "--workers", "--serial", "--fp8", "--fp16_operands"}
    assert _flags(infer_vae.build_parser()) - REF_VAE_FLAGS == {"--batch_size", "--host_resize", "--workers", "--fp16_operands"}
    a = infer_full.build_parser().parse_args(["--vae_checkpoint", "v", "--decoder_checkpoint", "d", "--image_path", "i",
                                              "--tags_csv_path", "t"])
    assert (a.resolution, a.confidence_threshold, a.output_dir, a.use_attention, a.use_cross_attention) == \
        (1024, 0.5, "inference_output", True, False)


def test_summarize_matches_reference_schema():
    conf = [0.98765, 0.7, 0.50004, 0.49, 0.1, 0.05]
    idx = [3, 0, 5, 1, 2, 4]
    tags = [f"t{i}" for i in range(6)]
    r = infer_full.summarize(conf, idx, tags, 0.5)
    assert r["predicted_tags"] == [{"tag": "t3", "confidence": 0.9877}, {"tag": "t0", "confidence": 0.7},
                                   {"tag": "t5", "confidence": 0.5}]
    assert r["total_tags_above_threshold"] == 3 and r["max_confidence"] == 0.9877
    assert r["avg_confidence_top5"] == float(f"{sum(conf[:5]) / 5:.4f}")
    short = infer_full.summarize([0.9, 0.2], [1, 0], tags, 0.5)          # fewer than 5 tags: still divides by 5
    assert short["avg_confidence_top5"] == float(f"{(0.9 + 0.2) / 5:.4f}")


@pytest.mark.gpu
def test_infer_full_and_infer_vae_end_to_end(tmp_path, monkeypatch):
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import decoder_ref, encoder_ref
    from vae_tagger_amd.modules import get_image_transform
    n_tags, res = 40, 128
    g = torch.Generator().manual_seed(5)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    for i, size in enumerate([(200, 150), (128, 128), (90, 160)]):
        arr = (torch.rand(size[1], size[0], 3, generator=g) * 255).to(torch.uint8).numpy()
        Image.fromarray(arr).save(imgs / f"img{i}.png")
    (imgs / "broken.png").write_bytes(b"not a png")                        # skip-and-count
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_e["decoder.conv_in.weight"] = torch.zeros(4)                       # real checkpoints carry decoder.* keys too
    save_file(sd_e, str(tmp_path / "vae.safetensors"))
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1)
    torch.save(sd_d, tmp_path / "dec.pth")
    (tmp_path / "tags.csv").write_text("name\n" + "\n".join(f"tag_{i:05d}" for i in range(n_tags)) + "\n")
    out = tmp_path / "out"
    res_full = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint",
                                str(tmp_path / "dec.pth"), "--image_path", str(imgs), "--tags_csv_path",
                                str(tmp_path / "tags.csv"), "--output_dir", str(out), "--resolution", str(res),
                                "--confidence_threshold", "0.5", "--batch_size", "2"])
    written = json.loads((out / "classification_results.json").read_text())
    assert written == res_full and len(written) == 3
    # the default route resizes + normalises on the GPU (Pillow's arithmetic reproduced exactly) and is pipelined; the reference's own
    # route (PIL transforms on the CPU: --host_resize) and its loop shape (--serial: one batch at a time) write the identical JSON,
    # key order included
    for extra in (["--host_resize"], ["--serial"], ["--host_resize", "--serial", "--workers", "1"], ["--device_resize", "--batch_size", "3"]):
        res_alt = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint",
                                   str(tmp_path / "dec.pth"), "--image_path", str(imgs), "--tags_csv_path",
                                   str(tmp_path / "tags.csv"), "--output_dir", str(tmp_path / "out_alt"), "--resolution", str(res),
                                   "--confidence_threshold", "0.5", "--batch_size", "2"] + extra)
        assert res_alt == res_full, extra
        assert json.dumps(res_alt, sort_keys=True) == json.dumps(written, sort_keys=True)
    # one image whose tensor holds a NaN (a decode that went wrong) in a batch of three: the device leg fails for the batch, is retried
    # image by image and loses exactly that ONE image, like the reference's per-image try/except (infer_full.py:130-132); the two
    # healthy images get the entries of the healthy run (batch composition does not change a bit), and fp16 storage stays on
    real_tf = infer_full.get_image_transform

    def poisoned(resolution, *a, **kw):
        tf = real_tf(resolution, *a, **kw)

        def f(img):
            t = tf(img)
            if img.size == (90, 160):
                t[0, 3, 5] = float("nan")
            return t
        return f
    monkeypatch.setattr(infer_full, "get_image_transform", poisoned)
    res_nan = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint",
                               str(tmp_path / "dec.pth"), "--image_path", str(imgs), "--tags_csv_path",
                               str(tmp_path / "tags.csv"), "--output_dir", str(tmp_path / "out_nan"), "--resolution", str(res),
                               "--confidence_threshold", "0.5", "--batch_size", "4", "--host_resize"])
    bad = [k for k in res_full if k.endswith("img2.png")]
    assert len(bad) == 1 and set(res_nan) == set(res_full) - set(bad)
    assert all(res_nan[k] == res_full[k] for k in res_nan)
    # the same poisoned image under --fp8: a NaN raises BOTH status bits (the e4m3 conversion clamps it and counts a clamp);
    # that is bad input, not a checkpoint that saturates e4m3 -- fp8 mode stays on, nothing is redone in bf16, the healthy images get exactly
    # the healthy fp8 run's entries
    f8_args = ["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint", str(tmp_path / "dec.pth"), "--image_path", str(imgs),
               "--tags_csv_path", str(tmp_path / "tags.csv"), "--resolution", str(res), "--confidence_threshold", "0.5", "--fp8", "--host_resize"]
    nan_f8 = infer_full.main(f8_args + ["--output_dir", str(tmp_path / "out_nan8"), "--batch_size", "4"])
    nan_f8_b1 = infer_full.main(f8_args + ["--output_dir", str(tmp_path / "out_nan8b"), "--batch_size", "1"])
    monkeypatch.setattr(infer_full, "get_image_transform", real_tf)
    healthy_f8 = infer_full.main(f8_args + ["--output_dir", str(tmp_path / "out_h8"), "--batch_size", "4"])
    assert set(nan_f8) == set(res_full) - set(bad) == set(nan_f8_b1)
    assert all(nan_f8[k] == healthy_f8[k] and nan_f8_b1[k] == healthy_f8[k] for k in nan_f8)
    assert any(healthy_f8[k] != res_full[k] for k in nan_f8)                  # (fp8 entries do differ from bf16 ones: the mode really stayed on)
    # opt-in fp8 mode (BASELINE configs[4]): the same schema; confidences within 1e-2 of the default path's
    res_f8 = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint",
                              str(tmp_path / "dec.pth"), "--image_path", str(imgs), "--tags_csv_path",
                              str(tmp_path / "tags.csv"), "--output_dir", str(tmp_path / "out_f8"), "--resolution", str(res),
                              "--confidence_threshold", "0.5", "--batch_size", "2", "--fp8"])
    assert set(res_f8) == set(res_full)
    for k in res_full:
        assert abs(res_f8[k]["max_confidence"] - res_full[k]["max_confidence"]) <= 1e-2
        assert abs(res_f8[k]["avg_confidence_top5"] - res_full[k]["avg_confidence_top5"]) <= 1e-2
    # --fp8 on a checkpoint whose activations exceed the e4m3 range: the status word says so and the CLI redoes the batch in bf16,
    # i.e. writes exactly what the default run of that checkpoint writes
    sd_big = dict(sd_e)
    for k in ("weight", "bias"):
        sd_big[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] = sd_e[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] * 100.0
    save_file(sd_big, str(tmp_path / "vae_big.safetensors"))
    common = ["--decoder_checkpoint", str(tmp_path / "dec.pth"), "--image_path", str(imgs), "--tags_csv_path", str(tmp_path / "tags.csv"),
              "--resolution", str(res), "--confidence_threshold", "0.5", "--batch_size", "2"]
    big_bf16 = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae_big.safetensors"), "--output_dir", str(tmp_path / "out_b0")] + common)
    big_fp8 = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae_big.safetensors"), "--output_dir", str(tmp_path / "out_b8"), "--fp8"] + common)
    assert len(big_bf16) == 3 and big_fp8 == big_bf16
    # saturation that first shows in a LATER batch (the redo of earlier fp8 entries was never run): the status word is forced
    # on the second device batch -- the run must end with ONE numeric mode, i.e. exactly the bf16 run's file
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    real_async, real_status, calls = EncodeTagPipeline.status_async, EncodeTagPipeline.status, {"n": 0, "armed": False}

    def late_saturation(self, out, clear=True):
        real_async(self, out, clear)
        calls["n"] += 1
        if calls["n"] == 2:
            torch.cuda.current_stream(self.device).synchronize()
            out[0] = 2                                                  # VT_STATUS_FP8_SATURATED reported for the second batch ...
            calls["armed"] = True

    def status_once(self, clear=True):
        st = real_status(self, clear)
        if calls["armed"]:
            calls["armed"] = False
            return st | 2                                               # ... and again when the synchronous leg reruns it in fp8 mode
        return st
    monkeypatch.setattr(EncodeTagPipeline, "status_async", late_saturation)
    monkeypatch.setattr(EncodeTagPipeline, "status", status_once)
    late = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--output_dir", str(tmp_path / "out_late"), "--fp8"] + common[:-1] + ["1"])
    monkeypatch.setattr(EncodeTagPipeline, "status_async", real_async)
    monkeypatch.setattr(EncodeTagPipeline, "status", real_status)
    assert calls["n"] >= 2 and late == res_full
    # --fp16_operands on a checkpoint whose first GroupNorm output leaves the fp16 range: the operands overflow, the next norm's statistics are
    # not finite, the health word says so and the CLI settles on bf16 operands + fp32 residual storage -- where the default run of that checkpoint
    # ends up too (its fp16-stored conv1 output overflows as well): the same file
    sd_huge = dict(sd_e)
    for k in ("weight", "bias"):
        sd_huge[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] = sd_e[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] * 1.0e5
    save_file(sd_huge, str(tmp_path / "vae_huge.safetensors"))
    huge_default = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae_huge.safetensors"), "--output_dir", str(tmp_path / "out_h0")] + common)
    huge_f16 = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae_huge.safetensors"), "--output_dir", str(tmp_path / "out_h1"), "--fp16_operands"] + common)
    assert len(huge_default) == 3 and huge_f16 == huge_default
    # the precision mode on the healthy checkpoint: same schema, confidences within 1e-3 of the default path's (logits move by ~3e-4)
    res_f16 = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--output_dir", str(tmp_path / "out_f16"), "--fp16_operands"] + common)
    assert set(res_f16) == set(res_full)
    for k in res_full:
        assert abs(res_f16[k]["max_confidence"] - res_full[k]["max_confidence"]) <= 1e-3
    lat = infer_vae.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--image_path", str(imgs),
                          "--output_dir", str(out), "--resolution", str(res)])
    assert len(lat) == 3 and all(len(v) == 16 * (res // 8) ** 2 for v in lat.values())
    lat_host = infer_vae.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--image_path", str(imgs),
                               "--output_dir", str(tmp_path / "out_lh"), "--resolution", str(res), "--host_resize", "--batch_size", "2"])
    assert lat_host == lat
    # against the CPU oracle on the same preprocessed pixels
    tf = get_image_transform(res)
    sd_e.pop("decoder.conv_in.weight")
    for path, entry in written.items():
        x = tf(Image.open(path).convert("RGB")).unsqueeze(0)
        ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x)
        assert (torch.tensor(lat[path]) - ref_lat.reshape(-1)).abs().max() <= 1e-2
        conf, idx = decoder_ref.get_confidence(decoder_ref.attention_decoder_forward(sd_d, ref_lat))
        want = infer_full.summarize(conf[0].tolist(), idx[0].tolist(), [f"tag_{i:05d}" for i in range(n_tags)], 0.5)
        assert abs(entry["max_confidence"] - want["max_confidence"]) <= 1e-2
        assert abs(entry["avg_confidence_top5"] - want["avg_confidence_top5"]) <= 1e-2
        got_tags = {t["tag"] for t in entry["predicted_tags"]}
        sure = {t["tag"] for t in want["predicted_tags"] if t["confidence"] > 0.51}
        maybe = {t["tag"] for t in want["predicted_tags"]} | {f"tag_{int(i):05d}" for c, i in zip(conf[0], idx[0]) if c > 0.49}
        assert sure <= got_tags <= maybe


@pytest.mark.gpu
def test_cli_at_configs0_shape_single_512_image(tmp_path):
    """BASELINE.json configs[0]: infer_full.py on a single 512x512 image, FLUX VAE + 8-head attention decoder (the reference's own
    CPU-runnable case, infer_full.py:73-141), and infer_vae.py on the same file: latents, confidences and the thresholded tag set
    against the CPU oracle on the same preprocessed pixels -- what the `encoder_enc_512x512` golden pins for synthetic floats,
    here through the file -> PIL -> transform -> encode -> decode -> JSON route."""
    from PIL import Image
    from safetensors.torch import save_file
    from oracle import decoder_ref, encoder_ref
    from vae_tagger_amd.modules import get_image_transform
    n_tags, res = 1000, 512
    g = torch.Generator().manual_seed(11)
    img_path = tmp_path / "one.png"
    low = torch.rand(16, 16, 3, generator=g)                     # a smooth picture plus noise: not a flat field, not white noise
    arr = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(res, res), mode="bicubic", align_corners=False)[0].permute(1, 2, 0)
    arr = ((arr + 0.08 * torch.randn(res, res, 3, generator=g)).clamp(0, 1) * 255).to(torch.uint8).numpy()
    Image.fromarray(arr).save(img_path)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    save_file(sd_e, str(tmp_path / "vae.safetensors"))
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1)
    torch.save(sd_d, tmp_path / "dec.pth")
    names = [f"tag_{i:05d}" for i in range(n_tags)]
    (tmp_path / "tags.csv").write_text("name\n" + "\n".join(names) + "\n")
    full = infer_full.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--decoder_checkpoint", str(tmp_path / "dec.pth"),
                            "--image_path", str(img_path), "--tags_csv_path", str(tmp_path / "tags.csv"), "--output_dir", str(tmp_path / "o"),
                            "--resolution", str(res)])
    lat = infer_vae.main(["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--image_path", str(img_path), "--output_dir", str(tmp_path / "o"),
                          "--resolution", str(res)])
    assert list(full) == [str(img_path)] == list(lat) and len(lat[str(img_path)]) == 16 * 64 * 64
    x = get_image_transform(res)(Image.open(img_path).convert("RGB")).unsqueeze(0)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x)
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    dl = (torch.tensor(lat[str(img_path)]) - ref_lat.reshape(-1)).abs().max().item()
    conf, idx = decoder_ref.get_confidence(ref_logits)
    want = infer_full.summarize(conf[0].tolist(), idx[0].tolist(), names, 0.5)
    entry = full[str(img_path)]
    # A SMOOTH picture is harder for 8-bit significands than the uniform noise of the synthetic batches: where the picture is flat the
    # rounding errors of neighbouring pixels are equal, so a 3x3 conv adds them coherently instead of averaging them.  The oracle with its
    # operands rounded to bf16 (encoder_ref's emulate_bf16: the arithmetic north_star prescribes) is itself 1.5e-2 from the fp32 oracle on
    # this image (7e-3 on noise of the same size; rms 1.7e-3 vs 1.6e-3) -- the bar for the latent MAXIMUM here is that emulation, not 1e-2;
    # rms, confidences and the tag set keep their bounds.  (fp16 operands would give 1.5e-3: tests/diagnostics/smooth_image_study.py.)
    emu = encoder_ref.vae_wrapper_encode(sd_e, x, emulate_bf16=True)
    d_emu = (emu - ref_lat).abs().max().item()
    rms = (torch.tensor(lat[str(img_path)]) - ref_lat.reshape(-1)).pow(2).mean().sqrt().item()
    print(f"configs[0] shape through the CLIs: max|dlatent| {dl:.3e} (bf16-operand emulation of the oracle: {d_emu:.3e}), rms {rms:.3e}; "
          f"tags >= 0.5: oracle {want['total_tags_above_threshold']}, HIP {entry['total_tags_above_threshold']}")
    assert dl <= max(1e-2, 1.5 * d_emu) and rms <= 3e-3
    assert abs(entry["max_confidence"] - want["max_confidence"]) <= 2.5e-3        # |dsigmoid| <= |dlogit| / 4
    assert abs(entry["avg_confidence_top5"] - want["avg_confidence_top5"]) <= 2.5e-3
    got_tags = {t["tag"] for t in entry["predicted_tags"]}
    sure = {names[int(i)] for lg, i in zip(ref_logits[0, idx[0]], idx[0]) if lg > 1e-2}
    maybe = {names[int(i)] for lg, i in zip(ref_logits[0, idx[0]], idx[0]) if lg >= -1e-2}
    assert sure <= got_tags <= maybe
    # the written order is the sort order: confidences descending
    cs = [t["confidence"] for t in entry["predicted_tags"]]
    assert cs == sorted(cs, reverse=True)


@pytest.mark.gpu
def test_sharded_cli_two_ranks_rehearsal(tmp_path):
    """`torchrun --nproc-per-node 2 -m vae_tagger_amd.infer_full / infer_vae`: the image list is split with sharding.shard_range, each rank runs its
    share through the pipelined loop and rank 0 writes the merged file -- the same entries as the one-process run.  Both ranks share GPU 0 here
    and the object gather runs on gloo (VT_CLI_GLOO=1; RCCL needs one GPU per rank); the process group is created before any GPU call."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_tags, res = 40, 96
    g = torch.Generator().manual_seed(7)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    for i in range(7):
        arr = (torch.rand(64 + 8 * i, 120 - 4 * i, 3, generator=g) * 255).to(torch.uint8).numpy()
        Image.fromarray(arr).save(imgs / f"img{i}.{'jpg' if i % 2 else 'png'}")
    (imgs / "broken.png").write_bytes(b"not a png")
    save_file(synth.synth_state_dict(synth.encoder_manifest(), seed=0), str(tmp_path / "vae.safetensors"))
    torch.save(synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1), tmp_path / "dec.pth")
    (tmp_path / "tags.csv").write_text("name\n" + "\n".join(f"tag_{i:05d}" for i in range(n_tags)) + "\n")
    common = ["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--image_path", str(imgs), "--resolution", str(res), "--batch_size", "2"]
    full = common + ["--decoder_checkpoint", str(tmp_path / "dec.pth"), "--tags_csv_path", str(tmp_path / "tags.csv")]
    one = infer_full.main(full + ["--output_dir", str(tmp_path / "one")])
    one_lat = infer_vae.main(common + ["--output_dir", str(tmp_path / "one")])
    assert len(one) == 7 == len(one_lat)
    env = dict(os.environ, VT_CLI_GLOO="1", PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    for mod, args, name, want in (("vae_tagger_amd.infer_full", full, "classification_results.json", one),
                                  ("vae_tagger_amd.infer_vae", common, "latent_vectors.json", one_lat)):
        port = str(29700 + os.getpid() % 200)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, "-m", mod] + args + ["--output_dir", str(tmp_path / "two")]
        r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = json.loads((tmp_path / "two" / name).read_text())
        assert got == want
        assert "失败: 1" in r.stdout.decode()                         # the broken file is counted once, on the rank that owns it


@pytest.mark.gpu
def test_sharded_cli_collectives_run_on_rccl_with_one_rank(tmp_path):
    """The sharded CLI path's three exchanges -- image-list broadcast, the fp8-abandoned all-reduce, the result gather -- on the REAL backend:
    VT_CLI_ONE_RANK_GROUP=1 makes `infer_full` / `infer_vae` create a one-rank "nccl" (= RCCL) group before any GPU call and go through the
    same collective calls a multi-GPU torchrun launch makes (this box has one GPU; a two-rank run has to use gloo).  Same JSON as the plain run.
    It proves the calls execute on RCCL with CUDA-side object collectives; it says nothing about scaling."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_tags, res = 40, 96
    g = torch.Generator().manual_seed(9)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    for i in range(5):
        Image.fromarray((torch.rand(80 + 4 * i, 100, 3, generator=g) * 255).to(torch.uint8).numpy()).save(imgs / f"img{i}.png")
    save_file(synth.synth_state_dict(synth.encoder_manifest(), seed=0), str(tmp_path / "vae.safetensors"))
    torch.save(synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1), tmp_path / "dec.pth")
    (tmp_path / "tags.csv").write_text("name\n" + "\n".join(f"tag_{i:05d}" for i in range(n_tags)) + "\n")
    common = ["--vae_checkpoint", str(tmp_path / "vae.safetensors"), "--image_path", str(imgs), "--resolution", str(res), "--batch_size", "2"]
    full = common + ["--decoder_checkpoint", str(tmp_path / "dec.pth"), "--tags_csv_path", str(tmp_path / "tags.csv"), "--fp8"]
    one = infer_full.main(full + ["--output_dir", str(tmp_path / "one")])
    one_lat = infer_vae.main(common + ["--output_dir", str(tmp_path / "one")])
    env = dict(os.environ, VT_CLI_ONE_RANK_GROUP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29990 - os.getpid() % 40), PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    for mod, args, name, want in (("vae_tagger_amd.infer_full", full, "classification_results.json", one),
                                  ("vae_tagger_amd.infer_vae", common, "latent_vectors.json", one_lat)):
        probe = ("import sys, runpy, torch.distributed as dist\n"
                 f"sys.argv = [{mod!r}] + {args + ['--output_dir', str(tmp_path / 'grp')]!r}\n"
                 f"runpy.run_module({mod!r}, run_name='__main__')\n"
                 "assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
                 "print('GROUP_OK', dist.get_backend())\n")
        r = subprocess.run([sys.executable, "-c", probe], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
        assert r.returncode == 0 and b"GROUP_OK nccl" in r.stdout, (r.stdout.decode()[-800:], r.stderr.decode()[-2000:])
        assert json.loads((tmp_path / "grp" / name).read_text()) == want
