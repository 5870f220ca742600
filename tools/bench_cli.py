"""End-to-end images/s of the CLI entry point: files on disk -> decode -> upload -> resize / normalise ->
encode + tag -> JSON, against the HBM-resident number bench.py reports for the same batch on the same box.

    python tools/bench_cli.py [--n 512] [--batch 16] [--workers 16] [--fp8] [--tags 10000] [--host_resize] [--serial]

N synthetic 1024 x 1024 pictures (smooth fields + noise; 32 distinct ones, repeated under different names) are written to a temporary
directory once as PNG and once as JPEG (quality 90); vae_tagger_amd.infer_full.main() runs over each directory (a short warm-up run first:
code objects, allocator, clocks), and the loop's own wall time gives images/s.  Then the same pipeline object shape runs 10 HBM-resident
steps (what bench.py times) for the comparison."""
import argparse, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import numpy as np
import torch
from PIL import Image
from safetensors.torch import save_file
from vae_tagger_amd import infer_full, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--workers", type=int, default=0)
ap.add_argument("--tags", type=int, default=10000)
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--confidence_threshold", type=float, default=0.75,
                help="synthetic-init logits are centred at 0, so the reference's 0.5 puts HALF of the 10 000 tags into every JSON entry; 0.75 leaves "
                     "a few dozen per image, what a trained tagger emits (0.5: the run measures Python dict construction)")
ap.add_argument("--host_resize", action="store_true")
ap.add_argument("--serial", action="store_true")
ap.add_argument("--src_res", type=int, default=0, help="size of the files' pictures (default: --res, i.e. no resize needed; e.g. 1536 exercises the device resize)")
a = ap.parse_args()

tmp = tempfile.mkdtemp(prefix="vt_bench_cli_")
try:
    rng = np.random.default_rng(0)
    src = a.src_res or a.res
    t0 = time.perf_counter()
    base = []
    for i in range(32):
        low = rng.random((12, 12, 3)).astype(np.float32)
        img = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((src, src), Image.BICUBIC), dtype=np.float32) / 255.0
        img = np.clip(img + 0.04 * rng.standard_normal((src, src, 3)).astype(np.float32), 0, 1)
        base.append(Image.fromarray((img * 255).astype(np.uint8)))
    dirs = {}
    for fmt, kw in (("png", {}), ("jpg", {"quality": 90})):
        d = os.path.join(tmp, fmt); os.makedirs(d)
        for i, im in enumerate(base):
            im.save(os.path.join(d, f"base{i:03d}.{fmt}"), **kw)
        for k in range(32, a.n):
            shutil.copyfile(os.path.join(d, f"base{k % 32:03d}.{fmt}"), os.path.join(d, f"img{k:05d}.{fmt}"))
        dirs[fmt] = d
    mb = {f: sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d)) / len(os.listdir(d)) / 1e6 for f, d in dirs.items()}
    print(f"wrote 2 x {a.n} files of {src}x{src} in {time.perf_counter() - t0:.1f} s (mean size: png {mb['png']:.2f} MB, jpg {mb['jpg']:.2f} MB)", flush=True)
    save_file(synth.synth_state_dict(synth.encoder_manifest(), seed=0), os.path.join(tmp, "vae.safetensors"))
    torch.save(synth.synth_state_dict(synth.attention_decoder_manifest(a.tags), seed=1), os.path.join(tmp, "dec.pth"))
    with open(os.path.join(tmp, "tags.csv"), "w") as f:
        f.write("name\n" + "\n".join(f"tag_{i:05d}" for i in range(a.tags)) + "\n")
    common = ["--vae_checkpoint", os.path.join(tmp, "vae.safetensors"), "--decoder_checkpoint", os.path.join(tmp, "dec.pth"), "--tags_csv_path",
              os.path.join(tmp, "tags.csv"), "--resolution", str(a.res), "--batch_size", str(a.batch), "--workers", str(a.workers),
              "--confidence_threshold", str(a.confidence_threshold)]
    common += (["--fp8"] if a.fp8 else []) + (["--host_resize"] if a.host_resize else []) + (["--serial"] if a.serial else [])
    rows = {}
    # warm-up: one small run (code objects, allocator pools, pinned staging)
    warm = os.path.join(tmp, "warm"); os.makedirs(warm)
    for i in range(2 * a.batch):
        shutil.copyfile(os.path.join(dirs["jpg"], f"base{i % 32:03d}.jpg"), os.path.join(warm, f"w{i:03d}.jpg"))
    with contextlib.redirect_stdout(sys.stderr):
        infer_full.main(common + ["--image_path", warm, "--output_dir", os.path.join(tmp, "out_warm")])
    for fmt in ("jpg", "png"):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            res = infer_full.main(common + ["--image_path", dirs[fmt], "--output_dir", os.path.join(tmp, "out_" + fmt)])
        wall = time.perf_counter() - t0
        st = dict(infer_full.LAST_RUN_STATS)
        ntag = sum(e["total_tags_above_threshold"] for e in res.values()) / max(1, len(res))
        rows[fmt] = {"images": len(res), "tags_per_image": round(ntag, 1), "loop_seconds": round(st["loop_seconds"], 3), "images_per_sec_loop": round(len(res) / st["loop_seconds"], 1),
                     "whole_call_seconds": round(wall, 2), "mean_file_mb": round(mb[fmt], 2)}
        print(f"{fmt}: {rows[fmt]}", flush=True)
    # the HBM-resident number on this box: what bench.py times
    from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
    from vae_tagger_amd.modules import create_attention_decoder
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    with contextlib.redirect_stdout(sys.stderr):
        vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
        vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
        vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
        dec = create_attention_decoder(16, a.res // 8, a.res // 8, a.tags, {"use_spatial_attention": True, "use_self_attention": True})
        dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(a.tags), seed=1), strict=False)
        pipe = EncodeTagPipeline(vm, dec.to("cuda").eval())
        pipe.check_finite = False
    if a.fp8: pipe.set_fp8(True)
    x = synth.synth_images(a.batch, a.res, a.res, seed=1000).cuda()
    for _ in range(3): pipe.logits(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pipe.logits(x)
    torch.cuda.synchronize()
    resident = a.batch * 10 / (time.perf_counter() - t0)
    out = {"workload": f"{a.n} files {src}x{src} -> --resolution {a.res}, batch {a.batch}, {a.tags} tags, threshold {a.confidence_threshold}, {'fp8' if a.fp8 else 'bf16'}"
                       f"{', host resize' if a.host_resize else ''}{', serial' if a.serial else ''}", "workers": a.workers or "min(16, cores)",
           "cores_available": len(os.sched_getaffinity(0)), "hbm_resident_images_per_sec": round(resident, 1), "cli": rows,
           "cli_fraction_of_resident": {f: round(r["images_per_sec_loop"] / resident, 3) for f, r in rows.items()}}
    print(json.dumps(out), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
