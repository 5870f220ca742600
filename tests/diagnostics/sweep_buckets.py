"""Parity sweep over every reachable aspect-ratio bucket (reference AspectRatioBucketing(512, 1024, 64), modules.py:180-222):
one image per bucket through the HIP encoder + decoder against the CPU oracle.  python tests/diagnostics/sweep_buckets.py [stride]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import decoder_ref, encoder_ref
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import AspectRatioBucketing, create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(100), seed=1)
vae = load_diffusers_vae_from_config(get_diffusers_vae_config()); vae.load_state_dict(sd_e, strict=False)
dec = create_attention_decoder(16, 128, 128, 100, {"use_spatial_attention": True, "use_self_attention": True})
dec.load_state_dict(sd_d, strict=False)
pipe = EncodeTagPipeline(DiffusersVAEWrapper(vae).to("cuda").eval(), dec.to("cuda").eval())
bk = AspectRatioBucketing(512, 1024, 64)
reach = sorted({bk.bucket_for_ratio(w / h) for w in range(256, 2049, 8) for h in range(256, 2049, 8)})[::stride]
worst = (0.0, 0.0, None)
t0 = time.time()
for i, (w, h) in enumerate(reach):
    x = synth.synth_images(1, h, w, seed=w * 4096 + h)
    lg, lat = pipe.logits(x.cuda(), return_latent=True)
    rl = encoder_ref.vae_wrapper_encode(sd_e, x)
    rg = decoder_ref.attention_decoder_forward(sd_d, rl)
    dl, dg = (lat.cpu() - rl).abs().max().item(), (lg.cpu() - rg).abs().max().item()
    if dl > worst[0]: worst = (dl, dg, (w, h))
    print(f"[{i + 1}/{len(reach)}] {w}x{h}: max|dlatent| {dl:.2e}  max|dlogit| {dg:.2e}  ({time.time() - t0:.0f} s)", flush=True)
    assert dl <= 1e-2 and dg <= 1e-2, (w, h)
print(f"sweep: {len(reach)} buckets within 1e-2; worst latent error {worst[0]:.2e} at {worst[2]}", flush=True)
