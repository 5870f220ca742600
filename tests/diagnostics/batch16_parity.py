"""Every image of the configs[2] batch (16 x 1024^2, 10 000 tags) against the CPU oracle -- the driver-run test compares two of them.
   python tests/diagnostics/batch16_parity.py [--fp8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import decoder_ref, encoder_ref
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
fp8 = "--fp8" in sys.argv
n = 10000
sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
vae = load_diffusers_vae_from_config(get_diffusers_vae_config()); vae.load_state_dict(sd_e, strict=False)
dec = create_attention_decoder(16, 128, 128, n, {"use_spatial_attention": True, "use_self_attention": True})
dec.load_state_dict(sd_d, strict=False)
pipe = EncodeTagPipeline(DiffusersVAEWrapper(vae).to("cuda").eval(), dec.to("cuda").eval())
if fp8:
    pipe.ctx.call("vt_set_flag", 11, 1)
x = synth.synth_images(16, 1024, 1024, seed=1000)            # the bench batch of rank 0
lg, lat = pipe.logits(x.cuda(), return_latent=True)
lg, lat = lg.cpu(), lat.cpu()
worst = [0.0, 0.0]
t0 = time.time()
for i in range(16):
    rl = encoder_ref.vae_wrapper_encode(sd_e, x[i:i + 1])
    rg = decoder_ref.attention_decoder_forward(sd_d, rl)
    dl, dg = (lat[i:i + 1] - rl), (lg[i:i + 1] - rg)
    worst = [max(worst[0], dl.abs().max().item()), max(worst[1], dg.abs().max().item())]
    print(f"image {i:2d}: max|dlatent| {dl.abs().max():.3e} rms {dl.pow(2).mean().sqrt():.3e}  max|dlogit| {dg.abs().max():.3e}  ({time.time() - t0:.0f} s)", flush=True)
print(f"batch of 16 x 1024^2 ({'fp8' if fp8 else 'bf16'}): worst max|dlatent| {worst[0]:.3e}, worst max|dlogit| {worst[1]:.3e}", flush=True)
