"""Determinism soak of the whole encode+tag step at the bench shape: N repetitions must give bit-identical logits and
latents (a timing race in an LDS ring shows up as rare differing patches).  python tools/soak.py [reps] [batch] [flag=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
vae_model = DiffusersVAEWrapper(vae).to("cuda").eval()
dec = create_attention_decoder(16, 128, 128, 1000, {"use_spatial_attention": True, "use_self_attention": True})
dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(1000), seed=1), strict=False)
pipe = EncodeTagPipeline(vae_model, dec.to("cuda").eval())
pipe.check_finite = False
for fv in sys.argv[3:]:
    f, v = fv.split("=")
    pipe.ctx.call("vt_set_flag", int(f), int(v))
x = synth.synth_images(B, 1024, 1024, seed=7).cuda()
ref_logits, ref_lat = pipe.logits(x, return_latent=True)
ref_logits, ref_lat = ref_logits.clone(), ref_lat.clone()
bad = 0
for i in range(reps):
    lg, lat = pipe.logits(x, return_latent=True)
    if not (torch.equal(lg, ref_logits) and torch.equal(lat, ref_lat)):
        bad += 1
        d = (lat != ref_lat).nonzero()
        print(f"rep {i}: {d.shape[0]} latent elements differ, first {d[:3].tolist()}", flush=True)
    if (i + 1) % 25 == 0:
        print(f"{i + 1}/{reps} repetitions, {bad} differing", flush=True)
print("soak:", "CLEAN" if bad == 0 else f"{bad} DIFFERING REPETITIONS", flush=True)
sys.exit(0 if bad == 0 else 1)
