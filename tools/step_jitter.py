"""Per-step wall times of the full path (batch 16 x 1024^2): looks for outlier steps.  python tools/step_jitter.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    dec = create_attention_decoder(16, 128, 128, 10000, {"use_spatial_attention": True, "use_self_attention": True})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(10000), seed=1), strict=False)
    pipe = EncodeTagPipeline(DiffusersVAEWrapper(vae).to("cuda").eval(), dec.to("cuda").eval())
x = synth.synth_images(16, 1024, 1024, seed=0).cuda()
for _ in range(2): pipe.logits(x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    pipe.logits(x)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
print("per-step ms:", " ".join(f"{m:.1f}" for m in ms))
print(f"min {min(ms):.2f}  median {sorted(ms)[len(ms)//2]:.2f}  max {max(ms):.2f}")
