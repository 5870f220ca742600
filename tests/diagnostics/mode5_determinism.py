"""Is the encoder deterministic with the attention's linear layers on the Q.K^T skeleton (flag 17), at a ragged token count?
   python tests/diagnostics/mode5_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
ctx = vae._context()
for (B, h, w) in ((2, 264, 136), (2, 72, 88), (2, 128, 192), (1, 512, 512)):
    x = synth.synth_images(B, h, w, seed=h + 2 * w).cuda()
    for f17 in (1, 0):
        ctx.call("vt_set_flag", 17, f17)
        for what, fn in (("moments", lambda: vae.encode(x).latent_dist.parameters.clone()), ("scaled mode", lambda: vm.encode(x).clone())):
            ref = fn(); bad = 0; worst = 0.0
            for rep in range(30):
                o = fn()
                if not torch.equal(o, ref):
                    bad += 1; worst = max(worst, (o - ref).abs().max().item())
            print(f"B{B} {w}x{h} flag17={f17} {what}: {bad}/30 repeats differ (max |d| {worst:.2e})", flush=True)
ctx.call("vt_set_flag", 17, 1)
