"""Which input shapes make the encoder non-deterministic?   python tests/diagnostics/nondeterminism_shapes.py [reps] [bf16|fp8|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
if mode == "fp8": vae._context().call("vt_set_flag", 11, 1)
if mode == "f16": vae._context().call("vt_set_flag", 18, 1)
print(f"mode {mode}", flush=True)
for (B, h, w) in ((2, 264, 136), (1, 264, 136), (2, 264, 128), (2, 256, 136), (2, 136, 264), (2, 200, 104), (2, 520, 264), (2, 248, 120), (4, 264, 136), (2, 72, 88), (2, 100, 76), (2, 328, 200)):
    x = synth.synth_images(B, h, w, seed=h + 2 * w).cuda()
    ref = vm.encode(x).clone(); bad = 0; worst = 0.0; imgs = set()
    for rep in range(reps):
        o = vm.encode(x)
        if not torch.equal(o, ref):
            bad += 1; worst = max(worst, (o - ref).abs().max().item())
            imgs |= set(torch.nonzero((o != ref).flatten(1).any(1)).flatten().tolist())
    print(f"B{B} {w}x{h} (latent {w // 8}x{h // 8}): {bad}/{reps} repeats differ (max |d| {worst:.2e}) images {sorted(imgs)}", flush=True)
