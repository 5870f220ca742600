"""Which images / shapes differ between fp8 tile shapes (vt_set_flag 16)?  python tests/diagnostics/fp8_tile_shape_diff.py"""
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
ctx.call("vt_set_flag", 11, 1)
for (B, hh, ww) in ((2, 100, 148), (2, 128, 192), (3, 64, 64), (2, 256, 256)):
    x = synth.synth_images(B, hh, ww, seed=hh + 1).cuda()
    x[-1] = x[0]                                  # the last image repeats the first
    ctx.call("vt_set_flag", 16, 0)
    base = vm.encode(x); base2 = vm.encode(x)
    print(f"{B}x{hh}x{ww}: flag16=0 twice identical: {torch.equal(base, base2)}; image 0 == its copy at index {B - 1}: {torch.equal(base[0], base[-1])}")
    for v in (1, 2, 5, 6):
        ctx.call("vt_set_flag", 16, v)
        lat = vm.encode(x); lat2 = vm.encode(x)
        d = [(lat[i] - base[i]).abs().max().item() for i in range(B)]
        print(f"   flag16={v}: deterministic {torch.equal(lat, lat2)}; image 0 == copy: {torch.equal(lat[0], lat[-1])}; max|dlatent| vs flag16=0 per image: " + " ".join(f"{e:.2e}" for e in d))
ctx.call("vt_set_flag", 16, 0)
print("status", ctx.status())
