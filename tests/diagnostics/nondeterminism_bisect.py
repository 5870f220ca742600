"""Which kernel makes the encoder non-deterministic at 136 x 264 (latent 17 x 33)?  Repeats per flag setting.
   python tests/diagnostics/nondeterminism_bisect.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
ctx = vae._context()
DEFAULTS = {0: 1, 1: 1, 8: 1, 9: 1, 12: 1, 13: 1, 17: 1, 19: 1, 20: 1, 7: 0, 3: 3, 5: 1, 4: 1}
x = synth.synth_images(2, 264, 136, seed=264 + 2 * 136).cuda()
for name, flags in (("default", {}), ("flag 19=0 (NHWC stride-2 input)", {19: 0}), ("flag 20=0 (conv_out generic)", {20: 0}), ("flag 13=0 (stride-2 generic)", {13: 0}),
                    ("flag 17=0", {17: 0}), ("flag 9=0,12=0 (attention on generic GEMMs)", {9: 0, 12: 0}), ("flag 7=1 (exact row max)", {7: 1}),
                    ("flag 12=0 (P.V generic)", {12: 0}), ("flag 1=0 (no epilogue GN stats)", {1: 0}), ("flag 8=0 (shortcut unfused)", {8: 0}),
                    ("flag 0=0 (generic 3x3 convs)", {0: 0}), ("flag 3=0 (8-wave halo tiles)", {3: 0}), ("flag 5=0 (fp32 conv_in)", {5: 0}),
                    ("flag 4=0 (fp32 residual)", {4: 0})):
    for f, v in DEFAULTS.items(): ctx.call("vt_set_flag", f, v)
    for f, v in flags.items(): ctx.call("vt_set_flag", f, v)
    ref = vm.encode(x).clone(); bad = 0; worst = 0.0
    for rep in range(reps):
        o = vm.encode(x)
        if not torch.equal(o, ref):
            bad += 1; worst = max(worst, (o - ref).abs().max().item())
    print(f"{name:48s}: {bad}/{reps} repeats differ (max |d| {worst:.2e})", flush=True)
for f, v in DEFAULTS.items(): ctx.call("vt_set_flag", f, v)
