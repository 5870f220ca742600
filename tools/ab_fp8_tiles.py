"""Round 4, VERDICT item 3: tile shapes of the fp8 halo conv on the layers where tile ends outweigh the main loop (128 -> 128 @1024^2:
18 K-steps per tile).  Interleaved A/B on ONE box of vt_set_flag(ctx, 16, v) inside the real fp8 step (batch 16 x 1024^2, 10 000 tags):
  v = 0  8 x 32 px x 128 couts, 4 waves, two workgroups per CU (default)      v = 1 / 5  16 x 32 px, 8 waves, one per CU (Cin <= 128 / every layer)
                                                                               v = 2 / 6  8 x 64 px, 8 waves, one per CU
Per variant: ms per step of the fp8 halo launches with Cin <= 128 (four per step) and of the other sixteen (library hipEvents, slots 19 / 11),
and the whole step.   python tools/ab_fp8_tiles.py [reps] [steps per measurement]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, W, N = 16, 1024, 1024, 10000
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
    dec = create_attention_decoder(16, H // 8, W // 8, N, {"use_spatial_attention": True, "use_self_attention": True})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(N), seed=1), strict=False)
    pipe = EncodeTagPipeline(vm, dec.to("cuda").eval())
    pipe.check_finite = False
pipe.set_fp8(True)
x = synth.synth_images(B, H, W, seed=1000).cuda()
ctx = pipe.ctx
ns = ctx.lib.vt_profile_num_configs()
la = (ctypes.c_longlong * ns)(); ms = (ctypes.c_double * ns)(); fl = (ctypes.c_double * ns)(); nm = (ctypes.c_char_p * ns)()
ref = None
rows = {}
for _ in range(3): pipe.logits(x)
for rep in range(reps):
    for v in (0, 1, 2, 5, 6):
        ctx.call("vt_set_flag", 16, v)
        out = pipe.logits(x); pipe.logits(x)
        torch.cuda.synchronize()
        ctx.call("vt_profile_begin")
        t0 = time.perf_counter()
        for _ in range(K): out = pipe.logits(x)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / K * 1e3
        ctx.call("vt_profile_end", ns, la, ms, fl, nm)
        c128 = ms[19] / K; rest = ms[11] / K
        tf128 = fl[19] / max(ms[19], 1e-9) / 1e9; tfr = fl[11] / max(ms[11], 1e-9) / 1e9
        if ref is None: ref = out.clone()
        d = (out - ref).abs().max().item()
        rows.setdefault(v, []).append((c128, rest, step))
        print(f"rep {rep} flag16={v}: Cin<=128 layers {c128:7.3f} ms/step ({la[19] // K} launches, {tf128:6.0f} TF/s)  other fp8 halo layers {rest:7.3f} ms/step "
              f"({la[11] // K} launches, {tfr:6.0f} TF/s)  step {step:7.3f} ms  max|dlogit vs flag16=0| {d:.2e}", flush=True)
ctx.call("vt_set_flag", 16, 0)
assert pipe.status() == 0
print("\nmedians over", reps, "interleaved repetitions:")
med = lambda v: sorted(v)[len(v) // 2]
for v, r in rows.items():
    print(f"  flag16={v}: Cin<=128 layers {med([a for a, _, _ in r]):7.3f} ms  others {med([b for _, b, _ in r]):7.3f} ms  step {med([c for _, _, c in r]):7.3f} ms "
          f"= {B / med([c for _, _, c in r]) * 1e3:6.1f} images/s")
