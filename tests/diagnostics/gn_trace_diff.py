"""Which GroupNorm's statistics differ between two runs of the same input?  vt_debug_trace records, per GroupNorm in launch order, a checksum of
the partials it consumed and of its (scale, shift) table.   python tests/diagnostics/gn_trace_diff.py [reps] [B H W]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib, torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 264, 136)
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to("cuda").eval(); vm.check_finite = False
ctx = vae._context()
x = synth.synth_images(B, h, w, seed=h + 2 * w).cuda()
# record order: [partials, table] per GroupNorm, and one record per stride-1 halo conv output right after it runs
SEQ = []
for blk in ("d0.r0", "d0.r1", "d1.r0", "d1.r1", "d2.r0", "d2.r1", "d3.r0", "d3.r1", "mid.r0"):
    SEQ += [f"partials consumed by {blk}.n1", f"scale/shift of {blk}.n1", f"OUTPUT of {blk}.conv1", f"partials consumed by {blk}.n2", f"scale/shift of {blk}.n2", f"OUTPUT of {blk}.conv2"]
SEQ += ["partials consumed by attn.gn", "scale/shift of attn.gn"]
SEQ += ["partials consumed by mid.r1.n1", "scale/shift of mid.r1.n1", "OUTPUT of mid.r1.conv1", "partials consumed by mid.r1.n2", "scale/shift of mid.r1.n2", "OUTPUT of mid.r1.conv2",
        "partials consumed by norm_out", "scale/shift of norm_out"]
NAMES = ["d0.r0.n1", "d0.r0.n2", "d0.r1.n1", "d0.r1.n2", "d1.r0.n1", "d1.r0.n2", "d1.r1.n1", "d1.r1.n2", "d2.r0.n1", "d2.r0.n2", "d2.r1.n1", "d2.r1.n2",
         "d3.r0.n1", "d3.r0.n2", "d3.r1.n1", "d3.r1.n2", "mid.r0.n1", "mid.r0.n2", "attn.gn", "mid.r1.n1", "mid.r1.n2", "norm_out"]
def trace():
    ctx.call("vt_debug_trace", 1, None, 0, None)
    out = vm.encode(x).clone()
    buf = (ctypes.c_ulonglong * 256)(); n = ctypes.c_int(0)
    ctx.call("vt_debug_trace", 0, buf, 256, ctypes.byref(n))
    return out, list(buf[: n.value])
ref_o, ref_t = trace()
print(f"B{B} {w}x{h}: {len(ref_t)} records traced (expected {len(SEQ)})")
first = {}
for rep in range(reps):
    o, t = trace()
    if t != ref_t:
        k = next(i for i in range(len(t)) if t[i] != ref_t[i])
        what = SEQ[k] if k < len(SEQ) else f"record {k}"
        first[what] = first.get(what, 0) + 1
print("first differing record over", reps, "repeats:", first or "none (deterministic)")
