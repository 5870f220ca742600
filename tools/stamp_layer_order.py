"""Do the FIRST 128-channel 1024^2 convs of a step run at a lower clock than the later, identical ones?  (per-dispatch traces show the first two
3x3 convs of every step 15-18 % slower than the third and fourth, which have the same shapes, types and buffers.)
Needs the diagnostic build:  tools/build_variant.sh stamp -DHALO_STAMP;   python tools/stamp_layer_order.py vae_tagger_amd/csrc/exp/libvt_stamp.so
Stamps of the nth matching launch of a bf16 step: main-loop time and in-kernel clock (d s_memtime / d s_memrealtime), median over workgroups."""
import ctypes, os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vm = DiffusersVAEWrapper(vae).to(dev).eval()
    dec = create_attention_decoder(16, 128, 128, 1000, {"use_spatial_attention": True, "use_self_attention": True})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(1000), seed=1), strict=False)
    pipe = EncodeTagPipeline(vm, dec.to(dev).eval())
    pipe.check_finite = False
L = pipe.ctx.lib
L.vt_debug_halo_stamps_nth.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]; L.vt_debug_halo_stamps_nth.restype = ctypes.c_int
x = synth.synth_images(16, 1024, 1024, seed=1000).to(dev)
st = torch.zeros(1 << 17, 16, dtype=torch.int64, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    pipe.logits(x); torch.cuda.synchronize()
for (H, Cin, n, label) in ((1024, 128, 4, "128->128 @1024^2"), (512, 256, 3, "256->256 @512^2"), (256, 512, 3, "512->512 @256^2")):
    for nth in range(n):
        st.zero_()
        assert L.vt_debug_halo_stamps_nth(st.data_ptr(), H, Cin, nth + n) == 0      # the SECOND of three back-to-back steps
        for _ in range(3):                       # back-to-back steps, as in the bench; the counter restarts with the setter, so exactly one launch is stamped
            pipe.logits(x)
        torch.cuda.synchronize()
        s = st.cpu(); s = s[s[:, 6] > 0]
        assert L.vt_debug_halo_stamps_nth(None, 0, 0, 0) == 0
        if len(s) == 0:
            print(f"{label} launch {nth}: no stamps"); continue
        loop = (s[:, 3] - s[:, 2]).float() / 100
        clk = (s[:, 12] - s[:, 11]).double() / (s[:, 3] - s[:, 2]).double().clamp(min=1) * 100.0
        span = (s[:, 6].max() - s[:, 0].min()).item() / 100
        print(f"{label} launch {nth} of the step: {len(s)} workgroups, kernel span {span:8.1f} us, main loop median {loop.median().item():6.2f} us, "
              f"in-kernel clock median {clk.median().item():7.1f} MHz (p10 {clk.quantile(0.1).item():7.1f}, p90 {clk.quantile(0.9).item():7.1f})", flush=True)
