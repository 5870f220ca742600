"""Where does an fp8 halo-conv workgroup spend its time IN THE PRODUCT CONFIGURATION (e4m3 input from the GroupNorm pass, fp16
residual + fp16 output, GroupNorm partials), and what clock does the chip hold inside the main loop?
Needs the diagnostic build:  tools/build_variant.sh stamp -DHALO_STAMP
   python tools/stamp_halo_fp8.py vae_tagger_amd/csrc/exp/libvt_stamp.so [seconds of load before the clock reading, default 2]
The whole encoder step runs (batch 16 x 1024^2, fp8 mode); a device-side filter (H, Cin, residual present) selects the launches
that write stamps.  Stamps (s_memrealtime, 100 MHz): 0 entry, 1 prologue DMA issued, 2 first operands landed + fragments read,
3 main loop done, 9 residual tile landed in LDS, 4 stores issued, 5 GroupNorm partials done, 6 stores acknowledged;
11 / 12 = s_memtime (shader clock) at 2 / 3: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6),
read after >= 2 s of back-to-back steps on random data, median over workgroups.  Also the bf16 kernel's clock (same pair)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
LOAD_S = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
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
L = pipe.ctx.lib
L.vt_debug_halo_fp8_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]; L.vt_debug_halo_fp8_stamps.restype = ctypes.c_int
L.vt_debug_halo_stamps.argtypes = [ctypes.c_void_p]; L.vt_debug_halo_stamps.restype = ctypes.c_int
B = 16
x = synth.synth_images(B, 1024, 1024, seed=1000).to(dev)
st = torch.zeros(1 << 17, 16, dtype=torch.int64, device=dev)


def load(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        pipe.logits(x)
        torch.cuda.synchronize()


def report(tag, nwg, gn, res, per_cu_slots=2):
    s = st[:nwg].cpu()
    ok = s[:, 6] > 0
    s = s[ok]
    t = s - s[:, 0:1].min()
    q = lambda d, p: d.float().quantile(p).item() / 100
    print(f"{tag}: {int(ok.sum())} workgroups stamped, kernel span {(s[:, 6].max() - s[:, 0].min()).item() / 100:.1f} us")
    rows = [("entry -> prologue DMA issued", 0, 1), ("-> first operands landed, fragments read", 1, 2), ("main loop", 2, 3)]
    if res:
        rows += [("loop end -> residual tile in LDS", 3, 9), ("-> stores issued", 9, 4)]
    else:
        rows += [("loop end -> stores issued", 3, 4)]
    rows += [("-> GroupNorm partials done" if gn else "-> (no partials)", 4, 5), ("-> stores acknowledged", 5, 6), ("workgroup total", 0, 6)]
    for lab, i0, i1 in rows:
        d = s[:, i1] - s[:, i0]
        print(f"   {lab:42s} median {q(d, 0.5):7.2f} us   p10 {q(d, 0.1):7.2f}  p90 {q(d, 0.9):7.2f}")
    clk = (s[:, 12] - s[:, 11]).double() / (s[:, 3] - s[:, 2]).double().clamp(min=1) * 100.0      # MHz
    print(f"   in-kernel clock over the main loop (d s_memtime / d s_memrealtime): median {clk.median().item():7.1f} MHz   p10 {clk.quantile(0.1).item():7.1f}  p90 {clk.quantile(0.9).item():7.1f}")
    key = (s[:, 7] >> 32) * 65536 + ((s[:, 7] & 0xffffffff) >> 8 & 0xff)
    gaps = []
    for k in key.unique().tolist():
        m = (key == k).nonzero().flatten()
        ent = t[m, 0].sort().values; end = t[m, 6].sort().values
        if len(ent) > per_cu_slots:
            gaps.append((ent[per_cu_slots:] - end[:-per_cu_slots]).float())
    # are the two resident workgroups of a CU in phase?  For every workgroup: the share of its main-loop interval [2, 3] during which
    # the other workgroup(s) on the same CU are inside THEIR main loops (1 = lockstep: both loops share the matrix pipe and both sets of
    # tile ends are exposed; 0 = alternating: one workgroup's ends run under the other's loop)
    import numpy as np
    k_np = key.numpy(); a2 = t[:, 2].numpy().astype(np.float64); a3 = t[:, 3].numpy().astype(np.float64)
    share, first = [], []
    for k in np.unique(k_np):
        m = np.nonzero(k_np == k)[0]
        o = m[np.argsort(a2[m])]
        for ii, w in enumerate(o):
            ov = 0.0
            for v in o[max(0, ii - 3):ii + 4]:
                if v != w:
                    ov += max(0.0, min(a3[w], a3[v]) - max(a2[w], a2[v]))
            share.append(ov / max(a3[w] - a2[w], 1.0))
        if len(o) >= 2:
            first.append((a2[o[1]] - a2[o[0]]) / 100.0)
    share = np.array(share)
    print(f"   share of a main loop spent beside the co-resident workgroup's main loop: median {np.median(share):.2f}  p10 {np.quantile(share, 0.1):.2f}  p90 {np.quantile(share, 0.9):.2f}"
          f"   (first two workgroups of a CU start their loops {np.median(first):.2f} us apart)")
    if gaps:
        g = torch.cat(gaps)
        print(f"   CUs seen {len(key.unique())}; next workgroup's entry minus the end of the one it replaces: median {g.median().item() / 100:.2f} us  p10 {g.quantile(0.1).item() / 100:.2f}  p90 {g.quantile(0.9).item() / 100:.2f}", flush=True)


def dispatch_map(tag):
    """which CU do the first workgroups of a grid land on? (xcc, cu) of blocks 0..7, 8..15, 256..263, 512..519"""
    s = st.cpu()
    cu = lambda i: ((s[i, 7] >> 32).item(), ((s[i, 7] & 0xffffffff) >> 8 & 0xff).item())
    print(f"{tag}: dispatch map (xcc, cu) " + "; ".join(f"b{i}:{cu(i)}" for i in (0, 1, 8, 16, 24, 256, 264, 512, 520, 2048, 2056)))


pipe.set_fp8(True)
for _ in range(3):
    pipe.logits(x)
torch.cuda.synchronize()
load(LOAD_S)
# (H, Cin, 1 + residual): stage-0 conv1 / conv2 at 128 channels, a 256-channel and a 512-channel layer
for tag, H, Cin, res, nwg, gn in (("fp8 128->128 @1024^2 conv1 (no residual)", 1024, 128, 1, 65536, 1), ("fp8 128->128 @1024^2 conv2 (fp16 residual)", 1024, 128, 2, 65536, 1),
                                   ("fp8 256->256 @512^2 conv2 (fp16 residual)", 512, 256, 2, 32768, 1), ("fp8 512->512 @256^2 conv2 (fp16 residual)", 256, 512, 2, 16384, 1)):
    st.zero_()
    assert L.vt_debug_halo_fp8_stamps(st.data_ptr(), H, Cin, res) == 0
    pipe.logits(x); torch.cuda.synchronize()
    assert L.vt_debug_halo_fp8_stamps(None, 0, 0, 0) == 0
    report(tag, nwg, gn, res == 2)
    if H == 1024 and res == 2:
        dispatch_map(tag)
pipe.set_fp8(False)
for _ in range(3):
    pipe.logits(x)
load(LOAD_S)
# bf16 kernel: no filter in that build -- the last launch that covers a block index wins; the 128-channel 1024^2 layers are the
# only ones with 65536+ workgroups, so block indices >= 32768 can only come from them
st.zero_()
assert L.vt_debug_halo_stamps(st.data_ptr()) == 0
pipe.logits(x); torch.cuda.synchronize()
assert L.vt_debug_halo_stamps(None) == 0
s = st.cpu()
for lab, lo, hi in (("bf16 halo, 128-channel 1024^2 layers (blocks 32768..)", 32768, 131072), ("bf16 halo, last launch covering blocks 0..4095 (512-channel 128^2 layers)", 0, 4096)):
    z = s[lo:hi]; z = z[z[:, 3] > z[:, 2]]
    clk = (z[:, 12] - z[:, 11]).double() / (z[:, 3] - z[:, 2]).double().clamp(min=1) * 100.0
    print(f"{lab}: {len(z)} workgroups; main loop median {(z[:, 3] - z[:, 2]).float().median().item() / 100:.2f} us; in-kernel clock median {clk.median().item():7.1f} MHz  p10 {clk.quantile(0.1).item():7.1f}  p90 {clk.quantile(0.9).item():7.1f}")
