"""Does running two half-batches on two HIP streams (kernels of different phases co-scheduled by the hardware) beat one full
batch on one stream?  python tools/two_stream.py [batch=16] [steps=10]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
from vae_tagger_amd import synth
from vae_tagger_amd._runtime import vp
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    dec = create_attention_decoder(16, 128, 128, 10000, {"use_spatial_attention": True, "use_self_attention": True})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(10000), seed=1), strict=False)
    pipe = EncodeTagPipeline(DiffusersVAEWrapper(vae).to(dev).eval(), dec.to(dev).eval())
    pipe.check_finite = False
for fv in sys.argv[3:]:
    f, v = fv.split("=")
    pipe.ctx.call("vt_set_flag", int(f), int(v))
x = synth.synth_images(B, 1024, 1024, seed=1000).to(dev)
lib, h = pipe.ctx.lib, pipe.ctx.handle

def make(n):
    need = lib.vt_encode_tag_workspace_bytes(h, n, 1024, 1024)
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    return ws, (ws.data_ptr() + 255) // 256 * 256, need, torch.empty(n, 10000, device=dev)

def run(xs, w, stream):
    ws, ptr, need, out = w
    pipe.ctx.call("vt_encode_tag", vp(xs), xs.shape[0], 1024, 1024, None, vp(out), ctypes.c_void_p(ptr), need, ctypes.c_void_p(stream.cuda_stream))
    return out

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
full = make(B)
ms1 = timeit(lambda: run(x, full, s0))
ref = run(x, full, s0).clone(); torch.cuda.synchronize()
for parts in (2, 4):
    n = B // parts
    ws = [make(n) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def split():
        return [run(x[i * n:(i + 1) * n], ws[i], streams[i % 2]) for i in range(parts)]
    msp = timeit(split)
    outs = torch.cat(split()); torch.cuda.synchronize()
    print(f"{parts} parts of {n} on 2 streams: {msp:.2f} ms/step ({B / msp * 1e3:.1f} img/s)  bit-identical to the single-stream result: {torch.equal(outs, ref)}", flush=True)
    # the same parts back to back on ONE stream (what splitting alone costs)
    def serial():
        return [run(x[i * n:(i + 1) * n], ws[i], s0) for i in range(parts)]
    print(f"{parts} parts of {n} on 1 stream : {timeit(serial):.2f} ms/step", flush=True)
print(f"one batch of {B} on 1 stream: {ms1:.2f} ms/step ({B / ms1 * 1e3:.1f} img/s)", flush=True)
