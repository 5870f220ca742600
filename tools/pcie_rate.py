"""What the step costs when the batch starts in HOST memory (the CLI's situation): fp32 NCHW tensors as the reference builds them vs
uint8 HWC pixels normalised on the device (vt_preprocess_u8), pageable vs pinned.   python tools/pcie_rate.py"""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
from vae_tagger_amd.modules import create_attention_decoder
from vae_tagger_amd.pipeline import EncodeTagPipeline
with contextlib.redirect_stdout(sys.stderr):
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    dec = create_attention_decoder(16, 128, 128, 10000, {"use_spatial_attention": True, "use_self_attention": True})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(10000), seed=1), strict=False)
    pipe = EncodeTagPipeline(DiffusersVAEWrapper(vae).to("cuda").eval(), dec.to("cuda").eval())
    pipe.check_finite = False
B = 16
x32 = synth.synth_images(B, 1024, 1024, seed=1000)
u8 = ((x32.permute(0, 2, 3, 1) * 0.5 + 0.5) * 255).round().to(torch.uint8).contiguous()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
xd = x32.cuda()
t_dev = timeit(lambda: pipe.logits(xd))
print(f"inputs resident in HBM: {t_dev:.2f} ms/step ({B / t_dev * 1e3:.1f} images/s)")
for name, host in (("fp32 NCHW pageable", x32), ("fp32 NCHW pinned", x32.pin_memory()), ("uint8 HWC pageable + device normalise", u8), ("uint8 HWC pinned + device normalise", u8.pin_memory())):
    if host.dtype == torch.uint8:
        fn = lambda h=host: pipe.logits(pipe.normalize_u8(h.cuda(non_blocking=True)))
    else:
        fn = lambda h=host: pipe.logits(h.cuda(non_blocking=True))
    t = timeit(fn)
    print(f"{name:40s}: {t:.2f} ms/step ({B / t * 1e3:.1f} images/s), +{t - t_dev:.2f} ms over resident inputs ({host.numel() * host.element_size() / 1e6:.0f} MB per batch)")
