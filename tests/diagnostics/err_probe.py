"""Latent error of the HIP encoder vs the fp32 oracle under the precision switches (flag 5: MFMA conv_in, flag 4: fp16 residual
stream), next to the oracle's own bf16-operand emulation.  python tests/diagnostics/err_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import encoder_ref
from vae_tagger_amd import synth
from vae_tagger_amd.diffusers_vae_loader import DiffusersVAEWrapper, get_diffusers_vae_config, load_diffusers_vae_from_config
sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
vae = load_diffusers_vae_from_config(get_diffusers_vae_config()); vae.load_state_dict(sd_e, strict=False)
m = DiffusersVAEWrapper(vae).to("cuda").eval()
ctx = m.vae._context()
for (w, h) in ((1024, 832), (512, 512)):
    x = synth.synth_images(1, h, w, seed=w * 4096 + h)
    ref = encoder_ref.vae_wrapper_encode(sd_e, x)
    emu = encoder_ref.vae_wrapper_encode(sd_e, x, emulate_bf16=True)
    print(w, h, "oracle bf16 emulation:", (emu - ref).abs().max().item())
    for flags in ({}, {5: 0}, {4: 0}, {5: 0, 4: 0}):
        for f, v in flags.items(): ctx.call("vt_set_flag", f, v)
        lat = m.encode(x.cuda()).cpu()
        print("   flags", flags, "max|d|", (lat - ref).abs().max().item(), "rms", (lat - ref).pow(2).mean().sqrt().item())
        for f in flags: ctx.call("vt_set_flag", f, 1)
