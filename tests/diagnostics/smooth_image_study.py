import sys, torch, time
sys.path.insert(0,'/root/repo')
from oracle import encoder_ref
from vae_tagger_amd import synth
import torch.nn.functional as F
torch.set_num_threads(8)
res=int(sys.argv[1]) if len(sys.argv)>1 else 256
def smooth(seed, res, noise=0.08):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(16, 16, 3, generator=g)
    arr = torch.nn.functional.interpolate(low.permute(2, 0, 1)[None], size=(res, res), mode="bicubic", align_corners=False)[0].permute(1, 2, 0)
    arr = ((arr + noise * torch.randn(res, res, 3, generator=g)).clamp(0, 1) * 255).to(torch.uint8)
    return (((arr.permute(2,0,1).float()/255.0)-0.5)/0.5)[None]
sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
def f16(t): return t.to(torch.float16).to(torch.float32)
def bf(t): return t.to(torch.bfloat16).to(torch.float32)
# level-aware rounding: patch _resnet/_attention/_conv via a global "current level" that the quantiser reads
cur = {"lvl": None}
orig_resnet, orig_attn, orig_conv, orig_gn = encoder_ref._resnet, encoder_ref._attention, encoder_ref._conv, encoder_ref._gn
def lvl_of(p):
    if "down_blocks" in p: return "down" + p.split("down_blocks.")[1][0]
    if "mid_block" in p: return "mid"
    return "out"
def resnet(h, sd_, p, q):
    cur["lvl"] = lvl_of(p); return orig_resnet(h, sd_, p, q)
def attn(h, sd_, p, q):
    cur["lvl"] = "attn"; return orig_attn(h, sd_, p, q)
def conv(x, sd_, name, q, stride=1, padding=1):
    if "downsamplers" in name: cur["lvl"] = "ds" + name.split("down_blocks.")[1][0]
    if "conv_out" in name: cur["lvl"] = "out"
    return orig_conv(x, sd_, name, q, stride, padding)
def gn(x, sd_, name, q, silu):
    if "conv_norm_out" in name: cur["lvl"] = "out"
    return orig_gn(x, sd_, name, q, silu)
encoder_ref._resnet, encoder_ref._attention, encoder_ref._conv, encoder_ref._gn = resnet, attn, conv, gn
class Q:
    def __init__(s, lo_levels): s.lo = lo_levels
    def __call__(s, t):
        return bf(t) if (cur["lvl"] in s.lo) else f16(t)
def run(x, q):
    old = encoder_ref._Q
    encoder_ref._Q = lambda on: q
    try: return encoder_ref.vae_wrapper_encode(sd, x, emulate_bf16=True)
    finally: encoder_ref._Q = old
ALL = ["down0","ds0","down1","ds1","down2","ds2","down3","mid","attn","out"]
for name, x in (("smooth s11", smooth(11,res)), ("noise", synth.synth_images(1,res,res,seed=3))):
    ref = encoder_ref.vae_wrapper_encode(sd, x)
    for lo in ([], ALL, *[[l] for l in ALL], ["down0","ds0"], ["down0","ds0","down1","ds1"], ["down0","ds0","down1","ds1","down2","ds2"]):
        d = run(x, Q(lo)) - ref
        print(f"{res} {name:12s} bf16 on {str(lo):60s} (fp16 elsewhere) max {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}")
