import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, torch.nn.functional as F
from _util import Ops, nhwc, nchw, vp
ops = Ops()
def e4(t): return t.clamp(-448, 448).to(torch.float8_e4m3fn)
B, Cin, Cout, H, W = 2, 128, 128, 40, 72
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, W, generator=g)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "sub": x[0, :, 0, 0] = 0.001
if mode == "sat": x[0, 0, 1, 1] = 100.0
if mode == "flush": x = torch.where(x.abs() * 8 < 2 ** -6, torch.zeros_like(x), x)      # no e4m3 subnormal anywhere
print("mode", mode)
w = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (Cin * 9) ** -0.5
xd = nhwc(x).cuda(); wd = w.cuda().contiguous()
out = torch.empty(B, H, W, Cout, device="cuda")
n = ops.ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
ws = torch.zeros(n + 256, device="cuda", dtype=torch.uint8)
ptr = (ws.data_ptr() + 255) // 256 * 256
off = ptr - ws.data_ptr()
ops.ctx.call("vt_op_conv3x3_fp8", vp(xd), vp(wd), None, None, vp(out), B, H, W, Cin, Cout, 1, ctypes.c_void_p(ptr), ctypes.c_void_p(0))
torch.cuda.synchronize()
x8 = ws[off:off + B * H * W * Cin].cpu()
ref8 = e4(nhwc(x) * 8.0).view(torch.uint8).flatten()
bad = (x8 != ref8).nonzero().flatten()
print("activation bytes differing from torch:", len(bad), "of", len(x8))
for i in bad[:10].tolist():
    print("  x*8 =", (nhwc(x).flatten()[i] * 8).item(), "device", hex(x8[i]), "torch", hex(ref8[i]))
xq = e4(x * 8).float() / 8
sc = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
wq = e4(w / sc).float() * sc
ref = F.conv2d(xq, wq, None, padding=1)
got = nchw(out.cpu())
d = (got - ref).abs()
print("max err", d.max().item(), "at", (d == d.max()).nonzero()[0].tolist(), "ref there", ref.flatten()[d.argmax()].item(), " mean err", d.mean().item())
ref64 = F.conv2d(xq.double(), wq.double(), None, padding=1)
print("torch fp32 conv vs fp64:", (ref.double() - ref64).abs().max().item(), " device vs fp64:", (got.double() - ref64).abs().max().item())
