"""One 256->256 @512^2 halo conv (batch 16) a few times: the workload for an SQ counter pass.
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python3 tools/pmc_halo.py [occ2 mode]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
ctx.call("vt_set_flag", 3, mode)
B, H, W, Cin, Cout = 16, 512, 512, 256, 256
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
b = torch.zeros(Cout, device=dev)
o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
torch.cuda.synchronize()
