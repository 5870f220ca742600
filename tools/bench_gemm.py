"""Time + determinism-screen the batched NT GEMM (attention shapes) through the op-level C ABI.
   python tools/bench_gemm.py [lib.so] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
print("lib:", _lib.LIB_PATH, flush=True)
for (batch, M, N, K) in ((4, 16384, 16384, 512), (4, 16384, 512, 16384), (16, 4096, 512, 2304), (4, 4096, 4096, 4096)):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(batch, M, K, generator=g).to(dev, torch.bfloat16)
    Bm = (torch.randn(batch, N, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16)
    out = torch.empty(batch, M, N, device=dev, dtype=torch.bfloat16)
    for ring in (0,):
        def run():
            ctx.call("vt_op_gemm_nt", vp(A), vp(Bm), None, None, vp(out), batch, M, N, K, K, K, N, M * K, N * K, M * N, 1.0, 0, None)
        run(); torch.cuda.synchronize()
        ref = out.clone()
        if True:
            chk = torch.matmul(A[0, :512].float(), Bm[0].float().t())
            err = (ref[0, :512].float() - chk).abs().max().item()
        bad = 0
        for _ in range(reps):
            run()
            if not torch.equal(out, ref): bad += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"batch {batch} M{M} N{N} K{K}: {ms:.3f} ms  {2.0*batch*M*N*K/ms/1e9:.0f} TFLOP/s  nondeterministic reps {bad}/{reps}  max|err| vs fp32 {err:.3e}", flush=True)
