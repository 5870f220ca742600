"""Where does a halo-conv workgroup spend its time?  Needs the diagnostic build (tools/build_variant.sh stamp -DHALO_STAMP).
   python tools/stamp_halo.py vae_tagger_amd/csrc/exp/libvt_stamp.so
Stamps are s_memrealtime (100 MHz): 0 entry, 1 prologue DMA issued, 2 first operands landed + fragments read, 3 main loop
done, 4 epilogue stores issued, 5 GroupNorm partials done, 6 stores acknowledged (vmcnt(0)); 7 = (XCC_ID << 32) | HW_ID."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
ctx.lib.vt_debug_halo_stamps.argtypes = [ctypes.c_void_p]; ctx.lib.vt_debug_halo_stamps.restype = ctypes.c_int
def med(x): return x.float().median().item() * 10.0      # 100 MHz ticks -> ns
for (B, H, W, Cin, Cout, occ2, gn) in ((16, 512, 512, 256, 256, 3, 1), (16, 512, 512, 256, 256, 0, 1), (8, 1024, 1024, 128, 128, 3, 1)):
    ctx.call("vt_set_flag", 3, occ2)
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev); gam = torch.ones(Cout, device=dev); bet = torch.zeros(Cout, device=dev)
    o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    ss = torch.zeros(B, Cout, 2, device=dev)
    ws = torch.zeros(ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, H, W, Cout) // 4 + 64, device=dev)
    rows = 16 if (Cout % 256 == 0 or occ2) else 32
    bc = 128 if (occ2 == 3 or Cout % 256) else 256
    nwg = B * (H // rows) * (W // 16) * (Cout // bc)
    st = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
    def call():
        if gn:
            ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, 32, 1e-6, vp(gam), vp(bet), vp(ss), vp(ws), None)
        else:
            ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    assert ctx.lib.vt_debug_halo_stamps(None) == 0
    for _ in range(3): call()
    assert ctx.lib.vt_debug_halo_stamps(st.data_ptr()) == 0
    call(); torch.cuda.synchronize()
    assert ctx.lib.vt_debug_halo_stamps(None) == 0
    s = st.cpu()
    t = s[:, :11] - s[:, 0:1].min()
    names = ["entry->prologue issued", "prologue issued->operands landed", "main loop", "epilogue stores issued", "GN partials", "stores acknowledged"]
    for lab, i0, i1 in (("  loop end -> last MFMA landed", 3, 8), ("  -> bias landed", 8, 9), ("  -> stores issued", 9, 4), ("  -> barrier before GN partials", 4, 10), ("  -> GN partials done", 10, 5)):
        if i1 == 10 and not gn: continue
        if i0 == 10 and not gn: continue
        d = t[:, i1] - t[:, i0]
        print(f"   {lab:34s} median {med(d) / 1e3:7.2f} us   p10 {d.float().quantile(0.1).item() / 100:7.2f}  p90 {d.float().quantile(0.9).item() / 100:7.2f}")
    print(f"B{B} {H}x{W} {Cin}->{Cout} occ2={occ2} gn={gn}: {nwg} workgroups, kernel span {t[:, 6].max().item() * 10 / 1e3:.1f} us")
    for i, n in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"   {n:34s} median {med(d) / 1e3:7.2f} us   p10 {d.float().quantile(0.1).item() / 100:7.2f}  p90 {d.float().quantile(0.9).item() / 100:7.2f}")
    tot = t[:, 6] - t[:, 0]
    print(f"   {'workgroup total':34s} median {med(tot) / 1e3:7.2f} us")
    # gap between consecutive workgroups on the same CU slot: key = (xcc, se/sh/cu bits of HW_ID) [+ which of the co-resident slots]
    key = (s[:, 7] >> 32) * 65536 + ((s[:, 7] & 0xffffffff) >> 8 & 0xff)
    gaps = []
    for k in key.unique().tolist():
        m = (key == k).nonzero().flatten()
        ent = t[m, 0].sort().values; end = t[m, 6].sort().values
        per = 2 if (occ2 == 3 or (occ2 and Cout == 128)) else 1
        if len(ent) > per:
            gaps.append((ent[per:] - end[:-per]).float())
    g = torch.cat(gaps)
    print(f"   CUs seen {len(key.unique())}; next workgroup's entry minus the end of the one it replaces: median {g.median().item() / 100:.2f} us  p10 {g.quantile(0.1).item() / 100:.2f}  p90 {g.quantile(0.9).item() / 100:.2f}", flush=True)
ctx.call("vt_set_flag", 3, 3)
