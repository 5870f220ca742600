"""GPU parity, one kernel at a time, through the C ABI.  Oracle = torch CPU fp32 of the same op on
the same bf16-rounded operands (a floating-point path: fp32 accumulate on both sides), so the
tolerances below are accumulation-order noise, not bf16 slack."""
import pytest
import torch
import torch.nn.functional as F

from _util import Ops, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    return Ops()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(shape, generator=g)


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad_lo, pad_hi
    (2, 128, 128, 20, 24, 3, 1, 1, 1),      # ragged pixel tile, Cout=128 config
    (1, 128, 256, 16, 16, 3, 1, 1, 1),      # 256x256 config
    (1, 256, 512, 9, 13, 3, 1, 1, 1),       # odd sizes, two cout tiles
    (2, 128, 128, 17, 22, 3, 2, 0, 1),      # Downsample2D: pad (0,1,0,1), stride 2, odd input
    (1, 512, 512, 8, 8, 3, 2, 0, 1),
    (1, 128, 256, 12, 20, 1, 1, 0, 0),      # conv_shortcut 1x1
    (1, 512, 32, 16, 24, 3, 1, 1, 1),       # conv_out-shaped (Cout=32 config)
    (3, 64, 64, 5, 7, 3, 1, 1, 1),          # Cin = one K chunk, tiny image
    (1, 128, 128, 40, 50, 3, 1, 1, 1),      # halo kernel <4,2>: several 32x16 tiles, ragged right/bottom edges
    (2, 256, 256, 33, 17, 3, 1, 1, 1),      # halo kernel <2,4>: ragged 16x16 tiles
    (1, 512, 512, 16, 16, 3, 1, 1, 1),      # 16 channel chunks, two cout tiles
    (1, 32, 128, 9, 9, 3, 1, 1, 1),         # single channel chunk (last-chunk path only)
    (1, 256, 256, 70, 39, 3, 2, 0, 1),      # stride-2 phase-plane kernel: 3 x 2 tiles of 16 x 16 outputs, ragged both ways, 8 chunks, 2 cout tiles
    (2, 128, 128, 64, 64, 3, 2, 0, 1),      # even sizes: the (0,1,0,1) padding row / column is read
    (1, 512, 512, 33, 34, 3, 2, 0, 1),      # 16 chunks (plane ring wraps 21 times), 4 cout tiles, one tile + one row
    (1, 32, 128, 6, 6, 3, 2, 0, 1),         # single channel chunk
]


@pytest.fixture(params=["halo", "generic"])
def conv_kernel(request, ops):
    """3x3 stride-1 convs have two kernels: the halo-tile one (default) and the generic implicit GEMM."""
    ops.ctx.call("vt_set_flag", 0, 1 if request.param == "halo" else 0)
    yield request.param
    ops.ctx.call("vt_set_flag", 0, 1)


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride,plo,phi", CONV_CASES)
def test_conv2d_matches_torch(ops, conv_kernel, B, Cin, Cout, H, W, k, stride, plo, phi):
    x = bf16_round(_rand((B, Cin, H, W), 1))
    w = bf16_round(_rand((Cout, Cin, k, k), 2, (Cin * k * k) ** -0.5))
    b = _rand((Cout,), 3, 0.1)
    ref = F.conv2d(F.pad(x, (plo, phi, plo, phi)), w, b, stride=stride)
    res = _rand(tuple(ref.shape), 4)
    got32, got16 = ops.conv2d(x, w, b, residual_nchw=res, stride=stride, pad_lo=plo, pad_hi=phi, want="both")
    ref = ref + res
    assert got32.shape == ref.shape
    assert torch.allclose(got32, ref, rtol=1e-4, atol=2e-4), (got32 - ref).abs().max()
    assert torch.allclose(got16, bf16_round(ref), rtol=1e-2, atol=1e-2)
    got = ops.conv2d(x, w, None, stride=stride, pad_lo=plo, pad_hi=phi)
    assert torch.allclose(got, ref - res - b.view(1, -1, 1, 1), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("s2_halo", [1, 0], ids=["phase_plane", "generic"])
def test_conv2d_stride2_identity_taps(ops, s2_halo):
    """Downsample2D's conv (pad (0,1,0,1), stride 2) with one-hot weights, one tap at a time, on an asymmetric input: every tap must read
    in(2y + ky, 2x + kx) exactly (a swapped plane, shift or weight step shows as a wrong pixel, not as a tolerance)."""
    Cin = Cout = 128
    x = torch.arange(2 * Cin * 21 * 38, dtype=torch.float32).reshape(2, Cin, 21, 38) % 251 - 125.0
    ops.ctx.call("vt_set_flag", 13, s2_halo)
    try:
        for tap in range(9):
            w = torch.zeros(Cout, Cin, 3, 3)
            w[torch.arange(Cout), torch.arange(Cin), tap // 3, tap % 3] = 1.0
            ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
            got = ops.conv2d(x, w, None, stride=2, pad_lo=0, pad_hi=1)
            assert got.shape == ref.shape and torch.equal(got, ref), f"tap {tap}"
    finally:
        ops.ctx.call("vt_set_flag", 13, 1)


def test_conv2d_identity_weights_asymmetric_input(ops, conv_kernel):
    """A = I style check with an asymmetric operand: catches a transposed C/D map or a swapped tap."""
    Cin = Cout = 128
    x = torch.arange(2 * Cin * 6 * 10, dtype=torch.float32).reshape(2, Cin, 6, 10) % 251 - 125.0
    for tap in range(9):
        w = torch.zeros(Cout, Cin, 3, 3)
        w[torch.arange(Cout), torch.arange(Cin), tap // 3, tap % 3] = 1.0
        ref = F.conv2d(x, w, padding=1)
        got = ops.conv2d(x, w, None)
        assert torch.equal(got, ref), f"tap {tap}"


FUSED_CASES = [(2, 128, 128, 40, 50), (1, 128, 256, 33, 17), (1, 256, 256, 20, 36), (1, 512, 512, 16, 16), (2, 256, 128, 7, 5),
               (1, 32, 128, 9, 9), (1, 64, 256, 18, 18)]


@pytest.mark.parametrize("in_bf16", [False, True])
@pytest.mark.parametrize("B,Cin,Cout,H,W", FUSED_CASES)
def test_fused_norm_silu_conv3x3(ops, B, Cin, Cout, H, W, in_bf16):
    """GroupNorm-apply + SiLU inside the conv's halo staging == conv(bf16(silu(x*scale+shift)))."""
    x = _rand((B, Cin, H, W), 31) * 1.5 + 0.3
    if in_bf16:
        x = bf16_round(x)
    scale = 0.5 + torch.rand(B, Cin, generator=torch.Generator().manual_seed(32))
    shift = _rand((B, Cin), 33, 0.5)
    w = bf16_round(_rand((Cout, Cin, 3, 3), 34, (Cin * 9) ** -0.5))
    b = _rand((Cout,), 35, 0.1)
    res = _rand((B, Cout, H, W), 36)
    a = bf16_round(F.silu(x * scale.view(B, Cin, 1, 1) + shift.view(B, Cin, 1, 1)))
    ref = F.conv2d(a, w, b, padding=1) + res
    got = ops.norm_silu_conv3x3(x, scale, shift, w, b, residual_nchw=res, in_bf16=in_bf16)
    # the device SiLU (v_exp + v_rcp) can round a bf16 operand the other way now and then: bf16-level tolerance
    assert torch.allclose(got, ref, rtol=2e-3, atol=4e-3), (got - ref).abs().max()


GN_EPI_CASES = [(2, 128, 128, 40, 50, 3, 1, 1, 1), (1, 128, 256, 33, 17, 3, 1, 1, 1), (1, 256, 512, 20, 20, 3, 1, 1, 1),
                (2, 128, 128, 17, 22, 3, 2, 0, 1), (1, 256, 512, 12, 20, 1, 1, 0, 0)]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride,plo,phi", GN_EPI_CASES)
def test_conv_epilogue_groupnorm_statistics(ops, conv_kernel, B, Cin, Cout, H, W, k, stride, plo, phi):
    """GroupNorm (scale, shift) produced by the conv epilogue == GroupNorm of the conv output."""
    x = bf16_round(_rand((B, Cin, H, W), 21))
    w = bf16_round(_rand((Cout, Cin, k, k), 22, (Cin * k * k) ** -0.5))
    b = _rand((Cout,), 23, 0.1) + 3.0                         # a large common offset: mean >> std inside groups
    g = 1 + 0.1 * _rand((Cout,), 24)
    bt = 0.1 * _rand((Cout,), 25)
    out, ss = ops.conv2d_gn(x, w, b, g, bt, stride=stride, pad_lo=plo, pad_hi=phi)
    ref = F.conv2d(F.pad(x, (plo, phi, plo, phi)), w, b, stride=stride)
    assert torch.allclose(out, ref, rtol=1e-4, atol=2e-4)
    want = F.group_norm(ref, 32, g, bt, eps=1e-6)
    got = out * ss[:, :, 0].view(B, Cout, 1, 1) + ss[:, :, 1].view(B, Cout, 1, 1)
    assert torch.allclose(got, want, rtol=1e-3, atol=1e-3), (got - want).abs().max()



@pytest.mark.parametrize("B,Cin,Cout,H,W,stride", [(4, 128, 128, 264, 136, 1), (2, 128, 128, 520, 264, 1), (4, 128, 256, 136, 72, 1), (2, 256, 256, 132, 68, 1),
                                                   (4, 128, 128, 264, 136, 2), (2, 512, 512, 66, 34, 1)])
def test_conv_epilogue_statistics_are_bit_stable_run_to_run(ops, B, Cin, Cout, H, W, stride):
    """Round 4: on ragged tiles (H, W = 1 mod 8 after the stride: 136 x 264 inputs) the raw (n, mean, M2) partials of the halo conv's
    epilogue differed in about one run of four -- one element of one partial summed against pivot 0 (a packed-fp32 op_sel hazard:
    DESIGN.md 4.14, vt_common.h) -- and with them the next GroupNorm's (scale, shift) and the latents.  200 launches, every word equal."""
    import ctypes
    vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    dev = ops.dev
    g = torch.Generator().manual_seed(H + W + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.randn(Cout, generator=g).to(dev); gam = torch.ones(Cout, device=dev); bet = torch.zeros(Cout, device=dev)
    plo, phi = (1, 1) if stride == 1 else (0, 1)
    Ho, Wo = (H + plo + phi - 3) // stride + 1, (W + plo + phi - 3) // stride + 1
    o16 = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    n = ops.ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, Ho, Wo, Cout)
    ws = torch.zeros(n // 4 + 64, device=dev)
    ss = torch.zeros(B, Cout, 2, device=dev)

    def run():
        ops.ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, stride, plo, phi, 32, 1e-6, vp(gam), vp(bet),
                     vp(ss), vp(ws), ops.stream)
    run(); torch.cuda.synchronize()
    ref_ws, ref_ss, ref_o = ws.clone(), ss.clone(), o16.clone()
    assert ref_ss.abs().sum() > 0
    bad = 0
    for _ in range(200):
        run()
        bad += int(not (torch.equal(ws.view(torch.int32), ref_ws.view(torch.int32)) and torch.equal(ss.view(torch.int32), ref_ss.view(torch.int32))))
    torch.cuda.synchronize()
    assert bad == 0 and torch.equal(o16, ref_o), bad


@pytest.mark.parametrize("batch,M,N,K", [(1, 300, 200, 512), (2, 64, 512, 128), (1, 257, 108, 72), (1, 100, 104, 1000),
                                            (1, 300, 200, 8), (1, 130, 260, 40), (2, 64, 300, 96), (1, 256, 256, 2048), (1, 520, 516, 168),
                                            (2, 200, 384, 512), (1, 385, 256, 72)])
def test_gemm_nt_matches_torch(ops, batch, M, N, K):
    a = bf16_round(_rand((batch, M, K), 5))
    b = bf16_round(_rand((batch, N, K), 6, K ** -0.5))
    bias = _rand((N,), 7)
    ref = torch.matmul(a, b.transpose(1, 2)) * 0.5 + bias
    got = ops.gemm_nt(a, b, bias, alpha=0.5, lda=(K + 7) // 8 * 8 + 8, ldb=(K + 7) // 8 * 8)
    assert torch.allclose(got, ref, rtol=1e-4, atol=2e-4), (got - ref).abs().max()
    rb = _rand((M,), 8)
    got = ops.gemm_nt(a, b[:1], rb, bias_per_row=True, out_bf16=True, lda=(K + 7) // 8 * 8, ldb=(K + 7) // 8 * 8)
    ref = torch.matmul(a, b[:1].transpose(1, 2)) + rb.view(1, M, 1)
    assert torch.allclose(got, bf16_round(ref), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("C,H,W,in_bf16,silu", [(128, 40, 36, False, True), (256, 17, 9, False, True),
                                                 (512, 16, 16, True, True), (512, 8, 8, False, False),
                                                 (128, 64, 64, True, True), (64, 10, 10, False, True)])
def test_groupnorm_silu_matches_torch(ops, C, H, W, in_bf16, silu):
    x = _rand((2, C, H, W), 9) * 1.7 + 0.8
    if in_bf16:
        x = bf16_round(x)
    g = 1 + 0.1 * _rand((C,), 10)
    b = 0.1 * _rand((C,), 11)
    ref = F.group_norm(x, 32, g, b, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    got = ops.groupnorm(x, g, b, silu=silu, in_bf16=in_bf16)
    # output is bf16: half an ulp of bf16 (2^-9 relative) + stats noise
    assert torch.allclose(got, ref, rtol=6e-3, atol=2e-3), (got - ref).abs().max()


def test_groupnorm_fp16_input(ops):
    """the residual stream is stored as fp16 by default: GroupNorm reads it directly"""
    x = (_rand((2, 256, 24, 20), 19) * 1.3 + 0.4).to(torch.float16).to(torch.float32)
    g = 1 + 0.1 * _rand((256,), 10)
    b = 0.1 * _rand((256,), 11)
    ref = F.silu(F.group_norm(x, 32, g, b, eps=1e-6))
    got = ops.groupnorm(x, g, b, silu=True, in_f16=True)
    assert torch.allclose(got, ref, rtol=6e-3, atol=2e-3), (got - ref).abs().max()


def test_groupnorm_large_mean_is_stable(ops):
    x = _rand((1, 128, 32, 32), 12) * 0.05 + 30.0          # mean >> std: catastrophic for naive E[x^2]-E[x]^2 in fp32
    g, b = torch.ones(128), torch.zeros(128)
    ref = F.group_norm(x, 32, g, b, eps=1e-6)
    got = ops.groupnorm(x, g, b, silu=False)
    assert torch.allclose(got, ref, rtol=2e-2, atol=6e-2), (got - ref).abs().max()


@pytest.mark.parametrize("H,W", [(16, 16), (33, 70), (7, 130)])
def test_conv_in_matches_torch(ops, H, W):
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(13)) * 2 - 1
    w = _rand((128, 3, 3, 3), 14, 27 ** -0.5)
    b = _rand((128,), 15, 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    try:
        for mfma in (0, 1):
            # 0: exact fp32 VALU conv; 1: matrix cores with split (hi + lo) bf16 operands, ~2^-16 relative
            ops.ctx.call("vt_set_flag", 5, mfma)
            got32, got16 = ops.conv_in(x, w, b)
            assert torch.allclose(got32, ref, rtol=1e-5, atol=1e-5 if mfma == 0 else 5e-5), (mfma, (got32 - ref).abs().max())
            assert torch.allclose(got16, bf16_round(ref), rtol=1e-2, atol=1e-2)
    finally:
        ops.ctx.call("vt_set_flag", 5, 1)


@pytest.mark.parametrize("rows,n", [(5, 64), (3, 108), (4, 101), (2, 4096), (2, 5000), (1, 16384), (1, 20000)])
def test_softmax_rows(ops, rows, n):
    s = _rand((rows, n), 16) * 3
    s[0, n // 2] = 40.0                                      # a spike: max-subtraction must hold
    ref = torch.softmax(s, dim=-1)
    got = ops.softmax_rows(s)
    assert got.shape[1] % 8 == 0 and torch.all(got[:, n:] == 0)
    assert torch.allclose(got[:, :n], ref, rtol=1e-2, atol=1e-6)


def _e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


FP8_CASES = [
    # B, Cin, Cout, H, W
    (2, 128, 128, 40, 72),       # ragged 8x32 tiles on both edges, two channel chunks
    (1, 64, 256, 16, 32),        # one chunk (last-chunk path only), two cout tiles
    (1, 256, 128, 33, 35),       # odd sizes, four chunks
    (2, 512, 512, 9, 13),        # image smaller than a tile, eight chunks, four cout tiles
    (1, 128, 128, 64, 64),       # whole tiles only
]


@pytest.mark.parametrize("B,Cin,Cout,H,W", FP8_CASES)
def test_fp8_conv_matches_torch_on_the_same_quantised_operands(ops, B, Cin, Cout, H, W):
    """conv3x3_halo_fp8.hip (BASELINE configs[4], vt_set_flag 11): activations e4m3(8 x), weights e4m3 with per-cout absmax
    scales, fp32 accumulate.  Reference = torch fp32 conv of the SAME quantised operands (torch's own float8_e4m3fn
    conversion): what is left is accumulation-order noise, so fragment layout, swizzle and tap order are all pinned."""
    x = _rand((B, Cin, H, W), 1)
    x[0, :, 0, 0] = 0.001                       # e4m3 subnormals of 8 x
    w = _rand((Cout, Cin, 3, 3), 2, (Cin * 9) ** -0.5)
    b = _rand((Cout,), 3, 0.1)
    sc = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
    wq = _e4m3(w / sc) * sc

    def ref_of(xx):
        return F.conv2d(_e4m3(xx * 8.0) / 8.0, wq, None, padding=1)
    # the matrix pipe sums the 64 products of a block in a fixed-point-like adder aligned to the largest one (measured here:
    # ~1e-5 of the output scale on ordinary data, ~1e-4 of the largest product): far below e4m3's 2^-4, far above fp32 noise
    res = _rand((B, Cout, H, W), 4)
    got = ops.conv3x3_fp8(x, w, b, residual_nchw=res)
    ref = ref_of(x) + b.view(1, -1, 1, 1) + res
    assert torch.allclose(got, ref, rtol=1e-4, atol=3e-4), (got - ref).abs().max()
    got = ops.conv3x3_fp8(x, w)
    assert torch.allclose(got, ref_of(x), rtol=1e-4, atol=3e-4)
    x[0, 0, 1, 1] = 100.0                       # saturates at 448 / 8: the clamp is part of the contract
    got = ops.conv3x3_fp8(x, w)
    assert torch.allclose(got, ref_of(x), rtol=1e-4, atol=3e-3), (got - ref_of(x)).abs().max()


@pytest.mark.parametrize("B,Cin,Cout,H,W", FP8_CASES + [(1, 128, 128, 37, 131), (2, 128, 256, 16, 64)])
def test_fp8_conv_tile_shapes_are_bit_identical(ops, B, Cin, Cout, H, W):
    """vt_set_flag(ctx, 16, v): the fp8 halo conv on 16 x 32 px (1) and 8 x 64 px (2) tiles of 8 waves instead of 8 x 32 px on 4 waves --
    the same K-step order per output pixel, so the results are the default tile's bit for bit (ragged edges, images smaller than a tile,
    several cout tiles, with and without residual); + 4 applies the shape to layers with Cin > 128 too."""
    x = _rand((B, Cin, H, W), 11)
    w = _rand((Cout, Cin, 3, 3), 12, (Cin * 9) ** -0.5)
    b = _rand((Cout,), 13, 0.1)
    res = _rand((B, Cout, H, W), 14)
    base = ops.conv3x3_fp8(x, w, b, residual_nchw=res)
    base_plain = ops.conv3x3_fp8(x, w)
    try:
        for v in (5, 6):
            ops.ctx.call("vt_set_flag", 16, v)
            assert torch.equal(ops.conv3x3_fp8(x, w, b, residual_nchw=res), base), v
            assert torch.equal(ops.conv3x3_fp8(x, w), base_plain), v
    finally:
        ops.ctx.call("vt_set_flag", 16, 0)
    with pytest.raises(Exception):
        ops.ctx.call("vt_set_flag", 16, 3)


def test_fp8_mfma_block_sum_keeps_a_bounded_window_below_its_largest_product(ops):
    """Isolates what the saturated case above tolerates (atol 3e-3): v_mfma_scale_f32_32x32x64_f8f6f4 does not add the 64 products
    of a K-block in fp32 -- it aligns them to the block's LARGEST product and keeps a bounded number of bits below it.
    One output, one K-block (Cin = 64, centre tap only), every operand exactly representable in e4m3: channel 0 carries one big
    product B = 448 x 448 (in quantised units), the other 63 channels the product s each.  In fp32 the sum is B + 63 s exactly
    (all values < 2^24).  Observed on gfx950: for s >= 2^-14.6 B the result is exact; below that exactly SEVEN of the 63 small
    products are lost (B + 56 s comes back) -- the companions of the big product in its group of eight, the other groups add up
    exactly; without the big product the same small products always sum exactly.  Asserted: products within 2^-13 of the block's
    largest are kept exactly and the total error stays below 2^-12 of it, so that a different part or compiler would show."""
    Cin, Cout, H, W = 64, 128, 8, 32
    big = 448.0
    rows = []
    for j in range(0, 13):
        s_x = 2.0 ** (j - (j // 2))          # split s = 2^j between the two operands (both stay e4m3-exact powers of two)
        s_w = 2.0 ** (j // 2)
        for with_big in (True, False):
            xq = torch.full((1, Cin, H, W), s_x)            # quantised-domain values e4m3(8 x) should hold
            wq = torch.zeros(Cout, Cin, 3, 3)
            wq[:, :, 1, 1] = s_w
            if with_big:
                xq[:, 0] = big
            wq[:, 0, 1, 1] = big                             # (per-cout absmax = 448 -> weight scale exactly 1)
            got = ops.conv3x3_fp8(xq / 8.0, wq)              # x = xq / 8 exactly; out = sum(xq wq) / 8
            exact = ((big * big if with_big else s_x * big) + 63.0 * s_x * s_w) / 8.0
            v = got[0, 0, 3, 7].item()
            assert (got == v).all()                          # every output sees the same block
            rows.append((j, with_big, exact, v))
    for j, with_big, exact, v in rows:
        print(f"s = 2^{j:2d}  {'B + 63 s' if with_big else '448 s_x + 63 s':14s} exact {exact:12.3f}  device {v:12.3f}  diff {v - exact:9.3f}")
    B8 = big * big / 8.0
    for j, with_big, exact, v in rows:
        if not with_big:
            assert v == exact, (j, exact, v)                 # no big product: nothing to align to, the small ones add exactly
        else:
            assert abs(v - exact) <= B8 * 2.0 ** -12, (j, exact, v)
            if 2.0 ** j >= big * big * 2.0 ** -13:
                assert v == exact, (j, exact, v)             # products within 2^-13 of the largest are kept exactly


@pytest.mark.parametrize("s2_halo", [1, 0], ids=["phase_plane", "generic"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 128, 34, 50), (1, 256, 256, 16, 16), (1, 512, 512, 18, 22), (1, 128, 256, 70, 133), (2, 64, 128, 64, 64),
                                            (1, 1024, 128, 17, 65)])
def test_fp8_stride2_conv_matches_torch_on_the_same_quantised_operands(ops, s2_halo, B, Cin, Cout, H, W):
    """Downsample2D's conv (pad (0,1,0,1), stride 2) on e4m3 operands, the fp8 mode's stride-2 layers: conv3x3_s2_halo_fp8.hip (phase planes
    of the input, v_mfma_scale_f32_32x32x64_f8f6f4; ragged 8 x 32 tiles, odd / even sizes, 1-16 channel chunks, 1-4 cout tiles) and, with
    vt_set_flag(13, 0), conv_gemm_kernel<..., F8>.  The un-normalised residual stream is quantised as e4m3(x)."""
    x = 3.0 * _rand((B, Cin, H, W), 5)
    w = _rand((Cout, Cin, 3, 3), 6, (Cin * 9) ** -0.5)
    b = _rand((Cout,), 7, 0.1)
    sc = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
    wq = _e4m3(w / sc) * sc
    ref = F.conv2d(F.pad(_e4m3(x), (0, 1, 0, 1)), wq, b, stride=2)
    res = _rand(tuple(ref.shape), 8)
    ops.ctx.call("vt_set_flag", 13, s2_halo)
    try:
        got = ops.conv3x3_fp8(x, w, b, residual_nchw=res, stride=2)
    finally:
        ops.ctx.call("vt_set_flag", 13, 1)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref + res, rtol=1e-4, atol=2e-3), (got - ref - res).abs().max()


def test_fp8_stride2_conv_one_hot_taps(ops):
    """The fp8 phase-plane kernel with one-hot weights, one tap at a time, on small-integer inputs (exact in e4m3): every tap must read
    in(2y + ky, 2x + kx) exactly.  (The weight is 0.875, not 1: the kernel returns acc * (amax / 448) and 1 / 448 is not a float -- with 1.0 the
    outputs +-3, +-6, +-7 come back one ulp off, as the first GPU run of this test showed.)"""
    Cin = Cout = 128
    x = (torch.arange(2 * Cin * 19 * 70, dtype=torch.float32).reshape(2, Cin, 19, 70) % 17) - 8.0
    for tap in range(9):
        w = torch.zeros(Cout, Cin, 3, 3)
        w[torch.arange(Cout), torch.arange(Cin), tap // 3, tap % 3] = 0.875      # per-cout scale 0.875 / 448 = 2^-9: the epilogue's acc * scale is exact
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
        got = ops.conv3x3_fp8(x, w, stride=2)
        assert got.shape == ref.shape and torch.equal(got, ref), f"tap {tap}"


def test_c_abi_error_paths_return_codes_and_messages(ops):
    """The library never aborts: bad arguments, missing weights, short or misaligned workspaces and calls out of order come back
    as error codes with a message (the reference's per-image try/except relies on failures being catchable)."""
    import ctypes
    from vae_tagger_amd import _lib, synth
    from _util import vp
    L = _lib.load()
    ctx = _lib.Context(0)
    x = torch.zeros(1, 3, 64, 64, device="cuda")
    lat = torch.zeros(1, 16, 32, 32, device="cuda")           # two blocks = ONE downsample: 64 x 64 -> 32 x 32 latents (an 8 x 8 buffer here was overrun by
                                                              # 60 KB until round 4: an abort in torch.cuda.synchronize() in about one full-suite run of three)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    # nothing configured yet
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), vp(ws), ws.numel(), None) == 3          # VT_ERR_STATE
    assert b"finalized" in L.vt_last_error(ctx.handle)
    assert L.vt_encoder_finalize(ctx.handle) == 3
    blocks = (ctypes.c_int * 2)(64, 100)
    assert L.vt_encoder_configure(ctx.handle, 3, 16, blocks, 2, 2, 32, 1.0, 1, 0.0, 1) == 1               # channels not a multiple of 64
    blocks = (ctypes.c_int * 2)(64, 64)
    assert L.vt_encoder_configure(ctx.handle, 4, 16, blocks, 2, 2, 32, 1.0, 1, 0.0, 1) == 1               # in_channels != 3
    assert L.vt_encoder_configure(ctx.handle, 3, 16, blocks, 2, 1, 32, 1.0, 1, 0.0, 1) == 0
    assert L.vt_encoder_finalize(ctx.handle) == 4 and b"missing weight" in L.vt_last_error(ctx.handle)     # VT_ERR_MISSING_WEIGHT
    sd = synth.synth_state_dict(synth.encoder_manifest((64, 64), 3, 16, 1), seed=0)
    bad = dict(sd); bad["encoder.conv_in.weight"] = torch.zeros(64, 3, 5, 5)
    for k, v in bad.items():
        ctx.set_weight(k, v)
    assert L.vt_encoder_finalize(ctx.handle) == 1 and b"shape mismatch" in L.vt_last_error(ctx.handle)
    for k, v in sd.items():
        ctx.set_weight(k, v)
    assert L.vt_encoder_finalize(ctx.handle) == 0
    need = L.vt_encode_workspace_bytes(ctx.handle, 1, 64, 64)
    assert need > 0 and L.vt_encode_workspace_bytes(ctx.handle, 0, 64, 64) == 0 and L.vt_encode_workspace_bytes(ctx.handle, 1, 4, 64) == 0
    big = torch.zeros(need + 512, dtype=torch.uint8, device="cuda")
    p = (big.data_ptr() + 255) // 256 * 256
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p), need - 1, None) == 5    # VT_ERR_WORKSPACE
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p + 8), need, None) == 1   # misaligned
    assert L.vt_encode(ctx.handle, None, 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p), need, None) == 1
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 7, vp(lat), ctypes.c_void_p(p), need, None) == 1      # bad mode
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p), need, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all() and ctx.status() == 0
    # finalize twice without new weights: the first one consumed the host copies and the second frees the packed ones, so the
    # context must fall back to "not finalized" (no dangling device pointers) until weights are set and finalized again
    assert L.vt_encoder_finalize(ctx.handle) == 4 and b"missing weight" in L.vt_last_error(ctx.handle)
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p), need, None) == 3       # VT_ERR_STATE
    for k, v in sd.items():
        ctx.set_weight(k, v)
    assert L.vt_encoder_finalize(ctx.handle) == 0
    assert L.vt_encode(ctx.handle, vp(x), 1, 64, 64, 2, vp(lat), ctypes.c_void_p(p), need, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all()
    # decoder: out of order, bad configuration, sort arguments
    lg = torch.zeros(1, 11, device="cuda")
    assert L.vt_decode_logits(ctx.handle, vp(lat), 1, 32, 32, vp(lg), vp(ws), ws.numel(), None) == 3
    assert L.vt_decoder_configure(ctx.handle, 11, 8, 0, 1, 1, 0, 8) == 1                                    # latent_channels != 16
    assert L.vt_decoder_configure(ctx.handle, 11, 16, 0, 1, 1, 0, 3) == 1                                   # heads must divide 8
    conf = torch.zeros(1, 11, device="cuda"); idx = torch.zeros(1, 11, dtype=torch.int64, device="cuda")
    assert L.vt_get_confidence(ctx.handle, vp(lg), 1, 0, vp(conf), vp(idx), None) == 1
    assert L.vt_get_confidence(ctx.handle, vp(lg), 1, 11, vp(conf), None, None) == 1
    assert L.vt_set_flag(ctx.handle, 99, 1) == 1 and L.vt_set_flag(ctx.handle, 7, 5) == 1
    st = ctypes.c_int(7)
    assert L.vt_status(ctx.handle, 1, None, None) == 1 and L.vt_status(ctx.handle, 1, ctypes.byref(st), None) == 0 and st.value == 0
    assert L.vt_create(10 ** 6, ctypes.byref(ctypes.c_void_p())) == 2                                        # no such device
    ctx.close()
