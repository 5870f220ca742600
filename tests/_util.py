"""Test helpers: golden loading, bf16 helpers, thin ctypes drivers for the single-operator ABI."""
import ctypes
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k].copy()) for k in z.files}


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def vp(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def latent_input(shape, seed):
    """Must match oracle/make_goldens.py:latent_input."""
    g = torch.Generator().manual_seed(seed)
    return 0.1159 + 0.8 * torch.randn(shape, generator=g)


def checksum(x):
    return torch.tensor([x.double().sum().item(), x.double().abs().sum().item()], dtype=torch.float64)


class Ops:
    """Single-operator entry points of the C ABI on cuda:0."""

    def __init__(self):
        from vae_tagger_amd import _lib
        self.L = _lib
        self.ctx = _lib.Context(0)
        self.dev = torch.device("cuda:0")
        self.stream = ctypes.c_void_p(0)

    def conv2d(self, x_nchw, w_oihw, bias=None, residual_nchw=None, stride=1, pad_lo=1, pad_hi=1, want="f32"):
        """x fp32 NCHW (rounded to bf16 on the way in), w fp32 OIHW -> fp32 NCHW result of the HIP conv."""
        B, Cin, H, W = x_nchw.shape
        Cout, _, k, _ = w_oihw.shape
        x = nhwc(x_nchw).to(self.dev, torch.bfloat16)
        w = w_oihw.permute(0, 2, 3, 1).contiguous().to(self.dev, torch.bfloat16)
        b = bias.to(self.dev, torch.float32).contiguous() if bias is not None else None
        Ho = (H + pad_lo + pad_hi - k) // stride + 1
        Wo = (W + pad_lo + pad_hi - k) // stride + 1
        r = nhwc(residual_nchw).to(self.dev, torch.float32) if residual_nchw is not None else None
        o32 = torch.full((B, Ho, Wo, Cout), float("nan"), device=self.dev, dtype=torch.float32) if want in ("f32", "both") else None
        o16 = torch.zeros((B, Ho, Wo, Cout), device=self.dev, dtype=torch.bfloat16) if want in ("bf16", "both") else None
        self.ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), vp(r), vp(o32), vp(o16), B, H, W, Cin, Cout, k, stride,
                      pad_lo, pad_hi, self.stream)
        torch.cuda.synchronize()
        outs = []
        if o32 is not None:
            outs.append(nchw(o32.cpu()))
        if o16 is not None:
            outs.append(nchw(o16.float().cpu()))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def norm_silu_conv3x3(self, x_nchw, scale, shift, w_oihw, bias, residual_nchw=None, in_bf16=False):
        """conv3x3(silu(x*scale[b,c] + shift[b,c])) with the affine+SiLU fused into the conv staging -> fp32 NCHW"""
        B, Cin, H, W = x_nchw.shape
        Cout = w_oihw.shape[0]
        x = nhwc(x_nchw).to(self.dev, torch.bfloat16 if in_bf16 else torch.float32)
        ss = torch.stack([scale, shift], dim=-1).to(self.dev, torch.float32).contiguous()     # [B, Cin, 2]
        w = w_oihw.permute(0, 2, 3, 1).contiguous().to(self.dev, torch.bfloat16)
        b = bias.to(self.dev, torch.float32).contiguous()
        r = nhwc(residual_nchw).to(self.dev, torch.float32) if residual_nchw is not None else None
        o32 = torch.full((B, H, W, Cout), float("nan"), device=self.dev, dtype=torch.float32)
        self.ctx.call("vt_op_norm_silu_conv3x3", vp(x), self.L.VT_BF16 if in_bf16 else self.L.VT_F32, vp(ss), vp(w), vp(b),
                      vp(r), vp(o32), None, B, H, W, Cin, Cout, self.stream)
        torch.cuda.synchronize()
        return nchw(o32.cpu())

    def conv2d_gn(self, x_nchw, w_oihw, bias, gamma, beta, residual_nchw=None, stride=1, pad_lo=1, pad_hi=1, groups=32, eps=1e-6):
        """-> (conv output fp32 NCHW, scale_shift [B,Cout,2]) with the statistics taken in the conv epilogue."""
        B, Cin, H, W = x_nchw.shape
        Cout, _, k, _ = w_oihw.shape
        x = nhwc(x_nchw).to(self.dev, torch.bfloat16)
        w = w_oihw.permute(0, 2, 3, 1).contiguous().to(self.dev, torch.bfloat16)
        b = bias.to(self.dev, torch.float32).contiguous()
        Ho = (H + pad_lo + pad_hi - k) // stride + 1
        Wo = (W + pad_lo + pad_hi - k) // stride + 1
        r = nhwc(residual_nchw).to(self.dev, torch.float32) if residual_nchw is not None else None
        o32 = torch.zeros((B, Ho, Wo, Cout), device=self.dev, dtype=torch.float32)
        ss = torch.zeros((B, Cout, 2), device=self.dev, dtype=torch.float32)
        n = self.ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, Ho, Wo, Cout)
        ws = torch.empty(n + 256, device=self.dev, dtype=torch.uint8)
        g = gamma.to(self.dev, torch.float32).contiguous()
        bt = beta.to(self.dev, torch.float32).contiguous()
        self.ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), vp(r), vp(o32), None, B, H, W, Cin, Cout, k, stride, pad_lo,
                      pad_hi, groups, float(eps), vp(g), vp(bt), vp(ss), vp(ws), self.stream)
        torch.cuda.synchronize()
        return nchw(o32.cpu()), ss.cpu()

    def conv3x3_fp8(self, x_nchw, w_oihw, bias=None, residual_nchw=None, stride=1):
        """x, w fp32 -> fp32 NCHW result of the fp8 (e4m3) halo conv; operands are quantised inside the call exactly as the
        encoder quantises them with vt_set_flag(ctx, 11, 1)."""
        B, Cin, H, W = x_nchw.shape
        Cout = w_oihw.shape[0]
        x = nhwc(x_nchw).to(self.dev, torch.float32)
        w = w_oihw.to(self.dev, torch.float32).contiguous()
        b = bias.to(self.dev, torch.float32).contiguous() if bias is not None else None
        r = nhwc(residual_nchw).to(self.dev, torch.float32) if residual_nchw is not None else None
        out = torch.full((B, H // stride, W // stride, Cout), float("nan"), device=self.dev, dtype=torch.float32)
        n = self.ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
        assert n > 0
        ws = torch.empty(n + 256, device=self.dev, dtype=torch.uint8)
        ptr = (ws.data_ptr() + 255) // 256 * 256
        self.ctx.call("vt_op_conv3x3_fp8", vp(x), vp(w), vp(b), vp(r), vp(out), B, H, W, Cin, Cout, stride, ctypes.c_void_p(ptr), self.stream)
        torch.cuda.synchronize()
        return nchw(out.cpu())

    def gemm_nt(self, a, b, bias=None, alpha=1.0, bias_per_row=False, out_bf16=False, lda=None, ldb=None, ldo=None):
        """a [batch,M,K] fp32, b [batch or 1,N,K] fp32 -> [batch,M,N]"""
        batch, M, K = a.shape
        N = b.shape[1]
        lda = lda or K
        ldb = ldb or K
        ldo = ldo or (N + 3) // 4 * 4
        A = torch.zeros(batch, M, lda, device=self.dev, dtype=torch.bfloat16)
        A[:, :, :K] = a.to(self.dev)
        Bm = torch.zeros(b.shape[0], N, ldb, device=self.dev, dtype=torch.bfloat16)
        Bm[:, :, :K] = b.to(self.dev)
        bs = bias.to(self.dev, torch.float32).contiguous() if bias is not None else None
        out = torch.full((batch, M, ldo), float("nan"), device=self.dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        # K is handed over rounded up to the 8-element chunk; pad columns are zero in both operands
        Kc = (K + 7) // 8 * 8
        self.ctx.call("vt_op_gemm_nt", vp(A), vp(Bm), vp(bs), None if out_bf16 else vp(out), vp(out) if out_bf16 else None,
                      batch, M, N, Kc, lda, ldb, ldo, M * lda, (N * ldb if b.shape[0] > 1 else 0), M * ldo,
                      float(alpha), int(bias_per_row), self.stream)
        torch.cuda.synchronize()
        return out[:, :, :N].float().cpu()

    def groupnorm(self, x_nchw, gamma, beta, groups=32, eps=1e-6, silu=True, in_bf16=False, in_f16=False):
        B, C, H, W = x_nchw.shape
        x = nhwc(x_nchw).to(self.dev, torch.bfloat16 if in_bf16 else (torch.float16 if in_f16 else torch.float32))
        y = torch.zeros(B, H, W, C, device=self.dev, dtype=torch.bfloat16)
        n = self.ctx.lib.vt_op_groupnorm_workspace_bytes(B, H * W, C)
        ws = torch.empty(n + 256, device=self.dev, dtype=torch.uint8)
        g = gamma.to(self.dev, torch.float32).contiguous()
        bt = beta.to(self.dev, torch.float32).contiguous()
        self.ctx.call("vt_op_groupnorm", vp(x), self.L.VT_BF16 if in_bf16 else (self.L.VT_F16 if in_f16 else self.L.VT_F32), B, H * W, C, groups,
                      float(eps), vp(g), vp(bt), int(silu), vp(y), vp(ws), self.stream)
        torch.cuda.synchronize()
        return nchw(y.float().cpu())

    def conv_in(self, x_nchw, w_oihw, bias):
        B, _, H, W = x_nchw.shape
        Cout = w_oihw.shape[0]
        x = x_nchw.to(self.dev, torch.float32).contiguous()
        w = w_oihw.to(self.dev, torch.float32).contiguous()
        b = bias.to(self.dev, torch.float32).contiguous()
        o32 = torch.zeros(B, H, W, Cout, device=self.dev, dtype=torch.float32)
        o16 = torch.zeros(B, H, W, Cout, device=self.dev, dtype=torch.bfloat16)
        ws = torch.empty(27 * Cout * 4 + 256, device=self.dev, dtype=torch.uint8)
        self.ctx.call("vt_op_conv_in", vp(x), vp(w), vp(b), vp(o32), vp(o16), B, H, W, Cout, vp(ws), self.stream)
        torch.cuda.synchronize()
        return nchw(o32.cpu()), nchw(o16.float().cpu())

    def softmax_rows(self, s, ldp=None):
        rows, n = s.shape
        ldp = ldp or (n + 7) // 8 * 8
        sd = s.to(self.dev, torch.float32).contiguous()
        p = torch.full((rows, ldp), float("nan"), device=self.dev, dtype=torch.bfloat16)
        self.ctx.call("vt_op_softmax_rows", vp(sd), vp(p), rows, n, n, ldp, self.stream)
        torch.cuda.synchronize()
        return p.float().cpu()
