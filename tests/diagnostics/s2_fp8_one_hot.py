"""Diagnostic for test_fp8_stride2_conv_one_hot_taps: repeat it and describe any mismatch (count, NaN-ness, positions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from _util import Ops
ops = Ops()
Cin = Cout = 128
x = (torch.arange(2 * Cin * 19 * 70, dtype=torch.float32).reshape(2, Cin, 19, 70) % 17) - 8.0
bad = 0
for rep in range(12):
    for tap in range(9):
        w = torch.zeros(Cout, Cin, 3, 3)
        w[torch.arange(Cout), torch.arange(Cin), tap // 3, tap % 3] = 0.875      # per-cout scale 0.875 / 448 = 2^-9: the epilogue's acc * scale is exact
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
        got = ops.conv3x3_fp8(x, w, stride=2)
        if not torch.equal(got, ref):
            bad += 1
            d = (got != ref) | torch.isnan(got)
            idx = d.nonzero()
            print(f"rep {rep} tap {tap}: {int(d.sum())} of {d.numel()} differ, NaN {int(torch.isnan(got).sum())}, inf {int(torch.isinf(got).sum())}; "
                  f"images {sorted(set(idx[:, 0].tolist()))} couts {idx[:, 1].min().item()}..{idx[:, 1].max().item()} rows {sorted(set(idx[:, 2].tolist()))} "
                  f"cols {idx[:, 3].min().item()}..{idx[:, 3].max().item()}", flush=True)
            for i in idx[:6].tolist():
                print("    ", i, "got", got[tuple(i)].item(), "ref", ref[tuple(i)].item(), flush=True)
print("mismatching calls:", bad, "of", 12 * 9)
