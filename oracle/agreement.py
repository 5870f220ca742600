"""ORACLE (test infrastructure, not product code): end-to-end argsort / top-k agreement between the
oracle's tag order and the order the measured path returns -- the second half of BASELINE.json's
metric ("bit-exact tag-index argsort"; SURVEY.md section 8(d): "argsort agreement computed on
logits with index-ascending tie-break").

Follows reference modules.py:470-475 (`get_confidence`: sigmoid, then a descending sort whose
indices are the tag order) as it is consumed at infer_full.py:106-125 (threshold on the sorted
confidences, tags named by the sorted indices).

What "bit-exact" can mean between two paths whose logits differ by up to d = max |dlogit|: a tag
whose oracle logit is more than 2 d away from both of its neighbours in the oracle's order cannot
change rank (every tag above it stays above it, every tag below it below), so at those ranks the
two index arrays MUST be identical -- a disagreement there is a defect of the sort or of the
logits, not rounding.  At the other ranks the order is decided inside the tolerance band and is
reported, not asserted: fraction of identical positions, largest rank displacement, top-k sets.
The thresholded tag set (confidence >= t  <=>  logit >= logit(t)) must contain every tag whose
oracle logit clears the threshold by more than d and no tag whose oracle logit misses it by more
than d.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity leg may import this.
"""
import math

import torch

from .decoder_ref import get_confidence


def argsort_agreement(ref_logits, got_logits, got_idx, threshold=0.5, max_abs_dlogit=None):
    """ref_logits, got_logits: fp32 [N] (one image); got_idx: int64 [N], the measured path's sorted tag indices.
    Returns a dict of plain Python values (JSON-ready)."""
    ref_logits = ref_logits.detach().reshape(-1).to(torch.float32).cpu()
    got_logits = got_logits.detach().reshape(-1).to(torch.float32).cpu()
    got_idx = got_idx.detach().reshape(-1).to(torch.int64).cpu()
    n = ref_logits.numel()
    d = float((got_logits - ref_logits).abs().max()) if max_abs_dlogit is None else float(max_abs_dlogit)
    _, ref_idx = get_confidence(ref_logits[None])
    ref_idx = ref_idx[0]
    srt = ref_logits[ref_idx]
    gap = srt[:-1] - srt[1:]                                # >= 0, n - 1 of them
    clear = torch.ones(n, dtype=torch.bool)
    clear[:-1] &= gap > 2.0 * d
    clear[1:] &= gap > 2.0 * d
    same = got_idx == ref_idx
    bad = torch.nonzero(clear & ~same).reshape(-1)
    differ = torch.nonzero(~same).reshape(-1)
    # rank displacement of every tag between the two orders
    pos_ref = torch.empty(n, dtype=torch.int64); pos_ref[ref_idx] = torch.arange(n)
    pos_got = torch.empty(n, dtype=torch.int64); pos_got[got_idx] = torch.arange(n)
    # the displaced tags must all sit inside the band: |ref logit of the tag at rank r in one order - in the other| <= 2 d
    band_ok = bool(((ref_logits[got_idx] - srt).abs() <= 2.0 * d + 1e-12).all())
    t = min(max(float(threshold), 1e-12), 1.0 - 1e-12)
    lt = math.log(t / (1.0 - t))
    got_set = got_logits >= lt                              # what `conf >= threshold` keeps, up to sigmoid's own rounding at the edge
    must = ref_logits > lt + d
    may = ref_logits >= lt - d
    out = {
        "definition": "indices identical at every rank whose oracle logit is > 2*max|dlogit| from both neighbours in the oracle's order "
                      "(descending logit, ascending index on ties; reference modules.py:470-475); other ranks reported, not asserted",
        "ranks": n, "max_abs_dlogit": float(f"{d:.3e}"),
        "ranks_compared": int(clear.sum()), "frac_ranks_compared": round(float(clear.float().mean()), 5),
        "identical_at_compared_ranks": bool(bad.numel() == 0),
        "first_disagreeing_compared_rank": None if bad.numel() == 0 else int(bad[0]),
        "frac_identical_positions": round(float(same.float().mean()), 5),
        "first_differing_rank": None if differ.numel() == 0 else int(differ[0]),
        "max_rank_displacement": int((pos_ref - pos_got).abs().max()),
        "swaps_stay_inside_the_2d_band": band_ok,
        "top1_identical": bool(got_idx[0] == ref_idx[0]),
        "top5_set_identical": bool(set(got_idx[:5].tolist()) == set(ref_idx[:5].tolist())) if n >= 5 else None,
        "top10_set_identical": bool(set(got_idx[:10].tolist()) == set(ref_idx[:10].tolist())) if n >= 10 else None,
        "threshold": float(threshold),
        "tags_above_threshold_oracle": int((ref_logits >= lt).sum()), "tags_above_threshold_measured": int(got_set.sum()),
        "threshold_set_matches_outside_the_band": bool((got_set | ~must).all() and (may | ~got_set).all()),
    }
    return out
