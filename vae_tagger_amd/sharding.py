"""Data-parallel sharding of the encode+tag path.

Images are independent, so the batch is split by image across ranks (one process per GPU, full
weight copy each: 68 MB bf16) and the only exchange is ONE all-gather of logits per step
([B/world, N] fp32 per rank -- 640 KB at 16 x 10 000) over RCCL/xGMI (`torch.distributed` backend
"nccl" on ROCm; "gloo" in the CPU tests).  No collective touches the encoder.

Bucketed input (reference AspectRatioBucketing, modules.py:180-222): whole same-shape batches are
assigned to ranks by a cost model -- conv work scales with pixels, mid-block attention with pixels^2.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank`; earlier ranks take the remainder."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def image_cost(width, height):
    """Relative encoder FLOPs of one image: 4.3326 TF of convs + projections
    scale with H*W, 0.5498 TF of QK^T + PV with (H*W)^2, both quoted at 1024^2."""
    px = (width * height) / float(1024 * 1024)
    return 4.3328 * px + 0.5498 * px * px


def assign_batches(batches, world):
    """batches: list of (width, height, n_images).  Greedy longest-processing-time assignment of whole
    same-shape batches to ranks.  Returns (per-rank list of batch indices, per-rank cost)."""
    order = sorted(range(len(batches)), key=lambda i: -image_cost(batches[i][0], batches[i][1]) * batches[i][2])
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        w, h, n = batches[i]
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += image_cost(w, h) * n
    return out, loads


def all_gather_logits(local_logits, counts=None, group=None, force_collective=False):
    """Gather per-rank [n_r, N] logits into [sum n_r, N] in rank order.

    Equal counts use one all_gather_into_tensor (a single RCCL all-gather); ragged counts pad to
    the max and trim.  Works on any backend (gloo on CPU for tests).  A one-rank group returns its
    input without a collective unless `force_collective` (the RCCL smoke test: the same call on a
    one-rank communicator)."""
    if not dist.is_available() or not dist.is_initialized():
        return local_logits
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return local_logits
    n_local, N = local_logits.shape
    if counts is None:
        c = torch.tensor([n_local], dtype=torch.int64, device=local_logits.device)
        cl = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(cl, c, group=group)
        counts = [int(t.item()) for t in cl]
    mx = max(counts)
    if all(c == mx for c in counts):
        out = torch.empty(world * mx, N, dtype=local_logits.dtype, device=local_logits.device)
        if dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, local_logits.contiguous(), group=group)
        else:
            dist.all_gather_into_tensor(out, local_logits.contiguous(), group=group)
        return out
    padded = torch.zeros(mx, N, dtype=local_logits.dtype, device=local_logits.device)
    padded[:n_local] = local_logits
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def sharded_logits(compute_logits, images, group=None):
    """Split `images` [B,...] by image over the ranks, run `compute_logits` on the local slice and
    all-gather.  Every rank returns the full [B, N] logits in the original image order."""
    if not dist.is_available() or not dist.is_initialized():
        return compute_logits(images)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = images.shape[0]
    lo, hi = shard_range(B, rank, world)
    counts = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    local = compute_logits(images[lo:hi]) if hi > lo else None
    if B < world:
        # some ranks hold no image (every rank knows which: B and world are global): they join the ragged gather with
        # zero rows, and learn the tag count from the ranks that computed something -- nobody is left waiting
        n = torch.tensor([0 if local is None else local.shape[1]], dtype=torch.int64, device=images.device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        if local is None:
            local = torch.zeros(0, int(n.item()), dtype=torch.float32, device=images.device)
    return all_gather_logits(local, counts, group)
