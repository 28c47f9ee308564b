"""Frame sharding across ranks (one process per GPU).

The per-frame path has no data-path collective: frames couple only through alpha_pre, and the reference's own
`--range a-b` runs (tools/unscreen/bg_offline.py:34-35, 184-192) restart that chain, so rank r processing the
contiguous range shard_range(F, r, G) is bit-identical to the reference invoked with that range.  The only
exchanges are set-up broadcasts (weights, overall background) from rank 0 - RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests - flattened into ONE bucket per dtype so each is a single large
collective instead of 426 small ones - and a final gather of per-rank statistics.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist

from .vmatting.net import checkpoint_layout


def shard_range(num_frames, rank, world):
    """Contiguous, balanced [lo, hi) ranges: the first (num_frames % world) ranks get one extra frame."""
    base, extra = divmod(num_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _bucket_broadcast(tensors, src, device):
    """Broadcast same-dtype tensors as one flat bucket; returns the received tensors (on `device`)."""
    sizes = [t.numel() for t in tensors]
    flat = torch.empty(sum(sizes), dtype=tensors[0].dtype, device=device)
    if dist.get_rank() == src:
        torch.cat([t.reshape(-1).to(device) for t in tensors], out=flat)
    dist.broadcast(flat, src=src)
    out, o = [], 0
    for t, n in zip(tensors, sizes):
        out.append(flat[o:o + n].view(t.shape))
        o += n
    return out


def broadcast_state(state_dict, bg_always, device, world, src=0):
    """Rank `src` supplies the matting checkpoint (reference key layout) and the overall background image;
    every rank returns (state_dict on CPU, bg_always on `device`).  world == 1 is a pass-through."""
    if world == 1:
        return state_dict, bg_always.to(device)
    layout = checkpoint_layout()
    is_src = dist.get_rank() == src
    fkeys = [(k, s) for k, s in layout if not k.endswith("num_batches_tracked")]
    ikeys = [(k, s) for k, s in layout if k.endswith("num_batches_tracked")]
    ft = [state_dict[k].float() if is_src else torch.empty(s, dtype=torch.float32) for k, s in fkeys]
    it = [state_dict[k].long() if is_src else torch.empty(s, dtype=torch.long) for k, s in ikeys]
    ft = _bucket_broadcast(ft, src, device)
    it = _bucket_broadcast(it, src, device)
    got = {k: t for (k, _), t in zip(fkeys, ft)}
    got.update({k: t for (k, _), t in zip(ikeys, it)})
    sd = OrderedDict((k, got[k].cpu()) for k, _ in layout)
    bg = bg_always.to(device).contiguous()
    dist.broadcast(bg, src=src)
    return sd, bg


def gather_stats(values, device):
    """All ranks contribute a small float vector; every rank gets the (world, n) table."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t[None]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)
