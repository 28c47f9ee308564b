"""Frame sharding across ranks (one process per GPU).

The per-frame path has no data-path collective: frames couple only through alpha_pre, and the reference's own
`--range a-b` runs (tools/unscreen/bg_offline.py:34-35, 184-192) restart that chain, so rank r processing the
contiguous range shard_range(F, r, G) is bit-identical to the reference invoked with that range.  The only
exchanges are set-up broadcasts (the weights of every network on the path, the overall background) from rank 0 - RCCL over
xGMI on the GPU box (backend "nccl"), gloo in the CPU tests - flattened into ONE bucket per dtype so each is a single large
collective instead of ~2000 small ones - and a final gather of per-rank statistics.

World-size dependence (stated, not hidden): a shard starts like a reference run with that --range does - alpha_pre = the
first frame's gate (bg_offline.py:162-163), and in the online mode the first frame goes through HumanSegAgent, not STM
(bg.py:45-56).  The frames right after a cut therefore differ from those of ONE run over the whole clip; every frame equals
the reference's own output for a run over that shard's range (tests/test_dist_cpu.py pins both statements).
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


def network_layouts():
    """Checkpoint layouts ((key, shape) lists in the reference's key order) of every network on the path."""
    from .binseg.deeplab_net import deeplab_checkpoint_layout
    from .binseg.schp_net import schp_checkpoint_layout
    from .stm.net import stm_checkpoint_layout
    return {"vmatting": checkpoint_layout, "schp": schp_checkpoint_layout, "deeplab": deeplab_checkpoint_layout,
            "stm": stm_checkpoint_layout}


def broadcast_networks(state_dicts, device, world, src=0):
    """state_dicts: {"vmatting" | "schp" | "deeplab" | "stm": state dict in the reference key layout} - the values are read on
    rank `src` only (other ranks pass the same keys, any values).  Every float tensor of every network travels in ONE flat
    float32 bucket (UNet 87 MB + SCHP 267 MB + ... = one large collective over RCCL / xGMI instead of ~2000 small ones), the
    BatchNorm counters in one int64 bucket.  Returns {name: state dict on the CPU} on every rank; world == 1 is a pass-through."""
    if world == 1:
        return dict(state_dicts)
    is_src = dist.get_rank() == src
    layouts = network_layouts()
    names = sorted(state_dicts)
    fkeys, ikeys = [], []
    for n in names:
        for k, shp in layouts[n]():
            (ikeys if k.endswith("num_batches_tracked") else fkeys).append((n, k, shp))
    ft = [state_dicts[n][k].float() if is_src else torch.empty(shp, dtype=torch.float32) for n, k, shp in fkeys]
    it = [state_dicts[n][k].long() if is_src else torch.empty(shp, dtype=torch.long) for n, k, shp in ikeys]
    got = {}
    if ft:
        got.update({(n, k): t for (n, k, _), t in zip(fkeys, _bucket_broadcast(ft, src, device))})
    if it:
        got.update({(n, k): t for (n, k, _), t in zip(ikeys, _bucket_broadcast(it, src, device))})
    return {n: OrderedDict((k, got[(n, k)].cpu()) for k, _ in layouts[n]()) for n in names}


def load_checkpoint(name, path):
    """One network's checkpoint in the reference's file convention -> state dict in the reference key layout.  "synthetic" / None:
    the keyed-deterministic weights (no trained checkpoint exists offline)."""
    synthetic = path is None or path == "synthetic"
    if name == "vmatting":                                  # {'state_dict': ...} (vmatting/agent.py:31)
        from .vmatting.net import synthetic_state_dict
        return synthetic_state_dict(0) if synthetic else torch.load(path, map_location="cpu")["state_dict"]
    if name == "schp":                                      # bare dict with DataParallel's module. prefix (agent_human.py:141-148)
        from .binseg.schp_net import schp_synthetic_state_dict, strip_module_prefix
        return schp_synthetic_state_dict(0) if synthetic else strip_module_prefix(torch.load(path, map_location="cpu"))
    if name == "deeplab":                                   # bare dict (binseg/agent.py:58)
        from .binseg.deeplab_net import deeplab_synthetic_state_dict
        return deeplab_synthetic_state_dict(0) if synthetic else torch.load(path, map_location="cpu")
    if name == "stm":                                       # bare dict (stm/agent.py:52)
        from .stm.net import stm_synthetic_state_dict
        return stm_synthetic_state_dict(0) if synthetic else torch.load(path, map_location="cpu")
    raise KeyError(name)


def load_and_broadcast(paths, device, world, src=0):
    """paths: {network name: checkpoint path | "synthetic"}.  Rank `src` reads the files, everyone gets the weights over the
    collective backend (one bucket): no rank but `src` touches the checkpoint files."""
    rank = dist.get_rank() if world > 1 else src
    sds = {n: (load_checkpoint(n, p) if rank == src else None) for n, p in paths.items()}
    return broadcast_networks(sds, device, world, src)


def broadcast_state(state_dict, bg_always, device, world, src=0, extra=None):
    """Rank `src` supplies the matting checkpoint (reference key layout), the overall background image and - `extra` - the
    other networks' checkpoints ({"schp" | "deeplab" | "stm": state dict}); every rank returns (state_dict on CPU, bg_always on
    `device`[, extra on CPU]).  world == 1 is a pass-through."""
    if world == 1:
        bg = bg_always.to(device)
        return (state_dict, bg) if extra is None else (state_dict, bg, dict(extra))
    nets = {"vmatting": state_dict}
    nets.update(extra or {})
    got = broadcast_networks(nets, device, world, src)
    bg = bg_always.to(device).contiguous()
    dist.broadcast(bg, src=src)
    sd = got.pop("vmatting")
    return (sd, bg) if extra is None else (sd, bg, got)


def gather_stats(values, device):
    """All ranks contribute a small float vector; every rank gets the (world, n) table."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t[None]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out)
