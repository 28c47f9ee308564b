"""world_size-2 gloo tests of the multi-GPU plumbing (frame sharding + set-up broadcasts), on the CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from video_unscreen_amd.dist import broadcast_state, gather_stats, shard_range


def test_shard_range_partitions_every_frame_once():
    for F in (0, 1, 7, 300, 301, 2400):
        for G in (1, 2, 3, 8):
            seen = []
            for r in range(G):
                lo, hi = shard_range(F, r, G)
                assert 0 <= lo <= hi <= F
                seen += list(range(lo, hi))
            assert seen == list(range(F))                               # contiguous, ordered, complete
            sizes = [shard_range(F, r, G)[1] - shard_range(F, r, G)[0] for r in range(G)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    torch.manual_seed(0)
    sd = synthetic_state_dict(0) if rank == 0 else None
    bg = torch.arange(4 * 6 * 3, dtype=torch.uint8).view(4, 6, 3) if rank == 0 else torch.zeros(4, 6, 3, dtype=torch.uint8)
    sd2, bg2 = broadcast_state(sd, bg, torch.device("cpu"), world)
    # every rank must now hold rank 0's checkpoint, bit for bit, in the reference key order
    ref = synthetic_state_dict(0)
    same = list(sd2.keys()) == list(ref.keys()) and all(torch.equal(sd2[k], ref[k]) for k in ref)
    same = same and torch.equal(bg2, torch.arange(4 * 6 * 3, dtype=torch.uint8).view(4, 6, 3))
    # and the network built from it computes the same thing on each rank's own frame range
    net = MattingNet(sd2, device="cpu")
    lo, hi = shard_range(5, rank, world)
    outs = []
    for t in range(lo, hi):
        g = torch.Generator().manual_seed(100 + t)
        outs.append(float(net(torch.rand(1, 7, 32, 32, generator=g)).sum()))
    table = gather_stats([float(rank), float(hi - lo), float(same)], torch.device("cpu"))
    q.put((rank, same, lo, hi, outs, table.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shards_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert [(r[2], r[3]) for r in res] == [(0, 3), (3, 5)]
    # the sharded run equals a single-process run over the same frames
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    net = MattingNet(synthetic_state_dict(0), device="cpu")
    single = []
    for t in range(5):
        g = torch.Generator().manual_seed(100 + t)
        single.append(float(net(torch.rand(1, 7, 32, 32, generator=g)).sum()))
    assert res[0][4] + res[1][4] == single
    assert res[0][5] == [[0.0, 3.0, 1.0], [1.0, 2.0, 1.0]]


def _clip_for_bg(n=14, H=24, W=44):
    rng = np.random.default_rng(5)
    back = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    frames, masks = [], []
    for t in range(n):
        m = np.zeros((H, W), np.uint8)
        m[4:16, 3 + 2 * t:13 + 2 * t] = 255
        m[20:, :6] = 255                                   # never seen: the always-masked hole
        f = back.copy()
        f[m > 0] = rng.integers(0, 256, (int((m > 0).sum()), 3), dtype=np.uint8)
        frames.append(f)
        masks.append(m)
    return frames, masks


def _bg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from video_unscreen_amd.bgestimate import OverallBackground
    frames, masks = _clip_for_bg()
    H, W = masks[0].shape
    ob = OverallBackground(H, W, device="cpu")
    lo, hi = shard_range(len(frames), rank, world)
    raw, cont = ob.raw.numpy(), ob.cont.numpy()[..., None]            # views of the bucket; the oracle stands in for the kernel
    for t in range(lo, hi):
        oracle.overall_bg_accumulate(frames[t], masks[t], raw, cont)
    ob.reduce()
    q.put((rank, ob.raw.numpy().copy(), ob.cont.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_overall_background_allreduce_world2():
    """Stage 2 of bg_offline sharded by frames: per-rank partial sums + one all-reduce == the single-process accumulators."""
    import oracle
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames, masks = _clip_for_bg()
    H, W = masks[0].shape
    raw = np.zeros((H, W, 3), np.float32)
    cont = np.zeros((H, W, 1), np.float32)
    for f, m in zip(frames, masks):
        oracle.overall_bg_accumulate(f, m, raw, cont)
    for _, r, c in res:
        assert np.array_equal(r, raw) and np.array_equal(c, cont[..., 0])
    bg_always, hole = oracle.overall_bg_finalize(raw, np.repeat(cont, 3, -1))
    assert hole[20:, :6].all() and bg_always[hole == 0].any()


def _all_nets_worker(rank, world, port, q):
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0)) // world))      # (two ranks on one box: each its share of the cores)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from video_unscreen_amd.dist import load_and_broadcast, network_layouts
    names = ("vmatting", "schp", "deeplab", "stm")
    got = load_and_broadcast({n: "synthetic" for n in names}, torch.device("cpu"), world)      # only rank 0 builds / reads them
    ok = {}
    if rank != 0:
        from video_unscreen_amd.dist import load_checkpoint
        for n in names:
            ref = load_checkpoint(n, "synthetic")
            lay = network_layouts()[n]()
            ok[n] = (list(got[n].keys()) == [k for k, _ in lay] == list(ref.keys())
                     and all(got[n][k].dtype == ref[k].dtype and torch.equal(got[n][k], ref[k]) for k in ref))
    q.put((rank, ok, {n: sum(t.numel() for t in got[n].values()) for n in names}))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_all_networks_world2():
    """the weights of EVERY network of the path (matting UNet, SCHP, DeepLabv3+, STM) reach the other ranks from rank 0
    in one float bucket + one int64 bucket, bit for bit, in the reference's key order and dtypes."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_all_nets_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[1][1] == {"vmatting": True, "schp": True, "deeplab": True, "stm": True}
    assert res[0][2] == res[1][2] and res[0][2]["schp"] > 60e6 and res[0][2]["vmatting"] > 20e6


def test_shard_start_semantics(oracle_lib):
    """What a frame shard computes (stated in dist.py): every shard restarts alpha_pre like the reference's --range run
    (bg_offline.py:34-35, 162-163).  Pinned on the oracle chain: shard [0, 3) equals the whole-clip run on its frames; shard [3, 6)
    equals a run started at frame 3 - and its frames differ from the whole-clip run's only through alpha_pre, i.e. inside the
    unknown band of the trimap (known pixels are clamped to the trimap, vmatting/agent.py:128-130)."""
    import json
    import __graft_entry__ as ge
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    cfg = json.loads(bytes(np.load(os.path.join(os.path.dirname(__file__), "golden", "orchestration.npz"))["cfg_json"]).decode())
    H, W, n = 64, 96, 6
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    frames = [clip.frame(t) for t in range(n)]
    bga = clip.bg_always.numpy()

    class P:
        net = MattingNet(synthetic_state_dict(0), device="cpu", dtype=torch.float32)

        def cpu_net(self):
            return self.net

    def run(lo, hi):
        outs, alpha_pre = [], None
        for t in range(lo, hi):
            f = frames[t]
            o = ge.oracle_frame(oracle_lib, cfg, P(), f["frame"].numpy(), f["seg"].numpy(), f["bg"].numpy(), bga, alpha_pre)
            alpha_pre = o["alpha"]
            outs.append(o)
        return outs

    whole = run(0, n)
    shards = [run(*shard_range(n, r, 2)) for r in range(2)]
    assert [shard_range(n, r, 2) for r in range(2)] == [(0, 3), (3, 6)]
    for t in range(3):                                     # rank 0: identical to the one-process run
        assert np.array_equal(shards[0][t]["alpha"], whole[t]["alpha"]) and np.array_equal(shards[0][t]["fg"], whole[t]["fg"])
    differs = False
    for i, t in enumerate(range(3, 6)):                    # rank 1: same trimaps; alpha may differ, only where the trimap is unknown
        a, b = shards[1][i], whole[t]
        assert np.array_equal(a["trimap"], b["trimap"])
        d = a["alpha"] != b["alpha"]
        assert not d[a["trimap"] != 128].any()
        differs |= bool(d.any())
    assert differs, "the cut must be visible somewhere, or this test pins nothing"


def test_bench_gpus_flag_is_not_ignored():
    """bench.py --gpus N means N ranks (the flag used to be parsed and never read).  Without a launcher it starts them
    itself - here, on a box without N GPUs, that is refused with a message instead of silently running one rank; under a launcher whose
    WORLD_SIZE differs from --gpus it exits non-zero before touching a GPU and prints no JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "USC_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--gpus 64" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
