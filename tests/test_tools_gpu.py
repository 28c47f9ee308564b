"""GPU test of the drop-in entry point: tools/unscreen/bg_step.py on a small synthetic dataset on disk."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bg_step_tool_end_to_end(tmp_path, oracle_lib):
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import read_img, save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, n = 270, 480, 4
    root = tmp_path / "data"
    src = root / "src_img" / "7"
    dst = root / "test_img" / "7"
    src.mkdir(parents=True)
    dst.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    save_img(clip.bg_always.numpy(), str(dst / "always_bg.jpg"))
    for t in range(n):
        f = clip.frame(t)
        save_img(f["frame"].numpy(), str(src / f"{t:06d}.jpg"))
        save_img(np.stack([f["seg"].numpy()] * 3, -1), str(dst / f"segmask_{t:06d}.jpg"))
        save_img(f["bg"].numpy(), str(dst / f"bg_{t:06d}.jpg"))
    wpath = tmp_path / "unet.pth"
    torch.save({"state_dict": synthetic_state_dict(0)}, str(wpath))
    cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
    cfg["vmatting"]["model_path"] = str(wpath)
    cpath = tmp_path / "bg.json"
    json.dump(cfg, open(cpath, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "bg_step.py"), "--cfg", str(cpath),
                        "--video_id", "7", "--data_root", str(root)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    for t in range(n):
        a = read_img(str(dst / f"alphamask_{t:06d}.jpg"))
        fg = read_img(str(dst / f"fg_{t:06d}.jpg"))
        assert a.shape == (H, W, 3) and fg.shape == (H, W, 3)
    # the files the tool wrote agree with an in-process run on the same decoded inputs (JPEG output is lossy: <= 8 levels)
    from video_unscreen_amd.pipeline import BgStepPipeline
    pipe = BgStepPipeline(cfg, H, W, state_dict=synthetic_state_dict(0))
    bga = torch.from_numpy(read_img(str(dst / "always_bg.jpg"))).cuda()
    for t in range(n):
        out = pipe.step(torch.from_numpy(read_img(str(src / f"{t:06d}.jpg"))).cuda(),
                        torch.from_numpy(read_img(str(dst / f"segmask_{t:06d}.jpg"))).cuda(),
                        torch.from_numpy(read_img(str(dst / f"bg_{t:06d}.jpg"))).cuda(), bga)
        a = read_img(str(dst / f"alphamask_{t:06d}.jpg"))[..., 0].astype(int)
        assert np.abs(a - out["alpha"].cpu().numpy().astype(int)).mean() < 1.0


def _bench(args, env, launcher=True, port="29533"):
    pre = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", port] if launcher else [sys.executable])
    return subprocess.run(pre + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)


def _one_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                         # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal():
    """bench.py for N = 2, with the two ranks sharing this box's single GPU and gloo instead of RCCL: (a) launched exactly as the
    driver launches it (torch.distributed.run, one process per rank), (b) as the bare `python bench.py --gpus 2`, which must start
    the two ranks itself, (c) the strong-scaling form: ONE clip of 31 frames (no segmentation batch divides either range) cut into
    two contiguous ranges.  Exercises the broadcast of weights / background, the frame sharding, the barrier-bracketed timing, the
    MAX over ranks and the single JSON line; (d) a launcher whose world size is not --gpus is refused."""
    env = dict(os.environ, USC_BENCH_BACKEND="gloo", USC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    common = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-table"]
    out = _one_line(_bench(["--gpus", "2"] + common, env))
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 2 * 6 / (out["ms_per_step"] * 6 / 1e3)) < 1e-2 * out["value"]
    out = _one_line(_bench(["--gpus", "2"] + common, env, launcher=False))           # the form the driver's N = 1 command has
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak" and out["value"] > 0
    out = _one_line(_bench(["--gpus", "2", "--clip-frames", "31", "--height", "270", "--width", "480"] + common[2:], env, launcher=False))
    assert out["n_gpus"] == 2 and out["steps"] == 31 and out["scaling"] == "strong"
    assert abs(out["value"] - 31 / (out["ms_per_step"] * 31 / 1e3)) < 1e-2 * out["value"]
    assert "rank 0: 16 frames" in out["config"]["workload"]
    r = _bench(["--gpus", "4"] + common, env, port="29534")                          # 2 ranks started, 4 claimed
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_replace_mode_two_ranks_rehearsal():
    """bench.py --mode replace (BASELINE.json configs[4], the compositing half) for N = 2 on this box's one GPU over gloo: frames shard
    over the ranks, the centroid offsets meet in one all-reduce, rank 0 prints one line whose output was checked against the stand-alone
    kernel chain inside the run."""
    env = dict(os.environ, USC_BENCH_BACKEND="gloo", USC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = _one_line(_bench(["--gpus", "2", "--mode", "replace", "--steps", "9", "--warmup", "2", "--height", "270", "--width", "480", "--no-cpu-baseline"],
                           env, launcher=False))
    assert out["n_gpus"] == 2 and out["steps"] == 9 and out["scaling"] == "weak" and out["value"] > 0
    assert out["self_check"]["loop_equals_stand_alone_kernel_chain"] and out["roofline"]["bound"] == "hbm"
    assert out["config"]["correspondence_mean"] != [0.0, 0.0]


def test_bg_step_tool_raw_files_are_bit_exact(tmp_path):
    """--io npy: no codec anywhere, so the files the streaming tool writes (decoder processes -> pinned ring -> uploads -> pipelined
    loop -> downloads -> encoder processes) equal the in-process synchronous loop bit for bit."""
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, n = 270, 480, 7
    root = tmp_path / "data"
    src, dst = root / "src_img" / "3", root / "test_img" / "3"
    src.mkdir(parents=True)
    dst.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    save_img(clip.bg_always.numpy(), str(dst / "always_bg.jpg"))
    frames = [clip.frame(t) for t in range(n)]
    for t, f in enumerate(frames):
        np.save(src / f"{t:06d}.npy", f["frame"].numpy())
        np.save(dst / f"segmask_{t:06d}.npy", f["seg"].numpy())
        np.save(dst / f"bg_{t:06d}.npy", f["bg"].numpy())
    wpath = tmp_path / "unet.pth"
    torch.save({"state_dict": synthetic_state_dict(0)}, str(wpath))
    cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
    cfg["vmatting"]["model_path"] = str(wpath)
    json.dump(cfg, open(tmp_path / "bg.json", "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "bg_step.py"), "--cfg", str(tmp_path / "bg.json"),
                        "--video_id", "3", "--data_root", str(root), "--io", "npy", "--io_workers", "2"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    from video_unscreen_amd.pipeline import BgStepPipeline
    from video_unscreen_amd.utils.fileio import read_img
    pipe = BgStepPipeline(cfg, H, W, state_dict=synthetic_state_dict(0))
    bga = torch.from_numpy(read_img(str(dst / "always_bg.jpg"))).cuda()
    for t, f in enumerate(frames):
        out = pipe.step(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga)
        assert np.array_equal(np.load(dst / f"alphamask_{t:06d}.npy"), out["alpha"].cpu().numpy()), t
        assert np.array_equal(np.load(dst / f"fg_{t:06d}.npy"), out["fg"].cpu().numpy()), t


def test_rccl_single_rank_collectives():
    """The `nccl` (= RCCL) code path of the multi-GPU set-up, on the one GPU this box has: process-group init bound to the device,
    the bucketed weight broadcast, the background broadcast, the all-reduce of the overall-background bucket and the MAX of the
    elapsed time - so the driver's 8-GPU launch is not the first time any of it runs."""
    code = r"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from video_unscreen_amd.dist import _bucket_broadcast
from video_unscreen_amd.vmatting.net import synthetic_state_dict
sd = synthetic_state_dict(0)
ft = [v.float() for k, v in sd.items() if not k.endswith("num_batches_tracked")]
got = _bucket_broadcast(ft, 0, dev)
assert all(torch.equal(a.cpu(), b) for a, b in zip(got, ft))
bg = torch.arange(270 * 480 * 3, dtype=torch.uint8, device=dev).view(270, 480, 3)
dist.broadcast(bg, src=0)
acc = torch.ones(270 * 480 * 4, dtype=torch.float32, device=dev); dist.all_reduce(acc); assert float(acc.sum()) == 270 * 480 * 4
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 1.5
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group(); print("rccl ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_replace_tool_end_to_end(tmp_path, oracle_lib):
    """tools/replace/replace.py (BASELINE.json configs[4], the compositing half) on a small dataset on disk: the correspondence mean it
    prints equals the reference's comp_dx_dy flow (replace.py:28-41) on the oracle, and its composites equal the oracle chain
    (shift_fg, rescale_fg 1.2, float64 blend: replace.py:59-76) on the same decoded inputs up to the JPEG of the output."""
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import read_img, save_img
    H, W, n = 120, 200, 3
    root = tmp_path / "replace"
    src, tgt, bgd = root / "unscreen_img" / "test7", root / "unscreenbg_img" / "out7", root / "unscreen_img" / "bg"
    for d in (src, tgt, bgd):
        d.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    clip2 = SyntheticClip(H + 30, W + 40, n, config_id=3, device="cpu")
    save_img(clip2.bg_always.numpy(), str(bgd / "test5_case.jpg"))
    for t in range(n):
        f, g = clip.frame(t), clip2.frame(t)
        a = (f["alpha_true"].numpy() * 255).astype(np.uint8)
        save_img((f["frame"].numpy() * (a[..., None] / 255.)).astype(np.uint8), str(tgt / f"fg_{t:06d}.jpg"))
        save_img(np.stack([a] * 3, -1), str(tgt / f"alphamask_{t:06d}.jpg"))
        save_img(g["frame"].numpy(), str(src / f"frame_{t:06d}.jpg"))
        save_img(np.stack([(g["alpha_true"].numpy() * 255).astype(np.uint8)] * 3, -1), str(src / f"alphamask_{t:06d}.jpg"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replace", "replace.py"), "--src", "test7", "--data_root", str(root)],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # the reference's comp_dx_dy on the oracle
    dxs, dys = [], []
    for t in range(n):
        dm = read_img(str(tgt / f"alphamask_{t:06d}.jpg"))
        sm = oracle_lib.resize_linear(read_img(str(src / f"alphamask_{t:06d}.jpg")), dm.shape[0], dm.shape[1])
        dc, sc = oracle_lib.get_center(dm), oracle_lib.get_center(sm)
        dxs.append(sc[0] - dc[0])
        dys.append(sc[1] - dc[1])
    dx, dy = float(np.mean(dxs)), float(np.mean(dys))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("Correspondence mean")][0]
    got_dx, got_dy = (float(v) for v in line.split(":")[1].split())
    assert abs(got_dx - dx) < 1e-9 and abs(got_dy - dy) < 1e-9, (line, dx, dy)
    out = root / "merge_test_img" / "test7_out7"
    bg = oracle_lib.resize_linear(read_img(str(bgd / "test5_case.jpg")), H, W)
    for t in range(n):
        fg, m = read_img(str(tgt / f"fg_{t:06d}.jpg")), read_img(str(tgt / f"alphamask_{t:06d}.jpg"))
        fs = oracle_lib.rescale_fg(oracle_lib.shift_fg(fg, dx, dy), 1.2)
        ms = oracle_lib.rescale_fg(oracle_lib.shift_fg(m, dx, dy), 1.2)
        want = oracle_lib.blend(fs, ms, bg)
        got = read_img(str(out / f"res_{t:06d}.jpg"))
        assert got.shape == want.shape and np.abs(got.astype(int) - want.astype(int)).mean() < 1.5      # the output JPEG is lossy
        cmp_img = read_img(str(out / f"compare_{t:06d}.jpg"))
        assert cmp_img.shape == (H, 2 * W, 3)


def _tool_dataset(tmp_path, H, W, n, config_id):
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import save_img
    root = tmp_path / "data"
    src = root / "src_img" / "1"
    src.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=config_id, device="cpu")
    for t in range(n):
        save_img(clip.frame(t)["frame"].numpy(), str(src / f"frame_{t:06d}.jpg"))
    return root, src


def _synthetic_cfg(tmp_path, name, keys):
    cfg = json.load(open(os.path.join(ROOT, "configs", name)))
    for k in keys:
        cfg[k]["model_path"] = "synthetic"
    path = tmp_path / name
    path.write_text(json.dumps(cfg))
    return cfg, str(path)


@pytest.mark.parametrize("tool,cfg_name,keys,config_id,out_dir", [("green.py", "green.json", ("binseg", "vmatting"), 1, "test_img"),
                                                                 ("bg.py", "bg.json", ("binseg", "vmatting", "stm"), 2, "test_bg_img")])
def test_green_and_bg_tools_write_what_the_pipeline_gives(tmp_path, tool, cfg_name, keys, config_id, out_dir):
    """tools/unscreen/green.py and bg.py (reference tools/unscreen/green.py:70-138, bg.py:47-104) on a small dataset on disk: every JPEG the
    tool writes - through the encoder processes (utils/frameio.py: AsyncSaver) - is byte for byte the file save_img() writes from the
    tensors the same pipeline returns in this process on the same decoded frames."""
    import torch
    from video_unscreen_amd.utils.fileio import read_img, save_img
    H, W, n = 136, 240, 5
    root, src = _tool_dataset(tmp_path, H, W, n, config_id)
    cfg, cfg_path = _synthetic_cfg(tmp_path, cfg_name, keys)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", tool), "--cfg", cfg_path, "--data_root", str(root), "--io_workers", "3"] +
                       (["--cf_seed", "0"] if tool == "green.py" else []),
                       capture_output=True, text=True, cwd=ROOT, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if tool == "green.py":
        from video_unscreen_amd.pipeline_green import GreenPipeline
        pipe, names = GreenPipeline(cfg, H, W, device="cuda:0", cf_random_state=0), (("fg", "fg"), ("alphamask", "alpha"), ("bg", "bg"))
    else:
        from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
        pipe, names = BgOnlinePipeline(cfg, H, W, device="cuda:0"), (("segmask", "segmask"), ("bg", "bgimg"), ("alphamask", "alpha"), ("fg", "fg"))
    dst = root / out_dir / "1"
    written = 0
    for t in range(n):
        out = pipe.step(torch.from_numpy(read_img(str(src / f"frame_{t:06d}.jpg"))).cuda())
        for fname, key in names:
            path = dst / f"{fname}_{t:06d}.jpg"
            if key not in out:
                assert not path.exists()
                continue
            ref = tmp_path / "ref.jpg"
            save_img(out[key].cpu().numpy(), str(ref))
            assert path.read_bytes() == ref.read_bytes(), (tool, fname, t)
            written += 1
    assert written >= 3 * n


def test_replace_tool_more_ranks_than_frames(tmp_path):
    """a rank whose frame range is empty (WORLD_SIZE > number of frames) used to die on a missing file before the
    collectives, leaving the other ranks in all_reduce until the time-out.  Three ranks (sharing this box's GPU, gloo) over two frames of
    raw files write exactly the files of the one-rank run and print the same correspondence mean."""
    from video_unscreen_amd.synth import SyntheticClip
    H, W, n = 96, 160, 2
    root = tmp_path / "replace"
    src, tgt, bgd = root / "unscreen_img" / "test7", root / "unscreenbg_img" / "out7", root / "unscreen_img" / "bg"
    for d in (src, tgt, bgd):
        d.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    clip2 = SyntheticClip(H, W, n, config_id=3, device="cpu", n_objects=1)
    np.save(str(bgd / "test5_case.npy"), clip2.bg_always.numpy())
    from video_unscreen_amd.utils.fileio import save_img
    save_img(clip2.bg_always.numpy(), str(bgd / "test5_case.jpg"))
    for t in range(n):
        f, g = clip.frame(t), clip2.frame(t)
        a = (f["alpha_true"].numpy() * 255).astype(np.uint8)
        np.save(str(tgt / f"fg_{t:06d}.npy"), (f["frame"].numpy() * (a[..., None] / 255.)).astype(np.uint8))
        np.save(str(tgt / f"alphamask_{t:06d}.npy"), np.stack([a] * 3, -1))
        np.save(str(src / f"frame_{t:06d}.npy"), g["frame"].numpy())
        np.save(str(src / f"alphamask_{t:06d}.npy"), np.stack([(g["alpha_true"].numpy() * 255).astype(np.uint8)] * 3, -1))
    tool = [os.path.join(ROOT, "tools", "replace", "replace.py"), "--src", "test7", "--data_root", str(root), "--io", "npy", "--io_workers", "2"]
    out = root / "merge_test_img" / "test7_out7"
    r1 = subprocess.run([sys.executable] + tool, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    want = {p.name: np.load(str(p)) for p in sorted(out.iterdir())}
    assert len(want) == 2 * n
    for p in out.iterdir():
        p.unlink()
    env = dict(os.environ, USC_BENCH_BACKEND="gloo", USC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                         "--master-port", "29631"] + tool, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r3.returncode == 0, r3.stdout[-3000:] + r3.stderr[-3000:]
    mean = lambda r: sorted({ln for ln in r.stdout.splitlines() if ln.startswith("Correspondence mean")})      # noqa: E731
    assert mean(r3) == mean(r1) and len(mean(r1)) == 1
    got = {p.name: np.load(str(p)) for p in sorted(out.iterdir())}
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)
