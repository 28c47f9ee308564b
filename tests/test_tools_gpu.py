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


def test_bench_two_ranks_rehearsal():
    """bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), with the
    two ranks sharing this box's single GPU and gloo instead of RCCL: exercises the broadcast of weights / background, the
    frame sharding, the barrier-bracketed timing, the MAX over ranks and the single JSON line."""
    env = dict(os.environ, USC_BENCH_BACKEND="gloo", USC_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-table"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                         # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 2 * 6 / (out["ms_per_step"] * 6 / 1e3)) < 1e-2 * out["value"]
