"""GPU tests of the overall-background stage (reference tools/unscreen/bg_offline.py:99-130, disabled there with `if 0:`):
the HIP accumulators / finalisation bit-exact against the oracle, the hole fill against the oracle's direct solve, and
tools/unscreen/bg_step.py --stages 123 (all three stages of bg_offline) on a dataset on disk."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clip(n, H, W, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:H, :W]
    back = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 255 // (H + W))], -1).astype(np.uint8)
    back = np.clip(back.astype(int) + rng.integers(-6, 7, back.shape), 0, 255).astype(np.uint8)
    frames, masks = [], []
    for t in range(n):
        cx = W // 4 + (W // 2) * t // max(n - 1, 1)
        m = (((xx - cx) / (W / 9.)) ** 2 + ((yy - H // 2) / (H / 3.5)) ** 2 < 1).astype(np.uint8) * 255
        m[H - H // 6:, :W // 8] = 255                      # a corner no frame ever shows: the always-masked hole
        m[::7, ::5] = np.where(m[::7, ::5] > 0, rng.integers(200, 256, m[::7, ::5].shape), 0)     # soft values around 250
        f = back.copy()
        f[m > 0] = rng.integers(0, 256, (int((m > 0).sum()), 3), dtype=np.uint8)
        frames.append(f)
        masks.append(m)
    return back, frames, masks


@pytest.mark.parametrize("shape", [(54, 96), (135, 241)])
def test_overall_background_matches_oracle(oracle_lib, shape):
    from video_unscreen_amd.bgestimate import OverallBackground
    H, W = shape
    back, frames, masks = _clip(16, H, W)
    ob = OverallBackground(H, W)
    raw = np.zeros((H, W, 3), np.float32)
    cont = np.zeros((H, W, 3), np.float32)
    for f, m in zip(frames, masks):
        ob.add(torch.from_numpy(f).cuda(), torch.from_numpy(m).cuda())
        oracle_lib.overall_bg_accumulate(f, m, raw, cont)
    assert np.array_equal(ob.raw.cpu().numpy(), raw) and np.array_equal(ob.cont.cpu().numpy(), cont[..., 0])
    out = ob.finalize()
    bg_always, hole = oracle_lib.overall_bg_finalize(raw, cont)
    assert np.array_equal(out["bg_always"].cpu().numpy(), bg_always)
    assert np.array_equal(out["hole"].cpu().numpy(), hole) and 0 < (hole > 0).mean() < 0.5
    want = np.stack([oracle_lib.regionfill(bg_always[..., c], hole) for c in range(3)], -1).astype(np.uint8)
    got = out["bg"].cpu().numpy()
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # where the clip showed the background often enough, the mean recovers it (hard mask values only: 1..254 count as
    # visible in the sum, 250..254 not even in the count - the reference's arithmetic, kept)
    dil = [oracle_lib.morph(m, 3, 2, "dilate") for m in masks]
    seen = (hole == 0) & np.all([(d == 0) | (d == 255) for d in dil], 0)
    assert seen.mean() > 0.3
    assert np.abs(got[seen].astype(int) - back[seen].astype(int)).max() <= 1


def test_overall_background_shards_sum_exactly():
    """Two half-clips accumulated separately and added (what the all-reduce does) == one pass over the clip."""
    from video_unscreen_amd.bgestimate import OverallBackground
    H, W = 54, 96
    _, frames, masks = _clip(12, H, W, seed=9)
    whole, a, b = OverallBackground(H, W), OverallBackground(H, W), OverallBackground(H, W)
    for t, (f, m) in enumerate(zip(frames, masks)):
        fd, md = torch.from_numpy(f).cuda(), torch.from_numpy(m).cuda()
        whole.add(fd, md)
        (a if t < 6 else b).add(fd, md)
    assert torch.equal(a.acc + b.acc, whole.acc)


def test_bg_step_tool_all_stages(tmp_path):
    from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
    from video_unscreen_amd.stm.net import stm_synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import read_img, save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, n = 135, 240, 3
    root = tmp_path / "data"
    src = root / "src_img" / "9"
    src.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    for t in range(n):
        save_img(clip.frame(t)["frame"].numpy(), str(src / f"{t:06d}.jpg"))
    torch.save({"state_dict": synthetic_state_dict(0)}, str(tmp_path / "unet.pth"))
    torch.save(schp_synthetic_state_dict(0), str(tmp_path / "schp.pth"))
    torch.save(stm_synthetic_state_dict(0), str(tmp_path / "stm.pth"))
    cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
    cfg["vmatting"]["model_path"] = str(tmp_path / "unet.pth")
    cfg["binseg"]["model_path"] = str(tmp_path / "schp.pth")
    cfg["stm"]["model_path"] = str(tmp_path / "stm.pth")
    cfg["stm"]["input_long_side"] = 240
    json.dump(cfg, open(tmp_path / "bg.json", "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "bg_step.py"), "--cfg", str(tmp_path / "bg.json"),
                        "--video_id", "9", "--data_root", str(root), "--stages", "123"], capture_output=True, text=True, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dst = root / "test_img" / "9"
    assert read_img(str(dst / "always_bg.jpg")).shape == (H, W, 3)
    for t in range(n):
        for name in ("segmask", "bg", "alphamask", "fg"):
            assert read_img(str(dst / f"{name}_{t:06d}.jpg")).shape == (H, W, 3)
    assert "stage 1" in r.stdout and "stage 2" in r.stdout
