"""GPU tests of the bg online mode (reference tools/unscreen/bg.py:45-104): BgOnlinePipeline stage by stage against the
oracle applied to the previous stage's device output, and the tools/unscreen/bg.py entry point on a dataset on disk."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BG_CFG = {
    "fg_exist_thr": 0.001, "bg_mask": {"thr": 25, "fusion_weight": 0.1},
    "binseg": {"model_path": "synthetic", "input_long_side": 962, "stride_ratio": 0.5, "flip": True},
    "stm": {"model_path": "synthetic", "input_long_side": 480, "memory_step": 2, "memory_capacity": 10},
    "objectremoval": {"score_map_center": {"landscape": [0.5, 0.5], "portrait": [0.6, 0.5]}, "saliency_thr": 0.00001,
                      "consensus_thr": 0.5},
    "trimap": {"input_long_side": 960, "kernelsize": 3, "iters": 5, "color_winsize": [10, 100, 180]},
    "vmatting": {"model_path": "synthetic", "input_long_side": 960},
}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


class _FixedSeg:
    def __init__(self, mask):
        self.mask = mask

    def forward_device(self, frame):
        return dev(self.mask)


def _close(got, want, frac=1e-3):
    d = np.abs(got.astype(int) - want.astype(int))
    return d.max() <= 1 and (d > 0).mean() < frac


@pytest.mark.parametrize("H,W,n,real", [(270, 480, 3, False), (1080, 1920, 2, True)])
def test_bg_online_pipeline_stage_by_stage(oracle_lib, H, W, n, real):
    """Every stage's device output against the oracle applied to the previous stage's device output (reference tools/unscreen/bg.py:47-104).
    270 x 480: three frames behind a known coarse mask.  1920 x 1080 (BASELINE.json configs[2], the whole loop at its full size): the
    agents the config names - frame 0 untracked through HumanSegAgent (SCHP on the device against the folded plan on torch-CPU: argmax
    ties only), frame 1 tracked through the space-time memory network at long side 960 (against oracle/chains.py:stm_track) - then
    object filter -> trimap -> matting -> get_bg -> Laplace fill -> difference gate -> object filter -> trimap -> matting -> get_fg."""
    from oracle import chains
    from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    cfg = copy.deepcopy(BG_CFG)
    if real:
        cfg["stm"]["input_long_side"] = 960                          # configs/bg.json
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    frames = [clip.frame(t) for t in range(n)]
    pipe = BgOnlinePipeline(cfg, H, W, segagent=None if real else _FixedSeg(frames[0]["seg"].numpy()))
    cpu_net = MattingNet(synthetic_state_dict(0), device="cpu", dtype=torch.float32)
    in_h, in_w = oracle_lib.vmat_input_size(H, W, 960)
    if real:
        from video_unscreen_amd.binseg.schp_net import SchpNetTorch
        from video_unscreen_amd.stm.net import StmNetTorch
        schp_cpu, stm_cpu = SchpNetTorch(pipe.segagent.state_dict), StmNetTorch(pipe.stmagent.state_dict)

    def matting(frame, alpha_pre, tri):
        x7 = oracle_lib.vmat_pre(frame, alpha_pre, tri, in_h, in_w)
        return oracle_lib.vmat_post(cpu_net(torch.from_numpy(x7)[None]).numpy()[0, 0], tri)

    alpha_pre = np.zeros((H, W), np.uint8)
    tracking = False
    for t, f in enumerate(frames):
        frame = f["frame"].numpy()
        out = pipe.step(dev(frame))
        assert out["tracking"] == tracking
        seg = host(out["segmask"])
        if not tracking and not real:
            assert np.array_equal(seg, f["seg"].numpy())
        elif not tracking:                               # HumanSegAgent: every differing pixel is an argmax tie (tests/test_gpu_seg.py)
            x = torch.from_numpy(oracle_lib.humanseg_pre(frame))[None]
            want = oracle_lib.humanseg_post(schp_cpu(x)[0].numpy(), H, W)
            assert (seg != want).sum() <= 64 and 0.02 < (want > 0).mean() < 0.98
        else:                                            # the STM step itself is covered by tests/test_gpu_stm.py
            seed = np.where(alpha_pre >= 128, 255, alpha_pre).astype(np.uint8)
            again = pipe.stmagent.forward_device([dev(frames[t - 1]["frame"].numpy()), dev(frame)], dev(seed))[-1]
            assert np.array_equal(seg, host(again)) and 0.01 < (seg > 0).mean() < 0.99
            if real:                                     # and against the CPU chain (the mask may flip where two probabilities tie)
                want = chains.stm_track(oracle_lib, stm_cpu, frames[t - 1]["frame"].numpy(), frame, seed, 960)
                assert (seg != want).mean() < 2e-3, float((seg != want).mean())
        want = oracle_lib.remove_invalid_objects(cfg, seg.copy())
        assert np.array_equal(host(out["alphaor1"]), want)
        tri1 = oracle_lib.trimap(want, 960, 3, 5)
        assert np.array_equal(host(out["trimap1"]), tri1)
        a1 = host(out["alpha1"])
        assert _close(a1, matting(frame, alpha_pre, tri1))
        assert np.array_equal(host(out["bg_holes"]), oracle_lib.get_bg(a1, frame))
        hole = oracle_lib.morph(np.where(a1 > 128, 255, 0).astype(np.uint8), 3, 2, "dilate")
        assert np.array_equal(host(out["hole"]), hole)
        bgh = oracle_lib.get_bg(a1, frame)
        exact = np.stack([oracle_lib.regionfill(bgh[..., c], hole) for c in range(3)], -1)
        bgimg = host(out["bgimg"])
        assert np.abs(bgimg.astype(int) - exact.astype(np.uint8).astype(int)).max() <= 1
        _, ensm = oracle_lib.bgdiff_mask(frame, bgimg, None, a1, 1.0, cfg["bg_mask"]["thr"])
        assert np.array_equal(host(out["alpha_ensm"]), ensm)
        want = oracle_lib.remove_invalid_objects(cfg, ensm.copy())
        assert np.array_equal(host(out["alphaor2"]), want)
        tri2 = oracle_lib.trimap(want, 960, 3, 5)
        assert np.array_equal(host(out["trimap2"]), tri2)
        a2 = host(out["alpha"])
        assert _close(a2, matting(frame, alpha_pre, tri2))
        want_fg, want_bg = oracle_lib.get_fg(frame, a2, bgimg, paste_thr=1)
        assert np.array_equal(host(out["fg"]), want_fg) and np.array_equal(host(out["bg"]), want_bg)
        alpha_pre = a2
        tracking = bool((a2 >= 128).sum() > cfg["fg_exist_thr"] * H * W)
        assert pipe.tracking_flag == tracking
    assert tracking


def test_bg_tool_end_to_end(tmp_path):
    from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
    from video_unscreen_amd.stm.net import stm_synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import read_img, save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    H, W, n = 270, 480, 3
    root = tmp_path / "data"
    src = root / "src_img" / "5"
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
    cfg["stm"]["input_long_side"] = 480
    json.dump(cfg, open(tmp_path / "bg.json", "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "bg.py"), "--cfg", str(tmp_path / "bg.json"),
                        "--video_id", "5", "--data_root", str(root)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    dst = root / "test_bg_img" / "5"
    for t in range(n):
        assert read_img(str(dst / f"segmask_{t:06d}.jpg")).shape == (H, W, 3)
    assert "tracked" in r.stdout
