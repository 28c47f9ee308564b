"""Parity report at BASELINE.json's full frame sizes (run on the GPU box: `python tests/parity_report.py > profiles/r01/parity_report.json`).

bg_step frames through the HIP path (human-parsing segmentation on every frame, then the stage-3 loop) against the oracle
chain on the same inputs - C oracle for the pixel stages, the folded network plans on torch-CPU float32 for the two nets -
reporting what SURVEY.md 8d asks for: mismatch counts of the index masks (segmentation mask, trimap), max |alpha_float
difference|, uint8 alpha mismatch count and max LSB, foreground mismatches given the same alpha.  Test infrastructure: it
uses oracle/ as the checker."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import __graft_entry__ as ge
import oracle
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import SchpNetTorch
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip


def run(H, W, nframes):
    cfg = default_config()
    clip = SyntheticClip(H, W, nframes, config_id=2, device="cpu")
    agent = HumanSegAgent(model_path=None)
    pipe = BgStepPipeline(cfg, H, W, seg_agent=agent)
    seg_cpu = SchpNetTorch(agent.state_dict)
    bga = clip.bg_always
    rows = []
    alpha_pre = None
    for t in range(nframes):
        f = clip.frame(t)
        out = pipe.step(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga.cuda(), want_float=True)
        torch.cuda.synchronize()
        want = ge.oracle_frame(oracle, cfg, pipe, f["frame"].numpy(), f["seg"].numpy(), f["bg"].numpy(), bga.numpy(), alpha_pre,
                               seg_net_cpu=seg_cpu)
        seg_dev = out["seg"].cpu().numpy()
        seg_flips = int((seg_dev != want["seg"]).sum())
        # downstream stages are compared on the SAME segmentation mask: re-run the oracle chain from the device mask
        want2 = ge.oracle_frame(oracle, cfg, pipe, f["frame"].numpy(), seg_dev, f["bg"].numpy(), bga.numpy(), alpha_pre)
        a8, af = out["alpha"].cpu().numpy(), out["alpha_f32"].cpu().numpy()
        d8 = np.abs(a8.astype(int) - want2["alpha"].astype(int))
        fg_same, _ = oracle.get_fg(f["frame"].numpy(), a8, want2["bg"], paste_thr=1)
        rows.append({
            "frame": t,
            "seg_mask_mismatch_px_vs_cpu_net": seg_flips,
            "gate_mismatch_px": int((out["gate"].cpu().numpy() != want2["alpha_gate"]).sum()),
            "object_filter_mismatch_px": int((out["alphaor"].cpu().numpy() != want2["alphaor"]).sum()),
            "trimap_mismatch_px": int((out["trimap"].cpu().numpy() != want2["trimap"]).sum()),
            "alpha_float_max_abs_diff": float(np.abs(af - want2["alpha_f32"]).max()),
            "alpha_u8_mismatch_px": int((d8 > 0).sum()), "alpha_u8_max_lsb": int(d8.max()),
            "fg_mismatch_px_given_same_alpha": int((out["fg"].cpu().numpy() != fg_same).any(-1).sum()),
        })
        alpha_pre = a8                                  # both chains continue from the device alpha
        pipe.alpha_pre = out["alpha"]
    return {"size": [H, W], "pixels": H * W, "frames": rows}


if __name__ == "__main__":
    torch.manual_seed(0)
    rep = {"what": "bg_step HIP path vs oracle chain (C oracle + folded nets on torch-CPU f32), synthetic clip, synthetic weights",
           "tolerance": "index masks bit-exact given the same inputs; alpha within 1e-3 (float) of the CPU chain",
           "runs": [run(1080, 1920, 3), run(2160, 3840, 2)]}
    print(json.dumps(rep, indent=1))
