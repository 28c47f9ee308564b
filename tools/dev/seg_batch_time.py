"""Parsing-net forward time by batch, for networks planned for 4 and for 12 frames per forward (all networks of one plan give the same
bits per frame): what would a short first group cost when the pipeline fills?   python tools/dev/seg_batch_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.synth import SyntheticClip
agent = HumanSegAgent(model_path=None)
clip = SyntheticClip(1080, 1920, 12, config_id=2, device="cuda")
frames = [clip.frame(t)["frame"] for t in range(12)]
for plan in (4, 12):
    agent.set_plan_batch(plan)
    for n in (1, 2, 3, 4, 6, 8, 12):
        if n > plan:
            continue
        net = agent.model_for(n)
        for _ in range(3):
            net.forward(None)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            net.forward(None)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 10
        print(f"planned for {plan:2d}, batch {n:2d}: {dt*1e3:6.2f} ms per forward, {dt*1e3/n:.2f} ms per frame", flush=True)
    agent._nets = {}
