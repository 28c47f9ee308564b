import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.synth import SyntheticClip
agent = HumanSegAgent(model_path=None)
clip = SyntheticClip(1080, 1920, 4, config_id=2, device="cuda")
frames = [clip.frame(t)["frame"] for t in range(4)]
for n in (1, 2, 3, 4):
    for _ in range(3):
        agent.forward_device_batch(frames[:n])
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        agent.forward_device_batch(frames[:n])
    torch.cuda.synchronize()
    dt = (time.time() - t) / 20
    print(f"batch {n}: {dt*1e3:.2f} ms per forward, {dt*1e3/n:.2f} ms per frame", flush=True)
