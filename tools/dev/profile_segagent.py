"""Time SegAgent.forward_device at 1080p (12 crops) and 256x256 (18 crops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import SegAgent
from video_unscreen_amd.synth import SyntheticClip

agent = SegAgent(model_path=None, input_long_side=962)
for (H, W) in ((1080, 1920), (256, 256)):
    frame = SyntheticClip(H, W, 1, config_id=2, device="cuda").frame(0)["frame"]
    for _ in range(2):
        agent.forward_device(frame)
    torch.cuda.synchronize()
    t = time.time()
    n = 5
    for _ in range(n):
        m = agent.forward_device(frame)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    ncrops = len(agent.grid_crop_locations(*agent.get_target_size(H, W)))
    print(f"{H}x{W}: {dt*1e3:.1f} ms / frame, {ncrops} crops, {dt*1e3/ncrops:.2f} ms / crop, layers {agent.model.n_layers()}, fg {float((m>0).float().mean()):.3f}")
