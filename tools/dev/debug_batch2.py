import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.synth import SyntheticClip
agent = HumanSegAgent(model_path=None)
clip = SyntheticClip(270, 480, 3, config_id=2, device="cuda")
frames = [clip.frame(t)["frame"] for t in range(3)]
single = []
for f in frames:
    m, lg = agent.forward_device(f, return_logits=True)
    single.append((m.clone(), lg.clone(), agent.net.x4.clone()))
for n in (2, 3):
    masks = agent.forward_device_batch(frames[:n])
    net = agent._batch_nets[n]
    for b in range(n):
        print(n, b, "x4 equal", torch.equal(net.image(net.x4, b), single[b][2]), "logits equal", torch.equal(net.image(net.logits32, b), single[b][1]),
              "maxdiff", (net.image(net.logits32, b) - single[b][1]).abs().max().item(), "mask equal", torch.equal(masks[b], single[b][0]))
