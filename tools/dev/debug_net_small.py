import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import torch
from gen_golden import unet_input
from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
from video_unscreen_amd.vmatting import net_hip
sd = synthetic_state_dict(0)
for (H, W) in ((64, 96), (544, 960)):
    x = unet_input(H, W, 5)
    want = MattingNet(sd, device="cuda").forward(x.cuda()).cpu()
    x8 = torch.cat([x, torch.zeros(1, 1, H, W)], 1)[0].permute(1, 2, 0).contiguous().cuda()
    for mode in ("slab", "noslab"):
        net = net_hip.HipMattingNet(sd, H, W)
        if mode == "noslab":
            for d in net._descs:
                d.slab = None; d.slab_bytes = 0
            net._arr = (net_hip.ConvDesc * len(net._descs))(*net._descs)
        got = net(x8).cpu()
        print(H, W, mode, "maxdiff", float((got - want).abs().max()))
