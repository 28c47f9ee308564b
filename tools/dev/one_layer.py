"""Run ONE conv layer of the matting net repeatedly (for PMC passes): python tools/dev/one_layer.py <index> [reps]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet, ConvDesc
idx = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
net = HipMattingNet(synthetic_state_dict(0), 544, 960)
net.x8.normal_(); net(); torch.cuda.synchronize()
lib = nv.load(); st = torch.cuda.current_stream().cuda_stream
arr = (ConvDesc * 1)(net._descs[idx])
for _ in range(reps):
    lib.usc_conv_sequence_f32(arr, 1, st)
torch.cuda.synchronize()
