"""Do the matting net (main stream) and the batch-4 human-parsing net (second stream) overlap on the device?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet

unet = HipMattingNet(synthetic_state_dict(0), 544, 960)
schp = HipSchpNet(schp_synthetic_state_dict(0), 473, 473, batch=4, plan_batch=4)
unet.x8.normal_(); schp.x4.normal_()
s2 = torch.cuda.Stream()


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def both():
    with torch.cuda.stream(s2):
        schp()
    for _ in range(4):
        unet()


def serial():
    schp()
    for _ in range(4):
        unet()


def schp_s2():
    with torch.cuda.stream(s2):
        schp()


print("4 x unet      %.3f ms" % timed(lambda: [unet() for _ in range(4)]))
print("schp (b4)     %.3f ms" % timed(lambda: schp()))
print("schp on s2    %.3f ms" % timed(schp_s2))
print("serial        %.3f ms" % timed(serial))
print("two streams   %.3f ms" % timed(both))
