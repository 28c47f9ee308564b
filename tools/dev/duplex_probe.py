"""PCIe duplex: 49.8 MB uploads and 33.2 MB downloads (one 3840 x 2160 frame group each way) on two copy streams, alone and together, with the
GPU otherwise idle and beside back-to-back matting forwards.   python tools/dev/duplex_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dev = torch.device("cuda", 0)
UP, DN, N = 49766400, 33177600, 8
d_up = [torch.empty(UP, dtype=torch.uint8, device=dev) for _ in range(N)]
h_up = [torch.empty(UP, dtype=torch.uint8).pin_memory() for _ in range(N)]
d_dn = [torch.empty(DN, dtype=torch.uint8, device=dev) for _ in range(N)]
h_dn = [torch.empty(DN, dtype=torch.uint8).pin_memory() for _ in range(N)]
s_up, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(up, dn, n=40, compute=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if up:
            with torch.cuda.stream(s_up):
                d_up[i % N].copy_(h_up[i % N], non_blocking=True)
        if dn:
            with torch.cuda.stream(s_dn):
                h_dn[i % N].copy_(d_dn[i % N], non_blocking=True)
        if compute is not None:
            compute(); compute()
    s_up.synchronize(); s_dn.synchronize()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return n / t, ((UP if up else 0) + (DN if dn else 0)) * n / t / 1e9, (time.perf_counter() - t0) / n * 1e3

for name, u, d in (("uploads alone", 1, 0), ("downloads alone", 0, 1), ("both directions", 1, 1)):
    for _ in range(2):
        fps, gbs, _ = run(u, d)
    print("%-18s idle GPU: %6.1f groups/s, %5.1f GB/s in total" % (name, fps, gbs), flush=True)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
for _ in range(10): net()
for name, u, d in (("no copies", 0, 0), ("uploads alone", 1, 0), ("downloads alone", 0, 1), ("both directions", 1, 1)):
    for _ in range(2):
        fps, gbs, ms = run(u, d, compute=net)
    print("%-18s beside 2 matting forwards per group: copies done at %6.1f groups/s (%5.1f GB/s), everything after %.3f ms per group" % (name, fps, gbs, ms), flush=True)
