"""Human-parsing net: ms per frame of one forward as a function of the frames per forward (plan_batch = batch).  Round 5: tile counts of
the 30 x 30 layers are M = 900 b pixels, so the rounds of workgroups - and the per-frame cost - are not monotonic in b."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict

sd = schp_synthetic_state_dict(0)
for b in [int(v) for v in (sys.argv[1:] or "4 6 8 9 10 11 12 13 14 15 16 18 20 24".split())]:
    net = HipSchpNet(sd, 473, 473, batch=b, plan_batch=b)
    net.x4.normal_()
    for _ in range(3):
        net()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            net()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    print(f"batch {b:3d}: {best:7.3f} ms per forward, {best / b:6.4f} ms per frame", flush=True)
    net.close()
    del net
    torch.cuda.empty_cache()
