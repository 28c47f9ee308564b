"""How much do the two conv streams of the frame loop gain from being resident TOGETHER?  A Winograd workgroup takes its CU's whole
register file, so a Winograd layer of one network and any layer of the other take turns; two direct layers (86 - 92 registers per wave
since round 5) can share a SIMD.  Probe: 12 matting forwards on one stream beside one 12-frame parsing forward on another, with the
matting net / both nets planned WITHOUT Winograd (usc_conv_force_winograd(0) while the plans are made) - slower alone, but able to overlap.
Prints: each network alone, the serial sum, both streams at once.   python tools/dev/coresidency_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict

lib = nv.load()
s2 = torch.cuda.Stream()
msd, ssd = synthetic_state_dict(0), schp_synthetic_state_dict(0)


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


for tag, m_direct, s_direct in (("both with Winograd (shipped)", False, False), ("matting net direct only", True, False), ("both nets direct only", True, True)):
    lib.usc_conv_force_winograd(0 if m_direct else -1)
    unet = HipMattingNet(msd, 544, 960)
    lib.usc_conv_force_winograd(0 if s_direct else -1)
    schp = HipSchpNet(ssd, 473, 473, batch=12, plan_batch=12)
    lib.usc_conv_force_winograd(-1)
    unet.x8.normal_(); schp.x4.normal_()
    for _ in range(100):
        unet()
    torch.cuda.synchronize()

    def m12():
        for _ in range(12):
            unet()

    def both():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            schp()
        m12()
        torch.cuda.current_stream().wait_stream(s2)

    tm, ts, tb = timed(m12), timed(lambda: schp()), timed(both)
    print(f"{tag:32s} 12 matting forwards {tm:7.3f} ms   parsing forward (12 frames) {ts:7.3f} ms   sum {tm + ts:7.3f}   two streams at once {tb:7.3f} ms  ({100 * (1 - tb / (tm + ts)):.1f} % under the sum)", flush=True)
    unet.close(); schp.close()
    del unet, schp
    torch.cuda.empty_cache()
