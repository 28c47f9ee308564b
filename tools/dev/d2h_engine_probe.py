"""When does a pinned device -> host copy run on the SDMA engine and when as a blit kernel on the CUs?  40 matting forwards with one
6 MB D2H per forward issued in different ways; the forward alone is the reference time (a blit kernel steals CUs from the conv
kernels and shows up as extra time per forward).   python tools/dev/d2h_engine_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
N = 6 * 1024 * 1024
h_up = torch.empty(N, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(N, dtype=torch.uint8).pin_memory()
h_tiny = torch.zeros(64, dtype=torch.uint8).pin_memory()
d_up, d_dn, d_tiny = torch.empty(N, dtype=torch.uint8, device=dev), torch.empty(N, dtype=torch.uint8, device=dev), torch.empty(64, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
for _ in range(10): net()
torch.cuda.synchronize()


def timed(body, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        net()
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def d2h_own_stream():
    with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)

def h2d_then_d2h_one_stream():
    with torch.cuda.stream(s1):
        d_up.copy_(h_up, non_blocking=True)
        h_dn.copy_(d_dn, non_blocking=True)

def d2h_after_event_wait():
    ev = torch.cuda.Event(); ev.record(main)
    s2.wait_event(ev)
    with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)

def tiny_h2d_then_d2h_after_event_wait():
    ev = torch.cuda.Event(); ev.record(main)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        d_tiny.copy_(h_tiny, non_blocking=True)
        h_dn.copy_(d_dn, non_blocking=True)

def h2d_wait_d2h_one_stream():
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(s1):
        d_up.copy_(h_up, non_blocking=True)
    s1.wait_event(ev)
    with torch.cuda.stream(s1):
        h_dn.copy_(d_dn, non_blocking=True)

def two_d2h_own_stream():
    with torch.cuda.stream(s2):
        h_dn.copy_(d_dn, non_blocking=True)
        h_up.copy_(d_up, non_blocking=True)

def d2h_on_main_stream():
    h_dn.copy_(d_dn, non_blocking=True)

print("forward alone                                  %.3f ms" % timed(lambda: None))
for name, fn in (("D2H on its own stream", d2h_own_stream), ("H2D then D2H, one stream", h2d_then_d2h_one_stream),
                 ("event wait, D2H (own stream)", d2h_after_event_wait), ("event wait, 64-byte H2D, D2H (own stream)", tiny_h2d_then_d2h_after_event_wait),
                 ("H2D, event wait, D2H (one stream)", h2d_wait_d2h_one_stream), ("two D2H back to back, own stream", two_d2h_own_stream),
                 ("D2H on the compute stream", d2h_on_main_stream)):
    print("%-46s %.3f ms" % (name, timed(fn)))
print("forward alone again                            %.3f ms" % timed(lambda: None))
