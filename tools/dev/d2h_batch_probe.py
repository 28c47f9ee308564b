"""What does downloading alpha + fg cost the conv stream, per frame, as two copies / one packed copy / one packed copy per 4 frames?
80 matting forwards, downloads on their own stream behind a compute event.   python tools/dev/d2h_batch_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
HW = 1080 * 1920
d = torch.empty((8, 4 * HW), dtype=torch.uint8, device=dev)
h = torch.empty((8, 4 * HW), dtype=torch.uint8).pin_memory()
s2 = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
for _ in range(10): net()
torch.cuda.synchronize()


def timed(body, n=80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        net()
        body(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def two(i):
    ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
    with torch.cuda.stream(s2):
        h[i % 8, :HW].copy_(d[i % 8, :HW], non_blocking=True)
        h[i % 8, HW:].copy_(d[i % 8, HW:], non_blocking=True)

def one(i):
    ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
    with torch.cuda.stream(s2):
        h[i % 8].copy_(d[i % 8], non_blocking=True)

def every(k):
    def f(i):
        if i % k != k - 1:
            return
        ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
        j = (i % 8) - (k - 1)
        with torch.cuda.stream(s2):
            h[j:j + k].copy_(d[j:j + k], non_blocking=True)
    return f

print("forward alone                      %.3f ms" % timed(lambda i: None))
print("2 copies (2 + 6 MB) per forward    %.3f ms" % timed(two))
print("1 packed copy (8.3 MB) per forward %.3f ms" % timed(one))
print("1 copy of 2 frames every 2nd       %.3f ms" % timed(every(2)))
print("1 copy of 4 frames every 4th       %.3f ms" % timed(every(4)))
print("1 copy of 8 frames every 8th       %.3f ms" % timed(every(8)))
print("forward alone again                %.3f ms" % timed(lambda i: None))
