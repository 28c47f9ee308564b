"""Raw rate of pinned host <-> device copies by hipMemcpyKind, nothing else on the device, and beside back-to-back matting forwards
(copies queued on their own stream without any dependency on the compute stream).   python tools/dev/copy_rate_probe.py"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int
dev = torch.device("cuda", 0)
HW = 1080 * 1920
d = torch.randint(0, 255, (8, 4 * HW), dtype=torch.uint8, device=dev)
h = torch.zeros((8, 4 * HW), dtype=torch.uint8).pin_memory()
s2 = torch.cuda.Stream(dev)


def rate(dst, src, kind, nbytes, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        rc = hip.hipMemcpyAsync(dst[i % 8].data_ptr(), src[i % 8].data_ptr(), nbytes, kind, s2.cuda_stream)
        assert rc == 0, rc
    t1 = time.perf_counter()
    s2.synchronize()
    t2 = time.perf_counter()
    return nbytes * n / (t2 - t0) / 1e9, (t1 - t0) / n * 1e6

for nbytes in (HW, 4 * HW):
    for name, dst, src, kind in (("D2H kind DeviceToHost", h, d, 2), ("D2H kind NoCU", h, d, 1024), ("H2D kind HostToDevice", d, h, 1), ("H2D kind NoCU", d, h, 1024)):
        for _ in range(2):
            r, issue = rate(dst, src, kind, nbytes)
        print("%-26s %5.1f MB: %6.1f GB/s, %6.1f us of host time per call" % (name, nbytes / 1e6, r, issue), flush=True)

from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
for _ in range(10): net()


def beside(kind, dst, src, n=80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        net()
        if kind is not None:
            hip.hipMemcpyAsync(dst[i % 8].data_ptr(), src[i % 8].data_ptr(), 4 * HW, kind, s2.cuda_stream)
    torch.cuda.current_stream().synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

for rep in range(2):
    for name, kind, dst, src in (("no copy", None, None, None), ("D2H DeviceToHost", 2, h, d), ("D2H NoCU", 1024, h, d), ("H2D HostToDevice", 1, d, h)):
        a, b = beside(kind, dst, src)
        print("80 forwards beside %-18s compute stream done after %.3f ms / forward, everything after %.3f" % (name, a, b), flush=True)
