"""Does hipMemcpyAsync(kind = hipMemcpyDeviceToDeviceNoCU) move a pinned device -> host copy off the CUs (SDMA engine instead of the
__amd_rocclr_copyBuffer blit kernel), does the data arrive, and what does a packed 8.3 MB download per matting forward then cost?
python tools/dev/d2h_nocu_probe.py   (under rocprofv3 --kernel-trace --stats to count the blit kernels)"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
HW = 1080 * 1920
d = torch.randint(0, 255, (8, 4 * HW), dtype=torch.uint8, device=dev)
h = torch.zeros((8, 4 * HW), dtype=torch.uint8).pin_memory()
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


def raw(kind):
    def f(i):
        ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
        rc = hip.hipMemcpyAsync(h[i % 8].data_ptr(), d[i % 8].data_ptr(), 4 * HW, kind, s2.cuda_stream)
        assert rc == 0, rc
    return f


def check(kind):
    h.zero_(); torch.cuda.synchronize()
    for i in range(8): raw(kind)(i)
    torch.cuda.synchronize()
    return bool((h == d.cpu()).all())

print("forward alone                          %.3f ms" % timed(lambda i: None))
print("packed copy, kind DeviceToHost         %.3f ms   data ok %s" % (timed(raw(2)), check(2)))
print("packed copy, kind DeviceToDeviceNoCU   %.3f ms   data ok %s" % (timed(raw(1024)), check(1024)))
print("packed copy, kind Default              %.3f ms   data ok %s" % (timed(raw(4)), check(4)))
print("forward alone again                    %.3f ms" % timed(lambda i: None))
# host -> device for completeness
def up(kind):
    def f(i):
        rc = hip.hipMemcpyAsync(d[i % 8].data_ptr(), h[i % 8].data_ptr(), 4 * HW, kind, s2.cuda_stream)
        assert rc == 0, rc
    return f
print("packed upload, kind HostToDevice       %.3f ms" % timed(up(1)))
print("packed upload, kind DeviceToDeviceNoCU %.3f ms" % timed(up(1024)))
