"""Does the kind of page-locked host memory decide whether a device -> host copy runs as a blit kernel (stealing CUs from the conv
kernels) or on an SDMA engine?  One packed 8.3 MB D2H per matting forward into (a) torch.pin_memory() (hipHostMalloc default:
coherent), (b) hipHostRegister()ed ordinary memory, (c) hipHostMalloc(NonCoherent), (d) hipHostMalloc(Coherent | Mapped | Portable).
python tools/dev/d2h_memtype_probe.py"""
import ctypes as C
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
N = 4 * 1080 * 1920
d = torch.empty(N, dtype=torch.uint8, device=dev)
s2 = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]


def host_alloc(flags):
    p = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(p), N, flags)
    assert rc == 0, rc
    return p.value


def timed(body, n=60):
    for _ in range(5): net()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        net()
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def raw_copy(ptr):
    def f():
        ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
        rc = hip.hipMemcpyAsync(ptr, d.data_ptr(), N, 2, s2.cuda_stream)       # hipMemcpyDeviceToHost
        assert rc == 0, rc
    return f


h_torch = torch.empty(N, dtype=torch.uint8).pin_memory()
plain = np.empty(N + 4096, np.uint8)
base = (plain.ctypes.data + 4095) & ~4095
assert int(torch.cuda.cudart().cudaHostRegister(base, N, 0)) == 0
cases = [("torch.pin_memory()", h_torch.data_ptr()), ("hipHostRegister()ed ordinary memory", base)]
for name, fl in (("hipHostMalloc(Default)", 0x0), ("hipHostMalloc(NonCoherent)", 0x80000000), ("hipHostMalloc(Mapped | Portable)", 0x3)):
    cases.append((name, host_alloc(fl)))
for rnd in range(3):
    order = cases if rnd % 2 == 0 else cases[::-1]
    print("round %d: forward alone %.3f ms" % (rnd, timed(lambda: None)))
    for name, ptr in order:
        print("   D2H -> %-40s %.3f ms" % (name, timed(raw_copy(ptr))))
