"""Are pinned-host <-> device copies executed by the SDMA engines or by blit kernels on the CUs?  Times 40 matting forwards alone and
beside a stream of 6 MB H2D + D2H copies; run under `rocprofv3 --kernel-trace --stats` the copy kernels (if any) show up as
__amd_rocclr_copyBuffer.   python tools/dev/copy_engine_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
print({k: v for k, v in os.environ.items() if any(s in k for s in ("HSA", "SDMA", "ROC", "HIP", "GPU_"))})
import __graft_entry__ as ge
ge.build_native()
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
h = torch.empty(6 * 1024 * 1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device=dev)
h2 = torch.empty_like(h).pin_memory()
up, down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for _ in range(10): net()
torch.cuda.synchronize()
d2 = torch.empty_like(d)
def run(h2d, d2h, d2d=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        net()
        if h2d:
            with torch.cuda.stream(up): d.copy_(h, non_blocking=True)
        if d2h:
            with torch.cuda.stream(down): h2.copy_(d, non_blocking=True)
        if d2d:
            with torch.cuda.stream(down): d2.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
def run_one_stream():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        net()
        with torch.cuda.stream(up):
            d.copy_(h, non_blocking=True)
            h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
def run_two_streams_distinct_buffers():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        net()
        with torch.cuda.stream(up): d.copy_(h, non_blocking=True)
        with torch.cuda.stream(down): h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
def host_issue_time():
    torch.cuda.synchronize(); t0 = time.perf_counter(); tc = 0.0
    for i in range(40):
        net()
        t1 = time.perf_counter()
        with torch.cuda.stream(up): d.copy_(h, non_blocking=True)
        with torch.cuda.stream(down): h2.copy_(d2, non_blocking=True)
        tc += time.perf_counter() - t1
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t2 - t0) / 40 * 1e3, tc / 40 * 1e3, (t3 - t0) / 40 * 1e3
print("host: loop issued after %.3f ms/iter (of which the two copy calls %.3f ms), GPU done after %.3f ms/iter" % host_issue_time())
print("H2D + D2H on ONE stream: %.3f ms; on two streams with distinct device buffers: %.3f ms" % (run_one_stream(), run_two_streams_distinct_buffers()))
print("forward alone %.3f ms; with one 6 MB H2D + one 6 MB D2H per forward: %.3f ms; H2D only %.3f; D2H only %.3f; a 6 MB D2D beside it %.3f"
      % (run(False, False), run(True, True), run(True, False), run(False, True), run(False, False, True)))
