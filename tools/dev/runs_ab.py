"""USC_CONV_LAUNCH_RUNS against the per-layer launches on the matting net (544 x 960): (1) bit-identity of every layer output of the
runs and of the prediction over repeated forwards (also beside a second stream that keeps the GPU unevenly busy), poll time-outs;
(2) HIP-event time per forward in both modes, and of the run's layers alone (layers [lo, hi) as a sub-network).
   python tools/dev/runs_ab.py [forwards]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from video_unscreen_amd import _native as nv  # noqa: E402
from video_unscreen_amd.vmatting.net import synthetic_state_dict  # noqa: E402
from video_unscreen_amd.vmatting.net_hip import ConvDesc, HipMattingNet  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
if os.environ.get("USC_LIB"):
    nv.LIB_PATH = os.path.abspath(os.environ["USC_LIB"])
lib = nv.load()
net = HipMattingNet(synthetic_state_dict(0), 544, 960)
net.x8.normal_()


def info(handle):
    a, b, c = C.c_int(), C.c_int(), C.c_uint()
    assert lib.usc_conv_graph_run_info(handle, C.byref(a), C.byref(b), C.byref(c)) == 0
    return a.value, b.value, c.value


def act_bufs():
    """The activation tensors (every float tensor a descriptor points to that is not a weight: 2-D+ buffers made by _buf)."""
    return [t for t in net._keep if t.dim() == 3]


def timed(n):
    for _ in range(5):
        net()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        net()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("runs in the matting net: %d covering %d layers" % info(net._graph)[:2])
net.set_launch_mode("stream")
net()
ref = net.pred.clone()
bufs_ref = [b.clone() for b in act_bufs()]
t_stream = timed(N)
net.set_launch_mode("runs")
bad = 0
for i in range(N):
    net()
    bad += int(not torch.equal(net.pred, ref))
torch.cuda.synchronize()
bad_bufs = sum(int(not torch.equal(a, b)) for a, b in zip(act_bufs(), bufs_ref))
print("runs mode, idle GPU: %d of %d forwards differ from the per-layer launches; intermediate tensors differing after the last: %d of %d; poll time-outs %d"
      % (bad, N, bad_bufs, len(bufs_ref), info(net._graph)[2]))
# beside another stream that keeps a varying number of CUs busy
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
bad = 0
for i in range(N):
    with torch.cuda.stream(side):
        for _ in range(1 + i % 3):
            junk = (junk @ junk).clamp_(-1, 1)
    net()
    bad += int(not torch.equal(net.pred, ref))
torch.cuda.synchronize()
print("runs mode, beside a busy side stream: %d of %d forwards differ; poll time-outs %d" % (bad, N, info(net._graph)[2]))
t_runs = timed(N)
net.set_launch_mode("stream")
t_stream2 = timed(N)
print("us per forward: per-layer launches %.1f / %.1f, runs mode %.1f  (difference %.1f)" % (t_stream, t_stream2, t_runs, min(t_stream, t_stream2) - t_runs))
