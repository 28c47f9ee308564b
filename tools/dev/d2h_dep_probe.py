"""Which part of 'download behind a compute event' costs the conv stream: the copy, or the event?  80 matting forwards; after each one a
packed 8.3 MB device -> host copy on a second stream, in one of these forms (argv[1]):
  none      no copy                          indep     copy with no dependency on the compute stream
  event     record on compute + wait on the copy stream (what hostio does), kind DeviceToHost
  event_nocu  same with kind DeviceToDeviceNoCU
  record_only / wait_only   just the event halves, no copy
  hostwait  a host thread waits for the event, then issues the copy on a stream that never sees a barrier packet
Under rocprofv3 --kernel-trace --stats the number of __amd_rocclr_copyBuffer launches says which engine moved the bytes."""
import ctypes, os, sys, time, threading, queue
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int
mode = sys.argv[1]
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
HW = 1080 * 1920
d = torch.randint(0, 255, (8, 4 * HW), dtype=torch.uint8, device=dev)
h = torch.zeros((8, 4 * HW), dtype=torch.uint8).pin_memory()
s2 = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)
for _ in range(10): net()
torch.cuda.synchronize()
jobs = queue.Queue()


def worker():
    while True:
        j = jobs.get()
        if j is None:
            return
        ev, i = j
        ev.synchronize()
        hip.hipMemcpyAsync(h[i % 8].data_ptr(), d[i % 8].data_ptr(), 4 * HW, 2, s2.cuda_stream)

th = threading.Thread(target=worker, daemon=True)
th.start()


def body(i):
    if mode == "none":
        return
    if mode == "indep":
        hip.hipMemcpyAsync(h[i % 8].data_ptr(), d[i % 8].data_ptr(), 4 * HW, 2, s2.cuda_stream); return
    ev = torch.cuda.Event()
    if mode != "wait_only":
        ev.record(main)
    if mode == "hostwait":
        jobs.put((ev, i)); return
    if mode != "record_only":
        s2.wait_event(ev)
    if mode in ("event", "event_nocu"):
        hip.hipMemcpyAsync(h[i % 8].data_ptr(), d[i % 8].data_ptr(), 4 * HW, 2 if mode == "event" else 1024, s2.cuda_stream)


def timed(n=80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        net()
        body(i)
    main.synchronize()
    t1 = time.perf_counter()
    while not jobs.empty():
        time.sleep(0.0005)
    time.sleep(0.002)
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3

r = [timed() for _ in range(3)]
print("%-12s compute stream: %s ms / forward; all done: %s" % (mode, " ".join("%.3f" % a for a, _ in r), " ".join("%.3f" % b for _, b in r)), flush=True)
jobs.put(None)
