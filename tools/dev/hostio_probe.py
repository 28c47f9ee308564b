"""How do pinned-host copies overlap the compute?  H2D / D2H rates alone, together, and beside a saturating conv loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
dev = torch.device("cuda:0")
N = 40
for mb in (6.2, 24.9):
    n = int(mb * 1e6)
    hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    ds = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    up, down = torch.cuda.Stream(), torch.cuda.Stream()
    net = HipMattingNet(synthetic_state_dict(0), 544, 960)
    net.x8.normal_()

    def timed(fn, sync=True):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t

    def h2d():
        with torch.cuda.stream(up):
            for i in range(N): ds[i % 4].copy_(hs[i % 4], non_blocking=True)

    def d2h():
        with torch.cuda.stream(down):
            for i in range(N): hs[i % 4].copy_(ds[i % 4], non_blocking=True)

    def compute(k=60):
        for _ in range(k): net()

    t_up, t_dn = timed(h2d), timed(d2h)
    t_both = timed(lambda: (h2d(), d2h()))
    t_c = timed(compute)
    t_all = timed(lambda: (compute(), h2d(), d2h()))
    t_all2 = timed(lambda: (h2d(), d2h(), compute()))
    print(f"{mb} MB buffers x {N}: H2D {n*N/t_up/1e9:.1f} GB/s, D2H {n*N/t_dn/1e9:.1f} GB/s, both at once {2*n*N/t_both/1e9:.1f} GB/s total")
    print(f"   60 matting forwards alone {t_c*1e3:.1f} ms; with both copy streams busy beside them: {t_all*1e3:.1f} ms (copies alone {t_both*1e3:.1f} ms); copies enqueued first: {t_all2*1e3:.1f} ms")
