"""Pinned host -> device copy rate against copy size and the number of streams the copies alternate over (are two 50 MB copies in flight on two
streams as fast as one 100 MB copy?).   python tools/dev/h2d_streams_probe.py"""
import time, torch
dev = torch.device("cuda:0")
for mb in (12, 25, 50, 100):
    n = mb * 1000 * 1000
    cnt = max(8, 1600 // mb)
    host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
    devb = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
    for ns in (1, 2, 3):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(cnt):
                with torch.cuda.stream(streams[i % ns]):
                    devb[i % 4].copy_(host[i % 4], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        print("H2D %3d MB copies, %d stream(s): %5.1f GB/s" % (mb, ns, cnt * n / dt / 1e9), flush=True)
    # and with a download running beside the uploads
    ds = torch.cuda.Stream(dev)
    for ns in (1, 2):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(cnt):
            with torch.cuda.stream(streams[i % ns]):
                devb[i % 2].copy_(host[i % 2], non_blocking=True)
            with torch.cuda.stream(ds):
                host[2 + i % 2].copy_(devb[2 + i % 2], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("H2D %3d MB on %d stream(s) + D2H of the same size beside it: %5.1f GB/s each way" % (mb, ns, cnt * n / dt / 1e9), flush=True)
