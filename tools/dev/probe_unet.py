"""GPU probe: MIOpen first-call cost and steady-state time of the matting net at 544x960."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.vmatting.net import synthetic_state_dict, MattingNet

def main():
    dev = "cuda:0"
    print(torch.cuda.get_device_name(0), flush=True)
    sd = synthetic_state_dict(0)
    x = torch.randn(1, 7, 544, 960, device=dev)
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        net = MattingNet(sd, device=dev, dtype=dtype)
        t = time.time(); y = net(x); torch.cuda.synchronize(); t1 = time.time() - t
        t = time.time(); y = net(x); torch.cuda.synchronize(); t2 = time.time() - t
        n = 20
        t = time.time()
        for _ in range(n):
            y = net(x)
        torch.cuda.synchronize()
        t3 = (time.time() - t) / n
        print(f"{dtype}: first {t1:.2f}s second {t2*1e3:.1f}ms steady {t3*1e3:.2f}ms  -> {132.75/t3/1e3:.1f} TFLOP/s", flush=True)
        if dtype == torch.float32:
            ref = y.clone()
        else:
            print("   maxdiff vs fp32:", (y - ref).abs().max().item(), flush=True)
    # graph capture of fp32
    net = MattingNet(sd, device=dev, dtype=torch.float32)
    y = net(x); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        net(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        yg = net(x)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph fp32 steady {(time.time()-t)/20*1e3:.2f}ms", flush=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=80), flush=True)

main()
