"""ReplaceHostLoop (both passes, pinned host -> pinned host, 1080p): frames/s against `group` (frames per PCIe copy) and `ring`.
    python tools/dev/replace_group_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.pipeline_replace import ReplaceHostLoop, ReplacePipeline
H, W, K = 1080, 1920, 240
g = torch.Generator().manual_seed(0)
masks_h = torch.randint(0, 256, (K, 2, H, W, 3), dtype=torch.uint8, generator=g).pin_memory()
inputs_h = torch.randint(0, 256, (K, 2, H, W, 3), dtype=torch.uint8, generator=g).pin_memory()
compare_h = torch.empty((K, H, 2 * W, 3), dtype=torch.uint8).pin_memory()
bg = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=g).cuda()
pipe = ReplacePipeline(H, W, bg, device="cuda:0", scale=1.2, max_frames=K)
for ring, group in ((3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (3, 12), (3, 16), (2, 24)):
    loop = ReplaceHostLoop(pipe, K, ring=ring, group=group)
    def job(k):
        sx, sy, n = loop.pass1(masks_h[:k])
        loop.pass2(inputs_h[:k], pipe.shift_matrix(sx / n, sy / n), compare_h)
    job(2 * group); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); job(K); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("ring %d group %2d: %7.1f frames/s (%.1f GB/s up, %.1f GB/s down)" % (ring, group, K / best, K * 4 * H * W * 3 / best / 1e9, K * 2 * H * W * 3 / best / 1e9), flush=True)
    del loop
    torch.cuda.empty_cache()
