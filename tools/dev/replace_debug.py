import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_replace import _inputs, dev
from video_unscreen_amd.pipeline_replace import ReplacePipeline
H, W, sh, sw = 120, 200, 150, 240
fg, m3, bg, src = _inputs(H, W, sh, sw, 7 * H + W)
pipe = ReplacePipeline(H, W, dev(bg), scale=1.2)
for dx, dy in ((-0.2072, -2.6028), (0.0, 0.0)):
    M = pipe.shift_matrix(dx, dy)
    res, cmp_img = pipe.compose(dev(fg), dev(m3), dev(src), M)
    res_c, cmp_c = pipe.compose_chain(dev(fg), dev(m3), dev(src), M)
    d = (res != res_c).any(-1).cpu().numpy()
    ys, xs = np.nonzero(d)
    print("shift", dx, dy, "diff px", d.sum(), "of", d.size, "cmp equal", torch.equal(cmp_img[:, :W], cmp_c[:, :W]))
    if len(ys):
        print(" rows", np.unique(ys)[:40], "\n cols", np.unique(xs)[:80])
        for y, x in list(zip(ys, xs))[:6]:
            print("  ", y, x, res[y, x].tolist(), res_c[y, x].tolist())
import oracle
from video_unscreen_amd import _native as nv
M = pipe.shift_matrix(-0.2072, -2.6028)
res, _ = pipe.compose(dev(fg), dev(m3), dev(src), M)
res_c, _ = pipe.compose_chain(dev(fg), dev(m3), dev(src), M)
fs = oracle.rescale_fg(oracle.warp_affine(fg, M, H, W), 1.2)
ms = oracle.rescale_fg(oracle.warp_affine(m3, M, H, W), 1.2)
want = oracle.blend(fs, ms, bg)
print("fused vs oracle", (res.cpu().numpy() != want).sum(0).sum(0), "chain vs oracle", (res_c.cpu().numpy() != want).sum(0).sum(0))
w_d = nv.warp_affine(dev(m3), M, H, W)
print("warp mask vs oracle", (w_d.cpu().numpy() != oracle.warp_affine(m3, M, H, W)).sum(0).sum(0))
r_d = nv.rescale_fg(w_d, 1.2)
print("rescale mask vs oracle", (r_d.cpu().numpy() != ms).sum(0).sum(0))
fd = nv.rescale_fg(nv.warp_affine(dev(fg), M, H, W), 1.2)
print("fg chain vs oracle", (fd.cpu().numpy() != fs).sum(0).sum(0))
b_d = nv.blend(fd, r_d, dev(bg))
print("blend vs oracle", (b_d.cpu().numpy() != oracle.blend(fd.cpu().numpy(), r_d.cpu().numpy(), bg)).sum(0).sum(0))
got = res.cpu().numpy()
for perm in ([0, 1, 1], [0, 1, 0], [0, 0, 0], [1, 2, 0], [2, 1, 0]):
    w2 = oracle.blend(fs, np.ascontiguousarray(ms[..., perm]), bg)
    print("mask perm", perm, (got != w2).sum(0).sum(0))
for perm in ([0, 1, 1], [0, 1, 0], [2, 1, 0]):
    w2 = oracle.blend(np.ascontiguousarray(fs[..., perm]), ms, bg)
    print("fg perm", perm, (got != w2).sum(0).sum(0))
w2 = oracle.blend(fs, ms, np.ascontiguousarray(bg[..., [0, 1, 1]]))
print("bg perm 011", (got != w2).sum(0).sum(0))
