import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.ndimage as ndi
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
H, W = 1080, 1920
f = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)
rng = np.random.default_rng(0)
clean = f["seg"].numpy().copy()
noisy = clean.copy(); noisy[ndi.binary_dilation(rng.random((H, W)) < 0.002, iterations=2)] = 255
speck = ((rng.random((H, W)) < 0.3) * 255).astype(np.uint8)
flt = nv.DeviceObjectFilter(); hflt = nv.ObjectFilter()
for name, m in (("clean silhouette", clean), ("silhouette + speckle", noisy), ("30% salt noise", speck)):
    d = torch.from_numpy(m).cuda()
    for _ in range(3): flt(d.clone(), None, (0.5, 0.5), 1e-5, 0.5)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): flt(d.clone(), None, (0.5, 0.5), 1e-5, 0.5)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    t = time.time()
    for _ in range(5): hflt(m.copy(), None, (0.5, 0.5), 1e-5, 0.5)
    ht = (time.time() - t) / 5
    print(f"{name:24s} device {dt*1e3:7.3f} ms   host {ht*1e3:7.3f} ms", flush=True)
