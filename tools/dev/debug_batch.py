import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
sd = schp_synthetic_state_dict(0)
n1 = HipSchpNet(sd, 473, 473, batch=1)
n2 = HipSchpNet(sd, 473, 473, batch=2)
g = torch.Generator().manual_seed(0)
x = torch.randn(473, 473, 4, generator=g).cuda(); x[..., 3] = 0
y = torch.randn(473, 473, 4, generator=g).cuda(); y[..., 3] = 0
n1.x4.copy_(x); n1.run()
n2.image(n2.x4, 0).copy_(x); n2.image(n2.x4, 1).copy_(y); n2.run()
b1 = [t for t in n1._keep if t.dim() == 3 and t.dtype == torch.float32]
b2 = [t for t in n2._keep if t.dim() == 3 and t.dtype == torch.float32]
print(len(b1), len(b2))
k = 0
for i, (a, b) in enumerate(zip(b1, b2)):
    if a.shape[1:] != b.shape[1:] or b.shape[0] != 2 * a.shape[0]:
        continue
    bb = b[:a.shape[0]]
    eq = torch.equal(a, bb)
    if not eq:
        d = (a - bb).abs().max().item()
        print(i, tuple(a.shape), "DIFF", d)
        k += 1
        if k > 5: break
n1.x4.copy_(y); n1.run()
b1 = [t for t in n1._keep if t.dim() == 3 and t.dtype == torch.float32]
k = 0
for i, (a, b) in enumerate(zip(b1, b2)):
    if a.shape[1:] != b.shape[1:] or b.shape[0] != 2 * a.shape[0]:
        continue
    bb = b[a.shape[0]:]
    if not torch.equal(a, bb):
        print("img1", i, tuple(a.shape), "DIFF", (a - bb).abs().max().item(), "rows differing:", (a != bb).any(-1).any(-1).nonzero().flatten()[:8].tolist())
        k += 1
        if k > 4: break
print("done")
