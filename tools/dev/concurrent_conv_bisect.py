"""Which conv launch goes wrong when two conv stacks share the GPU?  The human-parsing net (12 frames per forward) runs on a second stream
beside back-to-back matting forwards on the main stream; after every forward ALL buffers of both nets are compared (as bits) with snapshots
of a forward computed alone.  The first differing buffer in build order is the output of the launch that went wrong.
    python tools/dev/concurrent_conv_bisect.py [rounds]"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd import _native as nv
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
full_matting = len(sys.argv) > 2 and sys.argv[2] == "full"
nv.load()
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
from video_unscreen_amd.synth import SyntheticClip
dev = torch.device("cuda", 0)
net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
g = torch.Generator(device="cpu"); g.manual_seed(5)
x = torch.rand((544, 960, 8), generator=g); x[..., 7] = 0
net.x8.copy_(x.to(dev))
seg = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
seg.set_plan_batch(12)
clip = SyntheticClip(1080, 1920, 12, config_id=2, device=dev)
frames = [clip.frame(t)["frame"] for t in range(12)]
masks = [torch.empty((1080, 1920), dtype=torch.uint8, device=dev) for _ in range(12)]
pnet = seg.model_for(12)


def layers_of(descs_and_plans):
    out = []
    for d, p in descs_and_plans:
        out.append({"out": d.out, "inp": d.inp, "desc": "in %dx%dx%d -> %dx%dx%d taps %d stride %d batch %d, plan %s, %d workgroups" % (
            d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout, d.ntaps, d.in_stride, d.batch, p["plan"], p["workgroups"])})
    return out

m_layers = layers_of(zip(net._descs, nv.conv_plans(net._arr, len(net._descs))))
p_layers = []
for kind, a, arr in pnet._segments:
    if kind == "conv":
        p_layers += layers_of(zip(list(arr), nv.conv_plans(arr, len(arr))))


def bufs(n):
    return [t for t in n._keep if t.dtype == torch.float32 and t.numel() >= 1024]

m_bufs, p_bufs = bufs(net), bufs(pnet)
net(); torch.cuda.synchronize()
seg.forward_device_batch(frames, outs=masks); torch.cuda.synchronize()
m_snap = [t.clone() for t in m_bufs]
p_snap = [t.clone() for t in p_bufs]
mask_snap = torch.stack(masks).clone()
pred_snap = net.pred.clone()
print("matting net: %d buffers, %d launches; parsing net: %d buffers, %d launches" % (len(m_bufs), len(m_layers), len(p_bufs), len(p_layers)), flush=True)
main = torch.cuda.current_stream(dev)
s2 = torch.cuda.Stream(dev)


def flags(bs, snaps):
    return torch.stack([(b.view(torch.int32) != s.view(torch.int32)).sum() for b, s in zip(bs, snaps)])


def report(tag, fl, bs, layers, tally):
    idx = [i for i, v in enumerate(fl.tolist()) if v]
    if not idx:
        return
    i0 = idx[0]
    t = bs[i0]
    lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * 4
    who = [l["desc"] for l in layers if lo <= l["out"] < hi]
    print("%s: %d buffers differ, first #%d shape %s (%d values); written by: %s" % (tag, len(idx), i0, tuple(t.shape), int(fl[i0]), who or "no conv launch (glue / input)"), flush=True)
    for w in who or ["glue"]:
        tally[w] += 1
    snap = (m_snap if bs is m_bufs else p_snap)[i0]
    dm = (t.view(torch.int32) != snap.view(torch.int32))
    nz = dm.nonzero()
    lo_, hi_ = nz.min(0).values.tolist(), nz.max(0).values.tolist()
    print("   differing box (row, col, channel): %s .. %s; per distinct (row // rows-per-image): %s" % (lo_, hi_, sorted(set((nz[:, 0] // (t.shape[0] // 12 if bs is p_bufs else t.shape[0])).tolist()))))
    got, want = t[dm], snap[dm]
    print("   max |got - want| %.4g, max |want| %.4g, non-finite got: %d; samples got/want: %s" % (
        float((got - want).abs().nan_to_num(1e30).max()), float(want.abs().max()), int((~torch.isfinite(got)).sum()),
        [(round(float(a), 4), round(float(b), 4)) for a, b in zip(got[:6], want[:6])]), flush=True)
    # which 8 x 16 blocks / 32-channel groups
    if t.dim() == 3:
        hh = t.shape[0] // 12 if bs is p_bufs else t.shape[0]
        keys_ = sorted(set(zip((nz[:, 0] // hh).tolist(), ((nz[:, 0] % hh) // 8).tolist(), (nz[:, 1] // 16).tolist(), (nz[:, 2] // 32).tolist())))
        print("   (image, block row, block col, channel group) touched: %s" % keys_[:24], flush=True)

tally = collections.Counter()
for r in range(rounds):
    with torch.cuda.stream(s2):
        seg.forward_device_batch(frames, outs=masks)
        mk = (torch.stack(masks) != mask_snap).sum()
        pf = flags(p_bufs, p_snap)
    mfs, pr = [], []
    for k in range(10):
        net()
        pr.append((net.pred != pred_snap).sum())
        if full_matting:
            mfs.append(flags(m_bufs, m_snap))
    torch.cuda.synchronize()
    if int(mk):
        print("round %d: %d mask pixels differ" % (r, int(mk)), flush=True)
    report("round %d parsing" % r, pf, p_bufs, p_layers, tally)
    for k, v in enumerate(pr):
        if int(v):
            print("round %d matting forward %d: %d values of pred differ" % (r, k, int(v)), flush=True)
    for k, mf in enumerate(mfs):
        report("round %d matting forward %d" % (r, k), mf, m_bufs, m_layers, tally)
print("---- launches that went wrong first, by count")
for w, c in tally.most_common():
    print(c, w)
